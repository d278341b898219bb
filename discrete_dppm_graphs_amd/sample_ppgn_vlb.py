"""Mirror of the reference's sample_ppgn_vlb.py (posterior-mix reverse chain).  Shares the outer loop with
sample_ppgn_simple; the per-batch chain is engine.sample_chain_vlb (sample_ppgn_vlb.py:79-131)."""
from . import engine
from . import sample_ppgn_simple as _s
from .engine import sigma_lin  # noqa: F401  (reference sample_ppgn_vlb.py:28-35)


def posterior(sigmatilde_t, sigma_t, sigmatilde_t1, x0, xt):
    """reference sample_ppgn_vlb.py:17-25 (scalar form, unused by the sampler itself)."""
    if xt < 0.01 and x0 < 0.01:
        return sigmatilde_t1 * sigma_t / (1 - sigmatilde_t)
    elif xt > 0.99 and x0 < 0.01:
        return sigmatilde_t1 * (1 - sigma_t) / (sigmatilde_t)
    elif xt > 0.99 and x0 > 0.99:
        return (1 - sigmatilde_t1) * (1 - sigma_t) / (1 - sigmatilde_t)
    if xt < 0.01 and x0 > 0.99:
        return (1 - sigmatilde_t1) * sigma_t / (sigmatilde_t)


def sample_main(config, modellink, epoch, noise_num):
    """reference sample_ppgn_vlb.py:38-146."""
    return _s.sample_main(config, modellink, epoch, noise_num, chain=engine.sample_chain_vlb)


def sample_testing(config, modellink, epoch, noise_num, train_dl):
    """reference sample_ppgn_vlb.py:149-259."""
    return _s.sample_testing(config, modellink, epoch, noise_num, train_dl, chain=engine.sample_chain_vlb)
