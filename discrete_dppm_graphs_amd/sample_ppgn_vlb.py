"""Mirror of the reference's sample_ppgn_vlb.py (posterior-mix reverse chain).  Shares the outer loop with
sample_ppgn_simple; the per-batch chain is engine.sample_chain_vlb (sample_ppgn_vlb.py:79-131)."""
from . import engine
from . import sample_ppgn_simple as _s
from .engine import sigma_lin  # noqa: F401  (reference sample_ppgn_vlb.py:28-35)


def posterior(sigmatilde_t, sigma_t, sigmatilde_t1, x0, xt):
    """q(x_{t-1} = 1 | x_t, x_0) for one binary edge, by Bayes' rule: prior q(x_{t-1}=1 | x_0) times likelihood
    q(x_t | x_{t-1}=1) over evidence q(x_t | x_0).  Same name and argument order as the scalar helper at reference
    sample_ppgn_vlb.py:17-25 (which the reference's sampler never calls either); the kernels evaluate the tensor form."""
    prior = 1 - sigmatilde_t1 if x0 > 0.5 else sigmatilde_t1
    likelihood = 1 - sigma_t if xt > 0.5 else sigma_t
    evidence = 1 - sigmatilde_t if (xt > 0.5) == (x0 > 0.5) else sigmatilde_t
    return prior * likelihood / evidence


def sample_main(config, modellink, epoch, noise_num):
    """reference sample_ppgn_vlb.py:38-146."""
    return _s.sample_main(config, modellink, epoch, noise_num, chain=engine.sample_chain_vlb)


def sample_testing(config, modellink, epoch, noise_num, train_dl):
    """reference sample_ppgn_vlb.py:149-259."""
    return _s.sample_testing(config, modellink, epoch, noise_num, train_dl, chain=engine.sample_chain_vlb)
