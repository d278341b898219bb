"""Mirror of the reference's ppgn_vlb.py: `sigma_lin`, `loss_func_kld`, `fit`, `train_main` for the VLB objective.

As shipped, the reference's train_main passes keywords its fit does not accept and fit reads undefined names
(SURVEY.md section 2 #9); this module is the working counterpart with the same structure as ppgn_simple."""
import logging
import os

import numpy as np
import torch

from . import engine, ops
from . import ppgn_simple as _simple
from .engine import sigma_lin  # noqa: F401   reference ppgn_vlb.py:27-34
from .sample_ppgn_vlb import sample_main, sample_testing
from .utils.arg_helper import get_config, load_data, parse_arguments, set_seed_and_logger
from .utils.easydict import EasyDict as edict
from .utils.loading_utils import get_score_model
from .utils.resident import STREAM_NOISE, host_rng, upload


class _VlbLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, A_t, A_0, sched, mask):
        loss, dscore = ops.loss_vlb(score.contiguous(), A_t.contiguous(), A_0.contiguous(), sched, mask.contiguous())
        ctx.save_for_backward(dscore)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dscore,) = ctx.saved_tensors
        return dscore * g, None, None, None, None


def loss_func_kld(score_list, train_noise_adj_b, train_adj_b, grad_log_q_noise_list, sigma_list, sigma_ind_list,
                  sigma_nontild_list, config, mask):
    """reference ppgn_vlb.py:79-125 (float32 scalar, differentiable w.r.t. score_list; the flip target is unused there too)."""
    st = np.asarray(sigma_list, dtype=np.float64)
    sched = np.stack([st, np.asarray(sigma_nontild_list, dtype=np.float64), st - st / np.asarray(sigma_ind_list, dtype=np.float64)], 1)
    return _VlbLoss.apply(score_list, train_noise_adj_b, train_adj_b, sched, mask)


def fit(model, optimizer, mcmc_sampler, train_dl, max_node_number, max_epoch=20, config=None, save_interval=50,
        sample_interval=1, test_dl=None, eval_set=None, resume=None, **_ignored):
    """reference ppgn_vlb.py:127-331; epoch bookkeeping shared with ppgn_simple.fit, step = engine.train_step_vlb."""
    T = config.num_levels[0]
    seed = int(getattr(config, "seed", 0))

    def plan(n, pass_no, dev):
        # t_b ~ UniformInt[1, T], one per graph (np.random.random_integers(1, T, B) in the reference, :161), from a generator
        # keyed by (seed, pass) and identical on every rank; uploaded as the [n, 3] table (sigmatilde_t, beta_t, sigmatilde_{t-1})
        t_idx = host_rng(seed, STREAM_NOISE, pass_no).integers(1, T + 1, size=n)
        return upload(engine.vlb_batch_schedule(T, t_idx), dev, torch.float64)

    def step_fn(model, state, adj, sched, Bg, gid0, offset, train):
        if train:
            return engine.train_step_vlb(model, state, adj, None, T, seed, offset, gid0, lr=state.lr, sched=sched)
        if adj.shape[0] == 0:
            return engine.empty_step(state, adj, torch.float32, train=False)
        flags = engine.node_flags(adj)
        sig = sched[:, 0].contiguous()
        A_t, _ = ops.edgeflip(adj, flags, sig, seed=seed + 1, graph_id0=gid0, offset=offset, want_target=False)
        mask = ops.generate_mask(flags)
        loss, _ = ops.loss_vlb(model.infer(A_t, mask, sig), A_t, adj, sched, mask, want_grad=False)
        return loss

    return _fit_generic(model, optimizer, train_dl, max_epoch, config, save_interval, sample_interval, test_dl, plan, step_fn, resume)


def _fit_generic(model, optimizer, train_dl, max_epoch, config, save_interval, sample_interval, test_dl, plan, step_fn, resume=None):
    import time
    dist, rank, world = _simple._dist()
    if isinstance(optimizer, engine.FlatState):
        state = optimizer
    else:
        g = optimizer.param_groups[0]
        state = engine.FlatState(model, lr=g['lr'], betas=tuple(g['betas']), eps=g['eps'], weight_decay=g['weight_decay'])
    lr0, gamma = state.lr, config.train.lr_dacey
    first_epoch = 0
    if resume is not None:
        lr0, gamma, first_epoch = _simple._resume(resume, model, state, config, lr0, gamma)
    if hasattr(train_dl, "epoch"):
        train_dl.epoch = first_epoch
    best_score, best_score_loss = np.inf, np.inf
    counters = {"plan": plan}
    for epoch in range(first_epoch, max_epoch):
        t0 = time.time()
        model.train()
        tr = _simple._epoch_loop(model, state, train_dl, config, step_fn, True, epoch, counters)
        state.lr = _simple.lr_after_epoch(lr0, gamma, epoch)
        model.eval()
        te = _simple._epoch_loop(model, state, test_dl, config, step_fn, False, epoch, counters) if test_dl is not None else float('nan')
        logging.info(f'epoch: {epoch:03d}| time: {time.time() - t0:.1f}s| train loss: {tr:+.3e} | test loss: {te:+.3e} | ')
        sched_state = {'gamma': gamma, 'last_epoch': epoch, 'base_lr': lr0}
        if rank == 0:
            if epoch % save_interval == save_interval - 1:
                _simple._save(config, model, state, sched_state, [0.0], epoch, tr, te)
            if tr < best_score_loss:
                best_score_loss = tr
                _simple._save(config, model, state, sched_state, [0.0], epoch, tr, te, "bestloss")
        if world > 1:
            dist.barrier()
        if epoch % sample_interval == sample_interval - 1 and config.eval_from < epoch:
            results = {}
            for n in config.num_levels:
                results = sample_testing(config, f"{config.model_save_dir}", epoch, n, train_dl)
            score = sum(v if "likelyhood" not in k else 1 - v for k, v in results.items())
            if score < best_score and rank == 0:
                best_score = score
                _simple._save(config, model, state, sched_state, [0.0], epoch, best_score, te, "best")
        if epoch % config.finalinterval == config.finalinterval - 1 and config.eval_from < epoch:
            for sub in ("best", "", "bestloss"):
                d = os.path.join(config.model_save_dir, sub) if sub else config.model_save_dir
                if os.path.isdir(d) and any(f.endswith(".pth") for f in os.listdir(d)):
                    logging.info(f"MMD of Epoch {epoch} ({sub or 'main'}): {sample_main(config, d, epoch, config.num_levels[-1])}")
    return state


def train_main(config, args):
    """reference ppgn_vlb.py:337-382."""
    config.train.sigmas = np.linspace(0, 0.5, config.num_levels[0] + 1).tolist()
    set_seed_and_logger(config, args)
    train_dl, test_dl = load_data(config)
    model = get_score_model(config)
    state = engine.FlatState(model, lr=config.train.lr_init, betas=(0.9, 0.999), eps=1e-8, weight_decay=config.train.weight_decay)
    return fit(model, state, None, train_dl, max_node_number=config.dataset.max_node_num, max_epoch=config.train.max_epoch,
               config=config, save_interval=config.train.save_interval, sample_interval=config.train.sample_interval,
               test_dl=test_dl, resume=getattr(args, "resume", None) or None)


def main(argv=None):
    args = parse_arguments('train_ego_small.yaml', argv)
    _simple.init_distributed()
    config_dict = edict(get_config(args).copy())
    config_dict.model.name = "ppgn"
    train_main(config_dict, args)


if __name__ == "__main__":
    main()
