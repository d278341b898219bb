"""Mirror of the reference's ppgn_simple.py: `loss_func_bce`, `fit`, `train_main` and the CLI.

`fit` keeps the reference's epoch bookkeeping (ExponentialLR per epoch, three checkpoint flavours, periodic sampling
for model selection) but runs each optimiser step through engine.train_step_simple: one batched denoiser call, fused
noising / mask / loss kernels, flat-bucket Adam, and -- under torch.distributed (one process per GPU, RCCL) -- one
gradient all-reduce per step with every rank holding a disjoint slice of the global batch."""
import logging
import os
import time

import numpy as np
import torch

from . import engine, ops
from .sample_ppgn_simple import sample_main, sample_testing
from .utils.arg_helper import edict2dict, get_config, load_data, parse_arguments, process_config, set_seed_and_logger
from .utils.easydict import EasyDict as edict
from .utils.loading_utils import get_score_model
from .utils.resident import STREAM_NOISE, batch_sizes_of, host_rng, upload

_SAMPLERS = (sample_main, sample_testing)


class _BceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, target, sigmas, mask):
        loss, dscore = ops.loss_bce(score.contiguous(), target.contiguous(), sigmas, mask.contiguous(), want_grad=True)
        ctx.save_for_backward(dscore)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dscore,) = ctx.saved_tensors
        return dscore * g.to(dscore.dtype), None, None, None


def loss_func_bce(score_list, grad_log_q_noise_list, sigma_list, config, mask):
    """reference ppgn_simple.py:25-34 (float64 scalar, differentiable w.r.t. score_list)."""
    return _BceLoss.apply(score_list, grad_log_q_noise_list, list(sigma_list), mask)


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def shard_bounds(Bg, rank, world):
    """Contiguous slice [lo, hi) of a global batch of Bg graphs owned by `rank` (whole graphs are the unit of sharding)."""
    return (Bg * rank) // world, (Bg * (rank + 1)) // world


def _save(config, model, state, scheduler_state, sigma_list, epoch, train_loss, test_loss, sub=""):
    to_save = {'model': model.state_dict(), 'sigma_list': [float(s) for s in sigma_list], 'config': edict2dict(config),
               'epoch': epoch, 'train_loss': float(train_loss), 'test_loss': float(test_loss),
               'train_loss_item': float('nan'), 'test_loss_item': float('nan'),
               'optimizer_state_dict': state.state_dict(), 'scheduler_state_dict': scheduler_state}
    d = os.path.join(config.model_save_dir, sub) if sub else config.model_save_dir
    os.makedirs(d, exist_ok=True)
    torch.save(to_save, os.path.join(d, f"{config.dataset.name}.pth"))


def _resume(resume, model, state, config, lr0, gamma):
    """Load a checkpoint written by _save into (model, state); returns (lr0, gamma, first_epoch)."""
    ckp = torch.load(resume, map_location=config.dev, weights_only=True) if isinstance(resume, (str, os.PathLike)) else resume
    model.load_state_dict(ckp['model'], strict=False)
    state.load_state_dict(ckp['optimizer_state_dict'])
    sched = ckp.get('scheduler_state_dict') or {}
    lr0 = float(sched.get('base_lr', lr0))
    gamma = float(sched.get('gamma', gamma))
    first_epoch = int(ckp['epoch']) + 1
    state.lr = lr_after_epoch(lr0, gamma, first_epoch - 1)
    logging.info(f"resumed from epoch {ckp['epoch']} (step {state.step_count}, lr {state.lr:.3e})")
    return lr0, gamma, first_epoch


def lr_after_epoch(lr0, gamma, epoch):
    """Learning rate in force once epoch `epoch` has finished (epoch = -1: before the first one).  The reference builds
    ExponentialLR and calls scheduler.step(epoch) after every epoch (ppgn_simple.py:56,95), which sets the closed form
    lr0 * gamma**epoch: epochs 0 and 1 both run at lr0, epoch 2 at lr0 * gamma, ..."""
    return lr0 * gamma ** max(epoch, 0)


def _epoch_loop(model, state, dl, config, step_fn, train, epoch, counters):
    """One pass over a loader; each rank processes its slice of every global batch.
    The host randomness of the whole pass (one noise draw per graph) is made up front by counters["plan"] from a generator
    keyed by (seed, epoch, train / test) -- identical on every rank, whatever the world size -- and uploaded once; with a
    ResidentLoader no step copies anything from the host (config.assert_no_host_sync = True turns every synchronising
    host<->device transfer after the first step of a pass into an error: tests/test_gpu_resident.py).
    A rank whose slice of a (tail) batch is empty still takes the step with a zero gradient, so the all-reduce of the
    other ranks is always matched."""
    dist, rank, world = _dist()
    sizes = batch_sizes_of(dl)
    pass_no = 2 * epoch + (0 if train else 1)
    plan = counters["plan"](int(sum(sizes)), pass_no, config.dev)     # device tensor, leading dimension = graphs of the pass
    guard = bool(getattr(config, "assert_no_host_sync", False)) and torch.cuda.is_available()
    old_mode = torch.cuda.get_sync_debug_mode() if guard else 0
    losses, first = [], 0
    try:
        for step, (adj_b, _x_b) in enumerate(dl):
            Bg = adj_b.size(0)
            lo, hi = shard_bounds(Bg, rank, world)
            adj = adj_b[lo:hi].to(config.dev).contiguous()
            # Philox counters: graph id = position of the graph in this pass, call offset = (pass, step); a resumed run
            # continues the streams of the run it resumes
            loss = step_fn(model, state, adj, plan[first + lo:first + hi], Bg, first + lo, (pass_no << 12) + (step & 0xFFF), train)
            first += Bg
            losses.append(loss)
            if guard:
                torch.cuda.set_sync_debug_mode("error")
    finally:
        if guard:
            torch.cuda.set_sync_debug_mode(old_mode)
    if not losses:
        return float('nan')
    tot = torch.stack([l.double() for l in losses])
    if world > 1:
        dist.all_reduce(tot)
    return float(tot.mean().item())                          # the only host sync of the epoch


def fit(model, optimizer, mcmc_sampler, train_dl, max_node_number, max_epoch=20, config=None, save_interval=50,
        sample_interval=1, sigma_length=None, sample_from_sigma_delta=0.0, test_dl=None, samplers=_SAMPLERS, resume=None):
    """reference ppgn_simple.py:37-237.  `optimizer` may be a torch.optim.Adam (its hyper-parameters are adopted; the
    step itself runs in the fused HIP Adam over the flat parameter bucket) or an engine.FlatState.
    resume: a checkpoint dict written by this module (or its path): weights, Adam moments, step count, the learning-rate
    schedule and the epoch counter continue where they stopped (the reference saves these fields but never reads them
    back, ppgn_simple.py:138-171)."""
    dist, rank, world = _dist()
    if isinstance(optimizer, engine.FlatState):
        state = optimizer
    else:
        g = optimizer.param_groups[0]
        state = engine.FlatState(model, lr=g['lr'], betas=tuple(g['betas']), eps=g['eps'], weight_decay=g['weight_decay'])
    lr0, gamma = state.lr, config.train.lr_dacey
    first_epoch = 0
    if resume is not None:
        lr0, gamma, first_epoch = _resume(resume, model, state, config, lr0, gamma)
    if hasattr(train_dl, "epoch"):
        train_dl.epoch = first_epoch                         # shuffle order of a ResidentLoader is keyed by the epoch
    best_score, best_score_loss = np.inf, np.inf
    lastmmd = {n: {"degree": 0, "cluster": 0, "orbit": 0.0} for n in config.num_levels}
    seed = int(getattr(config, "seed", 0))

    def plan(n, pass_no, dev):
        # sigma_b ~ U(0, 1/2), one per graph (ppgn_simple.py:67 draws them from numpy's unseeded global generator); here they
        # come from a generator keyed by (seed, pass), so every rank holds the same draws for the same global graph
        return upload(host_rng(seed, STREAM_NOISE, pass_no).uniform(low=0.0, high=0.5, size=n), dev, torch.float64)

    counters = {"plan": plan}

    def step_fn(model, state, adj, sigmas, Bg, gid0, offset, train):
        if train:
            return engine.train_step_simple(model, state, adj, sigmas, seed, offset, gid0, B_global=Bg, lr=state.lr)
        if adj.shape[0] == 0:
            return engine.empty_step(state, adj, torch.float64, train=False)
        flags = engine.node_flags(adj)
        A_t, target = ops.edgeflip(adj, flags, sigmas, seed=seed + 1, graph_id0=gid0, offset=offset)
        mask = ops.generate_mask(flags)
        loss, _ = ops.loss_bce(model.infer(A_t, mask, sigmas), target, sigmas, mask, B_global=Bg, want_grad=False)
        return loss

    sigma_list = [0.0]
    for epoch in range(first_epoch, max_epoch):
        t_start = time.time()
        model.train()
        mean_train_loss = _epoch_loop(model, state, train_dl, config, step_fn, True, epoch, counters)
        state.lr = lr_after_epoch(lr0, gamma, epoch)         # scheduler.step(epoch) (:56,95)
        model.eval()
        mean_test_loss = _epoch_loop(model, state, test_dl, config, step_fn, False, epoch, counters) if test_dl is not None else float('nan')
        logging.info(f'epoch: {epoch:03d}| time: {time.time() - t_start:.1f}s| train loss: {mean_train_loss:+.3e} | '
                     f'test loss: {mean_test_loss:+.3e} | ')
        sched_state = {'gamma': gamma, 'last_epoch': epoch, 'base_lr': lr0}
        if rank == 0:
            if epoch % save_interval == save_interval - 1:
                _save(config, model, state, sched_state, sigma_list, epoch, mean_train_loss, mean_test_loss)
            if mean_train_loss < best_score_loss:
                best_score_loss = mean_train_loss
                _save(config, model, state, sched_state, sigma_list, epoch, mean_train_loss, mean_test_loss, "bestloss")
        if world > 1:
            dist.barrier()
        if epoch % sample_interval == sample_interval - 1 and config.eval_from < epoch:
            results = {}
            for num_noiselevel in config.num_levels:
                results = samplers[1](config, f"{config.model_save_dir}", epoch, num_noiselevel, train_dl)
                lastmmd[num_noiselevel] = results
            logging.info({f"{k}_mmd": v for k, v in results.items()})
            score = sum(v if "likelyhood" not in k else 1 - v for k, v in results.items())
            if score < best_score and rank == 0:
                best_score = score
                _save(config, model, state, sched_state, sigma_list, epoch, best_score, mean_test_loss, "best")
        if epoch % config.finalinterval == config.finalinterval - 1 and config.eval_from < epoch:
            for sub in ("bestloss", "", "best"):
                d = os.path.join(config.model_save_dir, sub) if sub else config.model_save_dir
                if os.path.isdir(d) and any(f.endswith(".pth") for f in os.listdir(d)):
                    results = samplers[0](config, d, epoch, config.num_levels[-1])
                    logging.info(f"MMD of Epoch {epoch} ({sub or 'main'}): {results}")
    return state


def train_main(config, args):
    """reference ppgn_simple.py:240-277."""
    config.train.sigmas = np.linspace(0, 0.5, config.num_levels[0] + 1).tolist()
    set_seed_and_logger(config, args)
    train_dl, test_dl = load_data(config)
    model = get_score_model(config)
    total_params = sum(p.numel() for p in model.parameters())
    logging.info(f"Parameters Count: {total_params}, Trainable: {sum(p.numel() for p in model.parameters() if p.requires_grad)}")
    state = engine.FlatState(model, lr=config.train.lr_init, betas=(0.9, 0.999), eps=1e-8, weight_decay=config.train.weight_decay)
    return fit(model, state, None, train_dl, max_node_number=config.dataset.max_node_num, max_epoch=config.train.max_epoch,
               config=config, save_interval=config.train.save_interval, sample_interval=config.train.sample_interval,
               sigma_length=len(config.train.sigmas), sample_from_sigma_delta=0.0, test_dl=test_dl,
               resume=getattr(args, "resume", None) or None)


def init_distributed():
    """One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE): pick the rank's device and join
    the RCCL process group BEFORE the configuration is processed, so that rank 0 names the run directory for everyone."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))


def main(argv=None):
    args = parse_arguments('train_ego_small.yaml', argv)
    init_distributed()
    ori_config_dict = get_config(args)
    config_dict = edict(ori_config_dict.copy())
    config_dict.model.name = "ppgn"          # reference ppgn_simple.py:285
    train_main(config_dict, args)


if __name__ == "__main__":
    main()
