"""Mirror of the reference's sample_ppgn_simple.py: `sample_main(config, modellink, epoch, noise_num)` and
`sample_testing(config, modellink, epoch, noise_num, train_dl)` for the "simple" sampler.

The per-batch work (x_T init, T x [re-noise, PPGN forward, reverse step]) runs in engine.sample_chain_simple on the GPU
with no host round trip; batches are independent, so under torch.distributed each rank takes every world_size-th batch
(graph ids are global, hence the union of all ranks' samples does not depend on the number of GPUs) and rank results
are gathered once at the end as uint8 adjacencies."""
import logging

import numpy as np
import torch

from . import engine
from .evaluation.stats import adjs_to_graphs, eval_graph_list
from .utils.arg_helper import graphs_to_tensor, load_data, load_model
from .utils.loading_utils import prepare_test_model_train
from .utils.resident import STREAM_POOL, host_rng, upload

_CALLS = {"n": 0}


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def _flags_from_adjs(base_adjs):
    return base_adjs.sum(-1).gt(1e-5).to(dtype=torch.float32)


def my_batches(n_batches, rank, world):
    """Batches are dealt round-robin: rank r runs batches r, r + world, ...  (no data-path collective)."""
    return list(range(rank, n_batches, world))


def gather_batches(mine, n_batches, B, rank, world, dist):
    """All ranks' uint8 adjacencies [k_r * B, N, N] -> [n_batches * B, N, N] in global batch order on every rank.
    One all_gather of equal-size (padded) buffers at the very end of sampling; the only collective of the path."""
    if world == 1:
        return mine
    per = -(-n_batches // world)
    pad = torch.zeros(per * B, *mine.shape[1:], dtype=mine.dtype, device=mine.device)
    pad[:mine.shape[0]] = mine
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = []
    for bi in range(n_batches):
        r, k = bi % world, bi // world
        out.append(bufs[r][k * B:(k + 1) * B])
    return torch.cat(out)


def run_sample(config, model, pool_adjs, eval_len, noise_num, chain, ref_graphs, methods=None):
    """reference sample_ppgn_simple.py:70-99 (run_sample).  pool_adjs: [G, N, N] padded training adjacencies from which
    node counts are drawn (gen_init_data, :24-41): moved to the device once (a ResidentLoader's pool already lives there),
    reduced to node flags [G, N] once, and indexed there.  The pool indices of ALL batches of the call come from one host
    generator keyed by (seed, call number) -- every rank draws the same table and keeps its own batches -- and cross to the
    device in one copy before the first chain starts."""
    dist, rank, world = _dist()
    B = config.test.batch_size
    dev = torch.device(config.dev)
    call = _CALLS["n"]
    _CALLS["n"] += 1
    seed = int(getattr(config, "seed", 0)) + 7919 * call
    n_batches = -(-eval_len // B)
    idx = host_rng(int(getattr(config, "seed", 0)), STREAM_POOL, call).integers(0, len(pool_adjs), size=(n_batches, B))
    m = engine.chain_width(model, pool_adjs.shape[-1], B, -(-n_batches // world))
    n_graphs = n_batches * B
    if m > 1:       # small graphs: m consecutive batches per chain call (same graph ids, same samples)
        flat = idx.reshape(-1)
        n_wide = -(-n_batches // m)
        if n_wide * m * B > flat.size:                                           # ragged tail: pad with repeats, cut after the gather
            flat = np.concatenate([flat, np.resize(flat[-((n_batches - 1) % m + 1) * B:], n_wide * m * B - flat.size)])
        idx, n_batches, B = flat.reshape(n_wide, m * B), n_wide, m * B
    pool_flags = _flags_from_adjs(pool_adjs.to(dev)).contiguous()               # [G, N], resident
    idx_dev = upload(idx, dev, torch.int64)
    mine = []
    with torch.no_grad():
        for bi in my_batches(n_batches, rank, world):
            flags = pool_flags.index_select(0, idx_dev[bi])
            adjs = chain(model, flags, noise_num, seed, graph_id0=bi * B)
            mine.append(adjs.to(torch.uint8))
    N = pool_adjs.shape[-1]
    mine = torch.cat(mine) if mine else torch.zeros(0, N, N, dtype=torch.uint8, device=dev)
    allg = gather_batches(mine, n_batches, B, rank, world, dist)[:n_graphs]
    gen_graph_list = adjs_to_graphs(allg.cpu().numpy().astype(np.float32))
    result_dict = eval_graph_list(ref_graphs, gen_graph_list, methods=methods)
    return result_dict, gen_graph_list


def _load(config, modellink):
    models = prepare_test_model_train(config, modellink)
    if not models:
        raise FileNotFoundError(f"no readable checkpoint under {modellink}")
    file, sigma_list, model_params = models[0]
    return load_model(*model_params)


def sample_main(config, modellink, epoch, noise_num, chain=None):
    """reference sample_ppgn_simple.py:17-102."""
    train_graph_list, test_graph_list = load_data(config, get_graph_list=True)
    model = _load(config, modellink)
    pool_adjs, _ = graphs_to_tensor(config, train_graph_list)
    result_dict, _ = run_sample(config, model, pool_adjs, config.samplesize, noise_num, chain or engine.sample_chain_simple,
                                test_graph_list)
    logging.info(f"sample_main epoch {epoch} levels {noise_num}: {result_dict}")
    return result_dict


def sample_testing(config, modellink, epoch, noise_num, train_dl, chain=None):
    """reference sample_ppgn_simple.py:106-187: as sample_main but node counts and the MMD reference come from the
    training loader, and 256 graphs are drawn (:186)."""
    pool = train_dl.pool() if hasattr(train_dl, "pool") else torch.cat([adj for adj, _ in train_dl])
    model = _load(config, modellink)
    train_graphs = adjs_to_graphs(pool.cpu().numpy())
    result_dict, _ = run_sample(config, model, pool, 256, noise_num, chain or engine.sample_chain_simple, train_graphs)
    return result_dict
