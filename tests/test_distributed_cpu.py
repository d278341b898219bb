"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: graph sharding, the one gradient all-reduce of training and
the one final all-gather of sampling.  The kernels themselves are not involved (they need a GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fn, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return [ret[r] for r in range(world)]


def _gather_fn(rank, world):
    from discrete_dppm_graphs_amd.sample_ppgn_simple import gather_batches, my_batches
    B, N, n_batches = 3, 4, 5
    mine = []
    for bi in my_batches(n_batches, rank, world):
        mine.append(torch.full((B, N, N), bi, dtype=torch.uint8) + torch.arange(B, dtype=torch.uint8).view(B, 1, 1) * 10)
    mine = torch.cat(mine)
    out = gather_batches(mine, n_batches, B, rank, world, dist)
    return out.numpy()


def test_sampling_shards_and_final_gather_order():
    outs = _run(_gather_fn)
    B, n_batches = 3, 5
    want = np.concatenate([np.full((B, 4, 4), bi, np.uint8) + (np.arange(B, dtype=np.uint8) * 10).reshape(B, 1, 1)
                           for bi in range(n_batches)])
    for o in outs:
        assert np.array_equal(o, want)      # identical on every rank, in global batch order, independent of world size


def _allreduce_fn(rank, world):
    from discrete_dppm_graphs_amd import engine
    from discrete_dppm_graphs_amd.ppgn_simple import shard_bounds

    class Fake(engine.FlatState):
        def __init__(self, g):
            self.grad = g
    Bg = 7
    per_graph = torch.arange(Bg, dtype=torch.float32).view(Bg, 1) * torch.ones(1, 5)
    lo, hi = shard_bounds(Bg, rank, world)
    st = Fake(per_graph[lo:hi].sum(0))      # this rank's gradient contribution = sum over its graphs
    st.all_reduce()
    return st.grad.numpy(), (lo, hi)


def test_training_shards_cover_batch_and_gradients_sum():
    outs = _run(_allreduce_fn)
    spans = sorted(o[1] for o in outs)
    assert spans[0][0] == 0 and spans[-1][1] == 7 and spans[0][1] == spans[1][0]
    for g, _ in outs:
        assert np.allclose(g, np.arange(7).sum())


def test_shard_bounds_properties():
    sys.path.insert(0, ROOT)
    from discrete_dppm_graphs_amd.ppgn_simple import shard_bounds
    for Bg in (1, 5, 32, 64):
        for world in (1, 2, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_bounds(Bg, r, world)
                cover += list(range(lo, hi))
            assert cover == list(range(Bg))
