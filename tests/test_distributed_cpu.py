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


def _epoch_fn(rank, world):
    """ppgn_simple._epoch_loop under two ranks with a stub step: 9 graphs in batches of 4 -> global batches of 4, 4, 1; the tail
    batch has fewer graphs than ranks, so one rank's slice is empty and must still join the all-reduce and the optimiser step."""
    import types
    from discrete_dppm_graphs_amd import engine, ppgn_simple as ps
    from discrete_dppm_graphs_amd.utils.resident import STREAM_NOISE, ResidentLoader, host_rng, upload

    class Fake(engine.FlatState):
        def __init__(self):
            self.grad = torch.zeros(3)
            self.steps = 0
            self.sums = []

        def adam_step(self, lr=None):
            self.steps += 1
            self.sums.append(float(self.grad[0]))

    G, N, bs = 9, 4, 4
    adjs = torch.arange(G, dtype=torch.float32).view(G, 1, 1).expand(G, N, N).contiguous()      # graph g is filled with g
    dl = ResidentLoader(adjs, torch.zeros(G, N, 1), bs, shuffle=True, seed=5, device="cpu")
    order = dl.order(0).tolist()
    config = types.SimpleNamespace(dev=torch.device("cpu"))
    st = Fake()
    seen = []

    def plan(n, pass_no, dev):
        return upload(host_rng(5, STREAM_NOISE, pass_no).uniform(0.0, 0.5, size=n), dev, torch.float64)

    def step_fn(model, state, adj, draws, Bg, gid0, offset, train):
        seen.append((adj[:, 0, 0].tolist(), draws.tolist(), Bg, gid0, offset))
        if adj.shape[0] == 0:
            return engine.empty_step(state, adj, torch.float64)
        state.grad.copy_(torch.full((3,), float(adj.shape[0])))
        state.all_reduce()
        state.adam_step()
        return draws.sum()

    mean = ps._epoch_loop(None, st, dl, config, step_fn, True, 0, {"plan": plan})
    want_draws = host_rng(5, STREAM_NOISE, 0).uniform(0.0, 0.5, size=G)
    return {"seen": seen, "steps": st.steps, "sums": st.sums, "mean": mean, "order": order, "want": want_draws.tolist()}


def test_epoch_loop_two_ranks_same_draws_and_no_hang_on_empty_shard():
    outs = _run(_epoch_fn)
    a, b = outs
    assert a["order"] == b["order"] and a["want"] == b["want"]            # same shuffle, same noise table on every rank
    assert a["steps"] == b["steps"] == 3                                   # the rank with the empty slice stepped too
    assert a["sums"] == b["sums"] == [4.0, 4.0, 1.0]                       # all-reduced "gradient" = size of the GLOBAL batch
    assert abs(a["mean"] - b["mean"]) < 1e-12
    assert abs(a["mean"] - sum(a["want"]) / 3) < 1e-12                     # per-batch losses summed over ranks, averaged over batches
    first = 0
    for s in range(3):
        ga, da, Bg, gid_a, off_a = a["seen"][s]
        gb, db, Bg_b, gid_b, off_b = b["seen"][s]
        assert Bg == Bg_b and off_a == off_b == s                           # pass 0, step s
        assert ga + gb == [float(g) for g in a["order"][first:first + Bg]]  # the two slices tile the global batch in order
        assert da + db == a["want"][first:first + Bg]                       # each graph meets the draw of its global position
        assert gid_a == first and gid_b == first + len(ga)
        first += Bg
    assert len(b["seen"][2][0]) == 0 or len(a["seen"][2][0]) == 0           # the tail batch left one rank empty


def test_resident_loader_covers_dataset_and_is_keyed_by_epoch():
    sys.path.insert(0, ROOT)
    from discrete_dppm_graphs_amd.utils.resident import ResidentLoader, batch_sizes_of, upload
    G = 11
    adjs = torch.arange(G, dtype=torch.float32).view(G, 1, 1).expand(G, 3, 3).contiguous()
    dl = ResidentLoader(adjs, torch.zeros(G, 3, 1), 4, shuffle=True, seed=1, device="cpu")
    assert len(dl) == 3 and batch_sizes_of(dl) == [4, 4, 3]
    n0 = upload.calls
    e0 = [a[:, 0, 0].tolist() for a, _ in dl]
    assert upload.calls == n0 + 1                                          # one host-to-device copy per pass (the permutation)
    e1 = [a[:, 0, 0].tolist() for a, _ in dl]
    assert sorted(sum(e0, [])) == sorted(sum(e1, [])) == [float(g) for g in range(G)] and e0 != e1
    dl2 = ResidentLoader(adjs, torch.zeros(G, 3, 1), 4, shuffle=True, seed=1, device="cpu")
    dl2.epoch = 1
    assert [a[:, 0, 0].tolist() for a, _ in dl2] == e1                     # the order depends on (seed, epoch) only


def test_lr_schedule_matches_torch_exponential_lr_stepped_with_epoch():
    """The reference builds ExponentialLR and calls scheduler.step(epoch) after every epoch (ppgn_simple.py:56,95)."""
    import warnings
    sys.path.insert(0, ROOT)
    from discrete_dppm_graphs_amd.ppgn_simple import lr_after_epoch
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=0.01)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)
    assert lr_after_epoch(0.01, 0.5, -1) == 0.01
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for epoch in range(5):
            opt.step()
            sched.step(epoch)
            assert abs(opt.param_groups[0]["lr"] - lr_after_epoch(0.01, 0.5, epoch)) < 1e-15


def test_bench_self_launch_starts_ranks_and_relays_one_line():
    """bench.py --gpus 2 without a torch.distributed environment: the parent starts two ranks (here: gloo rendezvous only, no
    GPU call) and prints rank 0's JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] == 3.0                      # 1 + 2: both ranks joined the all-reduce
