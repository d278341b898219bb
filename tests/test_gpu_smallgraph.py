"""Whole-graph-per-workgroup PPGN forward (csrc/dppm_smallgraph.hip: N <= 20, hidden 32, inference, one launch) through
dppm_ppgn_forward, against the oracle and against the general multi-launch path (DPPM_DISABLE_SMALLGRAPH=1) on the same inputs."""
import os

import numpy as np
import pytest
import torch

from test_gpu_ppgn import close, dev, rand_case, run_gpu

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    from discrete_dppm_graphs_amd import ops
    return ops


def _general(ops, *args):
    with ops.debug_switch(DISABLE_SMALLGRAPH=1):
        return run_gpu(ops, *args)[0]


@pytest.mark.parametrize("B,N,L", [(1, 1, 1), (2, 2, 2), (3, 4, 1), (2, 5, 3), (4, 12, 2), (3, 16, 6), (5, 17, 2), (32, 18, 6), (32, 20, 6),
                                   (2, 20, 16), (70, 19, 1)])
def test_fused_forward_vs_oracle_and_general_path(ops, B, N, L):
    from oracle import ppgn as op
    rng = np.random.default_rng(100 * N + L)
    A, mask = rand_case(rng, B, N, ragged=N >= 6)
    if N < 6:                                        # tiny graphs: every node active, arbitrary edges
        up = np.triu(rng.random((B, N, N)) < 0.5, 1).astype(np.float32)
        A = up + up.transpose(0, 2, 1)
        mask = np.ones((B, N, N), np.float32)
    sigma = rng.uniform(0, 0.5, size=B)
    W = op.init_weights(L, 32, seed=N)
    with torch.no_grad():
        ref = op.powerful_forward(W, A, mask, sigma, L).squeeze(-1).numpy()
    got, _ = run_gpu(ops, W, A, mask, sigma, L, 32)
    close(got, ref)                                  # padded positions carry the reference's unmasked values too
    close(got, _general(ops, W, A, mask, sigma, L, 32), tol=5e-5)


@pytest.mark.parametrize("B,N,L", [(2, 5, 3), (3, 16, 6), (32, 18, 6), (32, 20, 6), (2, 20, 16)])
def test_fused_forward_split_arithmetic(ops, B, N, L):
    """precision bf16x3 (the samplers' default): the 32-wide products of the one-launch forward run as three bf16 MFMAs on split operands.
    North-star bar 1e-3 of max |logit| against the oracle (the measured distance is printed); bit-stable; independent of the batch;
    the golden logits of the reference at ego_small's geometry hold to the same bar."""
    from oracle import ppgn as op
    rng = np.random.default_rng(300 * N + L)
    A, mask = rand_case(rng, B, N, ragged=N >= 6)
    if N < 6:
        up = np.triu(rng.random((B, N, N)) < 0.5, 1).astype(np.float32)
        A = up + up.transpose(0, 2, 1)
        mask = np.ones((B, N, N), np.float32)
    sigma = rng.uniform(0, 0.5, size=B)
    W = op.init_weights(L, 32, seed=N + 1)
    with torch.no_grad():
        ref = op.powerful_forward(W, A, mask, sigma, L).squeeze(-1).numpy()
    got, _ = run_gpu(ops, W, A, mask, sigma, L, 32, precision="bf16x3")
    exact, _ = run_gpu(ops, W, A, mask, sigma, L, 32)
    scale = np.abs(ref).max()
    print(f"\none-launch forward N={N} L={L}: bf16x3 {np.abs(got - ref).max() / scale:.2e}, f32 {np.abs(exact - ref).max() / scale:.2e} of max |logit|")
    close(got, ref, tol=1e-3)
    assert not np.array_equal(got, exact)
    again, _ = run_gpu(ops, W, A, mask, sigma, L, 32, precision="bf16x3")
    assert np.array_equal(got, again)
    one, _ = run_gpu(ops, W, A[-1:], mask[-1:], sigma[-1:], L, 32, precision="bf16x3")
    assert np.array_equal(got[-1:], one)
    with ops.debug_switch(DISABLE_MLP32S=1):
        off, _ = run_gpu(ops, W, A, mask, sigma, L, 32, precision="bf16x3")
    assert np.array_equal(off, exact)


def test_fused_forward_golden_reference_logits_split(ops):
    g = np.load(os.path.join(G, "g1_ppgn_n18_h32_l6.npz"))
    from discrete_dppm_graphs_amd.ops import used_param_names
    W = {n: g["w::" + n] for n in used_param_names(6)}
    lb, _ = run_gpu(ops, W, g["A_t"], g["mask"], float(g["sigmas"][0]), 6, 32, precision="bf16x3")
    close(lb, g["logits_batched_sigma0"], tol=1e-3)


def test_fused_forward_golden_reference_logits(ops):
    """The reference's own logits for ego_small's geometry (N = 18, 6 x 32), one sigma for the whole batch."""
    g = np.load(os.path.join(G, "g1_ppgn_n18_h32_l6.npz"))
    from discrete_dppm_graphs_amd.ops import used_param_names
    W = {n: g["w::" + n] for n in used_param_names(6)}
    lb, _ = run_gpu(ops, W, g["A_t"], g["mask"], float(g["sigmas"][0]), 6, 32)
    close(lb, g["logits_batched_sigma0"])


def test_fused_forward_is_deterministic_and_batch_independent(ops):
    from oracle import ppgn as op
    rng = np.random.default_rng(7)
    A, mask = rand_case(rng, 9, 20)
    sigma = rng.uniform(0, 0.5, size=9)
    W = op.init_weights(6, 32, seed=3)
    a, _ = run_gpu(ops, W, A, mask, sigma, 6, 32)
    b, _ = run_gpu(ops, W, A, mask, sigma, 6, 32)
    assert np.array_equal(a, b)
    c, _ = run_gpu(ops, W, A[4:5], mask[4:5], sigma[4:5], 6, 32)
    assert np.array_equal(a[4:5], c)                 # one workgroup per graph: no dependence on the rest of the batch


def test_run_sample_carries_several_batches_per_chain_call(monkeypatch):
    """Small graphs: run_sample merges consecutive reference batches into one chain call (engine.chain_width); the generated
    graphs are exactly those of the batch-by-batch loop, ragged tail included."""
    import types
    from discrete_dppm_graphs_amd import engine, sample_ppgn_simple as sps
    from discrete_dppm_graphs_amd.model.ppgn import Powerful
    torch.manual_seed(5)
    model = Powerful(num_layers=3, hidden=32, hidden_final=32, n_nodes=18, normalization="instance", adj_out=True, input_features=2,
                     dropout_p=1e-6).cuda().eval()
    rng = np.random.default_rng(11)
    pool, _ = rand_case(rng, 12, 18)
    pool = torch.from_numpy(pool)
    cfg = types.SimpleNamespace(test=types.SimpleNamespace(batch_size=4), dev="cuda", seed=77)
    calls = []

    def chain(model, flags, T, seed, graph_id0=0):
        calls.append(flags.shape[0])
        return engine.sample_chain_simple(model, flags, T, seed, graph_id0=graph_id0)

    got = {}
    monkeypatch.setattr(sps, "eval_graph_list", lambda ref, gen, methods=None: {})
    monkeypatch.setattr(sps, "adjs_to_graphs", lambda a: a)
    for width in ("merged", "single"):
        if width == "single":
            monkeypatch.setattr(engine, "chain_width", lambda *a, **k: 1)
        np.random.seed(3)
        sps._CALLS["n"] = 0
        calls.clear()
        _, gen = sps.run_sample(cfg, model, pool, 22, 6, chain, None)
        got[width] = (np.asarray(gen).copy(), list(calls))
    assert got["merged"][1] == [24] and got["single"][1] == [4] * 6
    assert got["merged"][0].shape[0] == 24 - 2 or got["merged"][0].shape[0] == 24      # 22 asked: 6 batches of 4 drawn, as the reference does
    assert np.array_equal(got["merged"][0][:22], got["single"][0][:22])
    assert got["merged"][0][:22].sum() > 0


@pytest.mark.parametrize("N", [13, 17, 18, 19])
def test_fused_forward_ignores_stale_lds(ops, N):
    """N % 4 != 0: rows N .. ld-1 of the o2 planes are LDS words the kernel never writes.  With every CU's LDS poisoned with a quiet
    NaN (then +Inf) bit pattern beforehand, the logits must still equal the general path's: no uninitialised word may reach a
    product (0 * NaN would)."""
    import ctypes as C
    from discrete_dppm_graphs_amd._lib import check, lib
    from oracle import ppgn as op
    rng = np.random.default_rng(300 + N)
    A, mask = rand_case(rng, 40, N, ragged=True)
    sigma = rng.uniform(0, 0.5, size=40)
    W = op.init_weights(3, 32, seed=N)
    want = _general(ops, W, A, mask, sigma, 3, 32)
    for pattern in (0x7FC00000, 0x7F800000):
        check(lib.dppm_test_fill_lds(pattern, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dppm_test_fill_lds")
        got, _ = run_gpu(ops, W, A, mask, sigma, 3, 32)
        assert np.isfinite(got).all()
        close(got, want, tol=5e-5)


@pytest.mark.parametrize("kind", ["simple", "vlb"])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("N,L,T", [(20, 6, 16), (13, 3, 7), (5, 2, 4), (1, 1, 3), (16, 4, 5)])
def test_chain_in_one_launch_is_the_level_by_level_chain(N, L, T, precision, kind):
    """dppm_ppgn_chain_simple / _vlb (x_T, then every level inside one kernel) against the same chain run through dppm_init_xT,
    dppm_edgeflip_fwd, dppm_ppgn_forward and dppm_reverse_step_simple / _vlb: the same Philox counters and the same float
    expressions, so the sampled adjacency is compared for EQUALITY; ragged graphs (inactive nodes) included, and the LDS is
    poisoned with NaN patterns first (the adjacency and logits planes of the chain live there)."""
    import ctypes as C
    from discrete_dppm_graphs_amd import engine
    from discrete_dppm_graphs_amd._lib import check, lib
    from discrete_dppm_graphs_amd.model.ppgn import Powerful
    torch.manual_seed(100 + N)
    model = Powerful(num_layers=L, hidden=32, hidden_final=32, n_nodes=N, normalization="instance", adj_out=True, input_features=2,
                     dropout_p=1e-6, config={"precision": precision}).cuda().eval()
    assert model.precision == precision
    B = 37
    rng = np.random.default_rng(N)
    flags = np.zeros((B, N), np.float32)
    for b in range(B):
        flags[b, :rng.integers(1, N + 1)] = 1.0
    flags = torch.from_numpy(flags).cuda()
    assert model.chain_supported(N)
    chain = engine.sample_chain_simple if kind == "simple" else engine.sample_chain_vlb
    check(lib.dppm_test_fill_lds(0x7FC00000, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dppm_test_fill_lds")
    one = chain(model, flags, T, seed=9, graph_id0=5)
    steps = []
    ref = chain(model, flags, T, seed=9, graph_id0=5, record=lambda c, a_t, logits, x: steps.append(c))
    assert steps == list(range(T))                     # the recording form went level by level
    assert torch.equal(one, ref)
    assert torch.equal(one, one.transpose(1, 2)) and float(one.diagonal(dim1=1, dim2=2).abs().sum()) == 0.0
    act = flags[:, :, None] * flags[:, None, :]
    assert float((one * (1 - act)).abs().sum()) == 0.0   # nothing outside the active block
    if N >= 13 and kind == "simple":
        assert 0 < float(one.sum()) < float(act.sum())


def test_chain_entry_points_refuse_what_they_cannot_run():
    from discrete_dppm_graphs_amd import engine, ops
    from discrete_dppm_graphs_amd.model.ppgn import Powerful
    model = Powerful(num_layers=2, hidden=32, hidden_final=32, n_nodes=24, normalization="instance", adj_out=True, input_features=2,
                     dropout_p=1e-6).cuda().eval()
    flags = torch.ones(3, 24, device="cuda")
    assert not model.chain_supported(24)
    tab = engine.level_table_simple(engine.schedule_simple(4), flags.device)
    with pytest.raises(RuntimeError, match="level by level"):
        ops.chain_simple(model._pack(), flags, ops.generate_mask(flags), tab, seed=1)
    with pytest.raises(RuntimeError, match="level by level"):
        ops.chain_vlb(model._pack(), flags, ops.generate_mask(flags), engine.level_table_vlb(engine.schedule_vlb(4), flags.device), seed=1)
    with pytest.raises(ValueError, match="level_table"):
        ops.chain_vlb(model._pack(), flags, ops.generate_mask(flags), tab, seed=1)
    for chain in (engine.sample_chain_simple, engine.sample_chain_vlb):     # the engine falls back by itself
        assert chain(model, flags, 4, seed=1).shape == (3, 24, 24)
