"""GPU graph statistics (csrc/dppm_stats.hip) against the CPU oracle: histograms bit-exact, MMDs to double rounding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _adjs(rng, G, N, p, ragged=True):
    out = np.zeros((G, N, N), np.float32)
    for g in range(G):
        n = int(rng.integers(1, N + 1)) if ragged else N
        up = np.triu(rng.random((n, n)) < p, 1).astype(np.float32)
        out[g, :n, :n] = up + up.T
    return out


@pytest.mark.parametrize("N,p", [(1, 0.5), (7, 0.5), (33, 0.3), (64, 0.1), (200, 0.05), (257, 0.02)])
def test_histograms_bit_exact(N, p):
    from discrete_dppm_graphs_amd import ops
    from oracle import stats as os_
    rng = np.random.default_rng(N)
    adjs = _adjs(rng, 9, N, p)
    adjs[0] = 0                      # edgeless graph -> one isolated node
    if N > 2:
        adjs[1, 1, 1] = 1.0          # self loop is ignored
    deg, clus = ops.graph_stats(torch.from_numpy(adjs).cuda(), bins=100)
    deg, clus = deg.cpu().numpy(), clus.cpu().numpy()
    for g in range(len(adjs)):
        want = os_.degree_histogram(adjs[g])
        assert np.array_equal(deg[g, :len(want)], want) and not deg[g, len(want):].any()
        assert np.array_equal(clus[g], os_.clustering_histogram(adjs[g]))


def test_clustering_bins_on_numpy_edges():
    """Coefficients that are exact multiples of 1/100 (7/100 != 7 * 0.01 in float64) must land where numpy.histogram puts them:
    wheels whose hub has degree d give the hub c = 2 / (d - 1), complete graphs give 1.0 (closed last bin)."""
    from discrete_dppm_graphs_amd import ops
    from oracle import stats as os_
    mats = []
    for d in (3, 5, 11, 21, 26, 41, 51):
        n = d + 1
        a = np.zeros((64, 64), np.float32)
        a[0, 1:n] = a[1:n, 0] = 1
        for i in range(1, n):
            j = i + 1 if i + 1 < n else 1
            a[i, j] = a[j, i] = 1
        mats.append(a)
    k = np.zeros((64, 64), np.float32)
    k[:9, :9] = 1 - np.eye(9)
    mats.append(k)
    adjs = np.stack(mats)
    _, clus = ops.graph_stats(torch.from_numpy(adjs).cuda(), bins=100)
    for g in range(len(adjs)):
        assert np.array_equal(clus[g].cpu().numpy(), os_.clustering_histogram(adjs[g]))


def test_mmd_matches_oracle_and_eval_hooks():
    from discrete_dppm_graphs_amd.evaluation import stats as ev
    from oracle import stats as os_
    rng = np.random.default_rng(3)
    ref = _adjs(rng, 17, 40, 0.2)
    pred = _adjs(rng, 11, 40, 0.35)
    want = os_.eval_adjs(ref, pred)
    got = ev.eval_torch_batch(torch.from_numpy(ref).cuda(), torch.from_numpy(pred).cuda())
    for k in ("degree", "cluster"):
        assert abs(got[k] - want[k]) <= 1e-12 + 1e-10 * abs(want[k]), (k, got[k], want[k])
    same = ev.eval_torch_batch(torch.from_numpy(ref).cuda(), torch.from_numpy(ref).cuda())
    assert abs(same["degree"]) < 1e-12 and abs(same["cluster"]) < 1e-12
    # the list-of-graphs hook the samplers call (networkx graphs with isolated nodes removed): same numbers
    lst = ev.eval_graph_list(ev.adjs_to_graphs(ref), ev.adjs_to_graphs(pred))
    for k in ("degree", "cluster"):
        assert abs(lst[k] - want[k]) <= 1e-12 + 1e-10 * abs(want[k])


def test_orbit_mmd_on_device_matches_oracle():
    from discrete_dppm_graphs_amd.evaluation import stats as ev
    from oracle import stats as os_
    rng = np.random.default_rng(5)
    ref = _adjs(rng, 6, 14, 0.4)
    pred = _adjs(rng, 5, 14, 0.6)
    want = os_.orbit_mmd(list(ref), list(pred))
    got = ev.eval_torch_batch(torch.from_numpy(ref).cuda(), torch.from_numpy(pred).cuda(), methods=["orbit"])
    assert abs(got["orbit"] - want) <= 1e-12
    lst = ev.eval_graph_list(ev.adjs_to_graphs(ref), ev.adjs_to_graphs(pred))
    assert set(lst) == {"degree", "cluster", "orbit"} and abs(lst["orbit"] - want) <= 1e-12
