"""CPU restatement of the graph statistics the samplers use for model selection (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference evaluation/stats.py:276-286 (adjs_to_graphs: self loops and isolated nodes dropped, an edgeless graph
becomes one isolated node), :41-69 (degree_stats: nx.degree_histogram), :140-186 (clustering_stats: nx.clustering +
np.histogram(bins=100, range=(0, 1)), kernel sigma = 1/10, distance_scaling = bins) and evaluation/mmd.py:14-41, :90-106
(Gaussian-EMD kernel, compute_mmd).  The reference obtains the EMD from pyemd 0.5.1 (environment.yml:195) on the
toeplitz(|i-j|) distance matrix; pyemd is absent from this image, so the published closed form for one-dimensional
histograms is restated (EMD = sum_k |CDF_x(k) - CDF_y(k)|) and pinned against scipy.stats.wasserstein_distance and, for
the histograms, against networkx itself (tests/test_oracle_golden.py).  Parity with pyemd's own output: unpinned.
"""
import numpy as np


def degree_histogram(adj):
    a = np.asarray(adj) > 0.5
    a = a & ~np.eye(a.shape[0], dtype=bool)
    deg = a.sum(-1)
    deg = deg[deg > 0]
    return np.bincount(deg).astype(np.float64) if deg.size else np.ones(1)


def clustering_histogram(adj, bins=100):
    a = (np.asarray(adj) > 0.5) & ~np.eye(len(adj), dtype=bool)
    ai = a.astype(np.int64)
    deg = ai.sum(-1)
    t2 = np.einsum("ij,jk,ki->i", ai, ai, ai)            # 2 x triangles through i (networkx's t)
    keep = deg > 0
    d, t = deg[keep].astype(np.float64), t2[keep].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.where(t == 0, 0.0, t / (d * (d - 1)))
    if c.size == 0:
        c = np.zeros(1)
    hist, _ = np.histogram(c, bins=bins, range=(0.0, 1.0), density=False)
    return hist.astype(np.float64)


def emd_1d(x, y, distance_scaling=1.0):
    n = max(len(x), len(y))
    x = np.pad(np.asarray(x, np.float64), (0, n - len(x)))
    y = np.pad(np.asarray(y, np.float64), (0, n - len(y)))
    return np.abs(np.cumsum(x) - np.cumsum(y)).sum() / distance_scaling


def gaussian_emd(x, y, sigma=1.0, distance_scaling=1.0):
    e = emd_1d(x, y, distance_scaling)
    return np.exp(-e * e / (2 * sigma * sigma))


def compute_mmd(s1, s2, sigma=1.0, distance_scaling=1.0):
    s1 = [np.asarray(s, np.float64) / np.sum(s) for s in s1]
    s2 = [np.asarray(s, np.float64) / np.sum(s) for s in s2]
    k = lambda p, q: sum(gaussian_emd(a, b, sigma, distance_scaling) for a in p for b in q) / (len(p) * len(q))
    return k(s1, s1) + k(s2, s2) - 2 * k(s1, s2)


def eval_adjs(ref, pred, bins=100):
    """{'degree', 'cluster'} MMDs between two batches of dense adjacencies, as eval_graph_list computes them."""
    return {"degree": compute_mmd([degree_histogram(a) for a in ref], [degree_histogram(a) for a in pred]),
            "cluster": compute_mmd([clustering_histogram(a, bins) for a in ref], [clustering_histogram(a, bins) for a in pred],
                                   sigma=1.0 / 10, distance_scaling=bins)}


# ---------------------------------------------------------------------------------------------------------
# Orbit statistic (reference evaluation/stats.py:210-273).  The reference shells out to ORCA ("orca node 4": per-node counts of
# the 15 orbits of the connected 2-, 3- and 4-node graphlets, Przulj's numbering) and keeps only the per-graph column sums divided
# by the number of (non-isolated) nodes.  The restatement below is the DEFINITION of those counts (enumeration of every node subset).
# Parity: PINNED by tests/golden/g7_orbits.npz, the output of the reference's own ORCA SOURCE (evaluation/orca/orca.cpp) compiled
# here by oracle/Makefile into oracle/_ref/orca (tests/golden/gen_golden_orbits.py); the prebuilt binary in the reference is never run.
# ---------------------------------------------------------------------------------------------------------
def orbit_totals_bruteforce(adj):
    """[15] sums over nodes of the orbit counts of a dense 0/1 adjacency (python loops: small graphs only)."""
    from itertools import combinations
    a = (np.asarray(adj) > 0.5)
    a = a & ~np.eye(a.shape[0], dtype=bool)
    n = a.shape[0]
    out = np.zeros(15, np.int64)
    out[0] = a.sum()                                   # every edge puts both its ends on orbit 0
    for s in combinations(range(n), 3):
        sub = a[np.ix_(s, s)]
        deg = sub.sum(1)
        e = deg.sum() // 2
        if e == 2:                                     # induced path: ends on orbit 1, middle on orbit 2
            out[1] += 2; out[2] += 1
        elif e == 3:
            out[3] += 3
    for s in combinations(range(n), 4):
        sub = a[np.ix_(s, s)]
        deg = np.sort(sub.sum(1))
        e = deg.sum() // 2
        key = tuple(deg)
        if e == 3 and key == (1, 1, 2, 2):             # path: ends 4, middle 5
            out[4] += 2; out[5] += 2
        elif e == 3 and key == (1, 1, 1, 3):           # claw: leaves 6, centre 7
            out[6] += 3; out[7] += 1
        elif e == 4 and key == (2, 2, 2, 2):           # cycle
            out[8] += 4
        elif e == 4 and key == (1, 2, 2, 3):           # tailed triangle: tail end 9, far triangle nodes 10, junction 11
            out[9] += 1; out[10] += 2; out[11] += 1
        elif e == 5:                                   # diamond: degree-2 nodes 12, degree-3 nodes 13
            out[12] += 2; out[13] += 2
        elif e == 6:
            out[14] += 4
    return out


def orbit_mmd(ref_adjs, pred_adjs, totals=orbit_totals_bruteforce, sigma=30.0):
    """evaluation/stats.py:239-273: per-graph totals / #non-isolated nodes, Gaussian kernel (util/dist_helper.py:91-96) on the raw
    15-vectors (is_hist=False), MMD as compute_mmd."""
    def feats(adjs):
        rows = []
        for a in adjs:
            a = np.asarray(a)
            nn = max(int(((a > 0.5).sum(-1) > 0).sum()), 1)
            rows.append(totals(a).astype(np.float64) / nn)
        return np.asarray(rows)
    x, y = feats(ref_adjs), feats(pred_adjs)

    def k(p, q):
        d2 = ((p[:, None, :] - q[None, :, :]) ** 2).sum(-1)
        return np.exp(-d2 / (2 * sigma * sigma)).mean()
    return float(k(x, x) + k(y, y) - 2 * k(x, y))
