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
