"""Boundary to the reference's evaluation side branch (evaluation/stats.py).  What the samplers call is provided:
`adjs_to_graphs`, `eval_torch_batch` and an `eval_graph_list` hook.

On a GPU the default evaluator computes the reference's `degree` and `cluster` MMDs (degree / clustering-coefficient
histograms, Gaussian-EMD kernel, evaluation/stats.py:41-69, :140-186, evaluation/mmd.py:32-41, :90-106) with the HIP
kernels of csrc/dppm_stats.hip, so periodic model selection does not stall training on
networkx / pyemd loops.  The `orbit` MMD (evaluation/stats.py:210-273, an ORCA subprocess in the reference) is computed from
closed forms over dense matrix products (`orbit_totals`, torch on the device); the SBM likelihood (graph_tool) stays out of
scope; register the reference's own evaluator (or any other) with `set_evaluator(fn)` to get it.  Without a GPU the default is a
numpy degree-histogram MMD (Gaussian-TV kernel)."""
import numpy as np

_EVALUATOR = None


def set_evaluator(fn):
    """fn(graph_ref_list, graph_pred_list, methods=None) -> dict of metric name -> float (e.g. the reference's
    evaluation.stats.eval_graph_list)."""
    global _EVALUATOR
    _EVALUATOR = fn


def adjs_to_graphs(adjs, node_flags=None):
    """reference evaluation/stats.py:276-286 (networkx 3: from_numpy_array)."""
    import networkx as nx
    graph_list = []
    for adj in adjs:
        G = nx.from_numpy_array(np.asarray(adj))
        G.remove_edges_from(nx.selfloop_edges(G))
        G.remove_nodes_from(list(nx.isolates(G)))
        if G.number_of_nodes() < 1:
            G.add_node(1)
        graph_list.append(G)
    return graph_list


def _degree_hist(g):
    import networkx as nx
    if isinstance(g, np.ndarray):
        deg = (g > 0.5).sum(-1).astype(int)
        deg = deg[deg > 0]
        return np.bincount(deg) if deg.size else np.zeros(1)
    return np.asarray(nx.degree_histogram(g), dtype=np.float64)


def _gaussian_tv(x, y, sigma=1.0):
    n = max(len(x), len(y))
    x = np.pad(x, (0, n - len(x))) / max(x.sum(), 1e-12)
    y = np.pad(y, (0, n - len(y))) / max(y.sum(), 1e-12)
    d = np.abs(x - y).sum() / 2.0
    return np.exp(-d * d / (2 * sigma * sigma))


def degree_mmd(ref, pred):
    a = [_degree_hist(g) for g in ref]
    b = [_degree_hist(g) for g in pred]
    k = lambda p, q: np.mean([_gaussian_tv(x, y) for x in p for y in q])
    return float(k(a, a) + k(b, b) - 2 * k(a, b))


def orbit_totals(adj, chunk=64):
    """[G, 15] float64: per graph, the sums over nodes of the 15 orbit counts of the connected 2-, 3- and 4-node graphlets
    (what `orca node 4` prints per node, summed; Przulj's orbit numbering).  adj: [G, N, N] 0/1 tensor (any device).
    Non-induced subgraph counts from degrees, A^2, A^4's trace and per-edge common-neighbour counts, K4 from the triangles inside
    every neighbourhood, then inclusion-exclusion to induced counts.  All arithmetic in float64 on integers (exact below 2^53)."""
    import torch
    A = (adj > 0.5).to(torch.float64)
    G, N, _ = A.shape
    A = A * (1.0 - torch.eye(N, dtype=torch.float64, device=A.device))
    d = A.sum(-1)                                             # [G, N]
    m = d.sum(-1) / 2
    A2 = A @ A
    te = A2 * A                                               # per ordered edge: common neighbours
    tv = te.sum(-1) / 2                                       # triangles through each node
    T = tv.sum(-1) / 3
    W = (d * (d - 1) / 2).sum(-1)                             # wedges
    S = (d * (d - 1) * (d - 2) / 6).sum(-1)                   # 3-stars
    P = (A * ((d - 1)[:, :, None] * (d - 1)[:, None, :])).sum((-1, -2)) / 2 - 3 * T      # 3-edge paths
    C = ((A2 * A2).sum((-1, -2)) - 4 * W - 2 * m) / 8         # 4-cycles (trace(A^4) = sum of squares of A^2)
    TT = (tv * (d - 2)).sum(-1)                               # triangle + pendant edge
    D = (te * (te - 1) / 2).sum((-1, -2)) / 2                 # two triangles sharing an edge
    K = torch.zeros(G, dtype=torch.float64, device=A.device)  # K4 = (1/4) sum_i triangles inside N(i)
    Af = A.to(torch.float32)                                  # entries of the masked squares are <= N: exact in float32
    for g in range(G):
        acc = torch.zeros((), dtype=torch.float64, device=A.device)
        for i0 in range(0, N, chunk):
            nb = Af[g, i0:i0 + chunk]                         # [c, N] neighbourhood indicators
            M = Af[g][None] * nb[:, :, None] * nb[:, None, :]   # [c, N, N] adjacency inside each neighbourhood
            acc += ((M @ M) * M).to(torch.float64).sum() / 6
        K[g] = acc / 4
    diam = D - 6 * K
    c4 = C - D + 3 * K
    paw = TT - 4 * diam - 12 * K
    claw = S - paw - 2 * diam - 4 * K
    p4 = P - 4 * c4 - 2 * paw - 6 * diam - 12 * K
    p3 = W - 3 * T
    return torch.stack([2 * m, 2 * p3, p3, 3 * T, 2 * p4, 2 * p4, 3 * claw, claw, 4 * c4, paw, 2 * paw, paw, 2 * diam, 2 * diam, 4 * K], dim=-1)


def orbit_mmd_batch(ref_batch, pred_batch, sigma=30.0):
    """reference orbit_stats_all (evaluation/stats.py:239-273) on dense batches: totals / #non-isolated nodes, Gaussian kernel on
    the raw 15-vectors (util/dist_helper.py:91-96, is_hist=False), compute_mmd."""
    import torch

    def feats(b):
        nn = ((b > 0.5).sum(-1) > 0).sum(-1).clamp(min=1).to(torch.float64)
        return orbit_totals(b) / nn[:, None]
    x, y = feats(ref_batch), feats(pred_batch)
    k = lambda p, q: torch.exp(-torch.cdist(p, q).pow(2) / (2 * sigma * sigma)).mean()
    return k(x, x) + k(y, y) - 2 * k(x, y)


def eval_torch_batch(ref_batch, pred_batch, methods=None, bins=100):
    """reference evaluation/stats.py:340-344 on the device: ref_batch / pred_batch are [G, N, N] adjacency tensors (cuda).
    Returns {'degree': mmd, 'cluster': mmd, 'orbit': mmd} as python floats (one host sync at the end)."""
    from .. import ops
    methods = list(methods) if methods is not None else ['degree', 'cluster', 'orbit']
    n = max(ref_batch.shape[-1], pred_batch.shape[-1])

    import torch

    def hist(batch):
        if batch.shape[-1] != n:     # pad to a common node count so that degree histograms share their support
            batch = torch.nn.functional.pad(batch, (0, n - batch.shape[-1], 0, n - batch.shape[-1]))
        return ops.graph_stats(batch, bins=bins)
    d1, c1 = hist(ref_batch)
    d2, c2 = hist(pred_batch)
    out = {}
    if 'degree' in methods:
        out['degree'] = ops.mmd_gaussian_emd(d1, d2, sigma=1.0, distance_scaling=1.0)[0]
    if 'cluster' in methods:
        out['cluster'] = ops.mmd_gaussian_emd(c1, c2, sigma=1.0 / 10, distance_scaling=float(bins))[0]
    if 'orbit' in methods:
        out['orbit'] = orbit_mmd_batch(ref_batch, pred_batch)
    return {k: float(v.item()) for k, v in out.items()}


def _dense_batch(graphs):
    """list of networkx graphs / dense numpy adjacencies -> [G, n, n] float32 array (zero padded)."""
    import networkx as nx
    mats = [np.asarray(g, dtype=np.float32) if isinstance(g, np.ndarray) else nx.to_numpy_array(g, dtype=np.float32) for g in graphs]
    n = max(max(m.shape[0] for m in mats), 1)
    out = np.zeros((len(mats), n, n), np.float32)
    for i, m in enumerate(mats):
        out[i, :m.shape[0], :m.shape[1]] = m
    return out


def eval_graph_list(graph_ref_list, graph_pred_list, methods=None):
    if _EVALUATOR is not None:
        return _EVALUATOR(graph_ref_list, graph_pred_list, methods=methods)
    import torch
    if torch.cuda.is_available():
        want = [m for m in (methods or ['degree', 'cluster', 'orbit']) if m in ('degree', 'cluster', 'orbit')]
        return eval_torch_batch(torch.from_numpy(_dense_batch(graph_ref_list)).cuda(),
                                torch.from_numpy(_dense_batch(graph_pred_list)).cuda(), methods=want)
    return {"degree": degree_mmd(graph_ref_list, graph_pred_list)}
