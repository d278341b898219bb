"""Boundary to the reference's evaluation side branch (evaluation/stats.py), which is OUT OF SCOPE for the accelerated
hot path (CPU graph statistics on networkx / pyemd / ORCA).  Only what the samplers call is provided:
`adjs_to_graphs` and an `eval_graph_list` hook.  Register the reference's own evaluator (or any other) with
`set_evaluator(fn)`; the default computes a degree-histogram MMD (Gaussian-TV kernel) with numpy only."""
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


def eval_graph_list(graph_ref_list, graph_pred_list, methods=None):
    if _EVALUATOR is not None:
        return _EVALUATOR(graph_ref_list, graph_pred_list, methods=methods)
    return {"degree": degree_mmd(graph_ref_list, graph_pred_list)}
