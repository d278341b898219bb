"""Orbit statistic (reference evaluation/stats.py:210-273): the oracle's subset enumeration and the product's closed forms
(evaluation/stats.py::orbit_totals, float64 integer arithmetic over dense products) against the counts of the reference's own
ORCA source (tests/golden/g7_orbits.npz, see gen_golden_orbits.py).  Integer work: bit-exact."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_orbits.npz")


def _golden():
    z = np.load(GOLD)
    return [(z[f"adj{i}"].astype(np.float32), z[f"counts{i}"]) for i in range(int(z["num"]))]


def test_oracle_enumeration_equals_orca():
    from oracle import stats as os_
    for adj, counts in _golden():
        if adj.shape[0] <= 30:                       # C(n, 4) python iterations
            assert np.array_equal(os_.orbit_totals_bruteforce(adj), counts.sum(0)), adj.shape


def _padded(graphs, N):
    out = np.zeros((len(graphs), N, N), np.float32)
    for g, (a, _) in enumerate(graphs):
        out[g, :a.shape[0], :a.shape[0]] = a
    return out


def _check_closed_forms(device):
    from discrete_dppm_graphs_amd.evaluation.stats import orbit_totals
    gold = _golden()
    batch = _padded(gold, 64)                        # padding = isolated nodes: contributes to no orbit
    batch[0, 5, 5] = 1.0                             # a self loop is ignored (adjs_to_graphs removes them)
    got = orbit_totals(torch.from_numpy(batch).to(device)).cpu().numpy()
    for g, (_, counts) in enumerate(gold):
        assert np.array_equal(got[g].astype(np.int64), counts.sum(0)), g
        assert np.array_equal(got[g], np.round(got[g]))


def test_closed_forms_equal_orca_cpu():
    _check_closed_forms("cpu")


@pytest.mark.gpu
def test_closed_forms_equal_orca_gpu():
    _check_closed_forms("cuda")


def test_orbit_mmd_equals_oracle():
    from discrete_dppm_graphs_amd.evaluation.stats import orbit_mmd_batch
    from oracle import stats as os_
    gold = _golden()
    small = [g for g in gold if g[0].shape[0] <= 20]
    ref, pred = small[: len(small) // 2], small[len(small) // 2:]
    # the oracle fed with ORCA's own totals = the reference's orbit_stats_all on these graphs
    tot = {a.tobytes(): c.sum(0) for a, c in small}
    want = os_.orbit_mmd([a for a, _ in ref], [a for a, _ in pred], totals=lambda a: tot[np.asarray(a, np.float32).tobytes()])
    got = float(orbit_mmd_batch(torch.from_numpy(_padded(ref, 24)), torch.from_numpy(_padded(pred, 24))))
    assert abs(got - want) <= 1e-12
    assert abs(float(orbit_mmd_batch(torch.from_numpy(_padded(ref, 24)), torch.from_numpy(_padded(ref, 24))))) < 1e-12
