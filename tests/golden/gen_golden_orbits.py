"""Golden vectors for the orbit statistic (reference evaluation/stats.py:210-273), produced by the reference's own ORCA source
(evaluation/orca/orca.cpp) compiled here by oracle/Makefile into oracle/_ref/orca and driven exactly as evaluation/stats.py:210-236
drives it (`orca node 4 <edge list> std`, counts parsed after 'orbit counts: '):

    make -C oracle _ref/orca && python tests/golden/gen_golden_orbits.py     # writes tests/golden/g7_orbits.npz

Each graph is stored as a dense adjacency with its isolated nodes already removed (what `adjs_to_graphs` hands to `orca`) next
to ORCA's per-node [n, 15] counts.  Only data is committed.
"""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORCA = os.path.join(HERE, "..", "..", "oracle", "_ref", "orca")
START = "orbit counts: \n"


def run_orca(adj):
    n = adj.shape[0]
    iu, ju = np.nonzero(np.triu(adj, 1))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(f"{n} {len(iu)}\n")
        for u, v in zip(iu, ju):
            f.write(f"{u} {v}\n")
        path = f.name
    out = subprocess.check_output([ORCA, "node", "4", path, "std"]).decode("utf8").strip()
    os.remove(path)
    out = out[out.find(START) + len(START):]
    return np.array([list(map(int, row.strip().split(" "))) for row in out.strip("\n").split("\n")], np.int64)


def graphs():
    rng = np.random.default_rng(707)
    out = []
    for n, p in [(4, 0.6), (6, 0.5), (9, 0.4), (12, 0.3), (12, 0.8), (17, 0.25), (20, 0.5), (25, 0.15), (40, 0.1), (60, 0.08)]:
        up = np.triu(rng.random((n, n)) < p, 1)
        a = (up | up.T).astype(np.float32)
        keep = a.sum(1) > 0
        out.append(a[np.ix_(keep, keep)])
    k5 = 1.0 - np.eye(5, dtype=np.float32)                       # complete graph
    c6 = np.zeros((6, 6), np.float32)
    for i in range(6):
        c6[i, (i + 1) % 6] = c6[(i + 1) % 6, i] = 1              # cycle
    star = np.zeros((7, 7), np.float32); star[0, 1:] = star[1:, 0] = 1
    grid = np.zeros((12, 12), np.float32)                        # 3 x 4 grid
    for r in range(3):
        for c in range(4):
            if c < 3: grid[4 * r + c, 4 * r + c + 1] = grid[4 * r + c + 1, 4 * r + c] = 1
            if r < 2: grid[4 * r + c, 4 * r + c + 4] = grid[4 * r + c + 4, 4 * r + c] = 1
    return out + [k5, c6, star, grid]


if __name__ == "__main__":
    data = {}
    gs = graphs()
    for i, a in enumerate(gs):
        data[f"adj{i}"] = a.astype(np.uint8)
        data[f"counts{i}"] = run_orca(a)
        assert data[f"counts{i}"].shape == (a.shape[0], 15)
    data["num"] = np.int64(len(gs))
    np.savez_compressed(os.path.join(HERE, "g7_orbits.npz"), **data)
    print("G7", len(gs), "graphs; totals of the last:", data[f"counts{len(gs) - 1}"].sum(0))
