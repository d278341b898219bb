#!/usr/bin/env python3
"""Write a synthetic SBM dataset as data/<name>.npz (k = ceil(N/25) blocks, p_intra 0.3,
p_inter 0.005, seed 1234).  The reference ships no N = 200 dataset (its sbm_200 pickle holds 21-27-node graphs)."""
import argparse
import os

import numpy as np


def sbm(rng, n, p_intra=0.3, p_inter=0.005):
    k = -(-n // 25)
    blocks = rng.integers(0, k, size=n)
    same = blocks[:, None] == blocks[None, :]
    up = np.triu(rng.random((n, n)) < np.where(same, p_intra, p_inter), 1).astype(np.float32)
    for i in range(n - 1):
        up[i, i + 1] = 1.0
    return up + up.T


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--min-nodes", type=int, default=None, help="ragged sizes in [min-nodes, nodes]")
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--name", default=None)
    ap.add_argument("--out", default="data")
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    sizes = [a.nodes if a.min_nodes is None else int(rng.integers(a.min_nodes, a.nodes + 1)) for _ in range(a.count)]
    adjs = np.zeros((a.count, a.nodes, a.nodes), np.float32)
    for g, n in enumerate(sizes):
        adjs[g, :n, :n] = sbm(rng, n)
    os.makedirs(a.out, exist_ok=True)
    name = a.name or f"sbm_n{a.nodes}_synthetic"
    np.savez_compressed(os.path.join(a.out, name + ".npz"), adjs=adjs, sizes=np.asarray(sizes))
    with open(os.path.join(a.out, name + ".txt"), "w") as f:
        f.write(f"synthetic SBM, {a.count} graphs, max node number: {a.nodes}")
    print("wrote", os.path.join(a.out, name + ".npz"))
