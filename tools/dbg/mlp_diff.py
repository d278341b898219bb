"""debug: where does the role-split MLP's forward differ from the two-group one (DPPM_W3MLP=1)?  whole forward, bf16x3"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
from test_gpu_ppgn import rand_case, run_gpu
B, N, L, H = (int(x) for x in (sys.argv[1:5] + ["5", "64", "3", "128"][len(sys.argv) - 1:]))
rng = np.random.default_rng(1200 + N + L)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=N + 3 * L)
outs = {}
for form in ("w4", "w4b", "w4c", "w3", "w3b", "w3c"):
    with ops.debug_switch(W3MLP=int(form.startswith("w3"))):
        outs[form], _ = run_gpu(ops, W, A, mask, sigma, L, H, precision="bf16x3")
print("w4 run-to-run max diff", np.abs(outs["w4"] - outs["w4b"]).max(), np.abs(outs["w4"] - outs["w4c"]).max(), " w3 run-to-run", np.abs(outs["w3"] - outs["w3b"]).max(), np.abs(outs["w3"] - outs["w3c"]).max())
d = np.abs(outs["w4"] - outs["w3"])
print("w4 vs w3 max diff", d.max(), "scale", np.abs(outs["w3"]).max())
for b in range(B):
    bad = np.argwhere(d[b] > 1e-4 * np.abs(outs["w3"]).max())
    n = int((mask[b].sum(-1) > 0).sum())
    if len(bad): print("graph", b, "n", n, "bad entries", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:20], "cols", sorted(set(bad[:, 1].tolist()))[:20])
