"""debug: fused instance norm vs the standalone pass on the ill-conditioned (bias-shifted) network of test_backward_kinkfree_h128"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from oracle import ppgn as op
from discrete_dppm_graphs_amd import ops
from test_gpu_ppgn import rand_case, run_gpu
B, N, H, L = 4, 96, 128, 2
for sh in [(8.0, 8.0, 8.0), (-8.0, -8.0, -8.0), (8.0, -8.0, -8.0), (0.0, 0.0, 0.0)]:
    rng = np.random.default_rng(7)
    A, mask = rand_case(rng, B, N)
    sigma = rng.uniform(0, 0.5, size=B)
    W = op.init_weights(L, H, seed=5)
    for k in list(W):
        if k.endswith("m1.0.bias"): W[k] = W[k] + sh[0]
        elif k.endswith("m2.0.bias"): W[k] = W[k] + sh[1]
        elif k == "after_conv.0.bias": W[k] = W[k] + sh[2]
    Wd = {k: v.to(torch.float64) for k, v in W.items()}
    l64 = op.powerful_forward(Wd, A, mask, sigma, L, dtype=torch.float64).squeeze(-1).numpy()
    sc = np.abs(l64).max()
    res = {}
    for tag, env in [("fused", {}), ("standalone", {"DPPM_DISABLE_NORMFUSE": "1"}), ("w3fused", {"DPPM_DISABLE_W4": "1"}), ("fused_exactstats", {"DPPM_DBG": "4"})]:
        with ops.debug_switch(**{k[5:]: int(v) for k, v in env.items()}):
            res[tag], _ = run_gpu(ops, W, A, mask, sigma, L, H, precision="bf16x3")
    m = mask.astype(bool) if mask.dtype != bool else mask
    print(sh, "scale", sc, {k: float(np.abs(v - l64).max() / sc) for k, v in res.items()},
          "fused-standalone", float(np.abs(res["fused"] - res["standalone"]).max() / sc))
    d = np.abs(res["fused"] - l64)
    b, i, j = np.unravel_index(d.argmax(), d.shape)
    print("   worst at", b, i, j, "mask there", mask[b, i, j], "n nodes", [int(mask[q, :, 0].sum()) if mask.ndim == 3 else None for q in range(B)])
