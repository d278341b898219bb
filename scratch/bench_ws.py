# stand-alone timing of the weight-stationary MLP pair at the bench geometry (B=32, P=200*200, H=128)
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discrete_dppm_graphs_amd import ops
B, H, P = 32, 128, 40000
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(B, H, P, device="cuda", generator=g)
maskp = (torch.rand(B, P, device="cuda", generator=g) < 0.9).float()
Wt = {}
for m in ("m1", "m2"):
    for k in ("0", "2"):
        Wt[f"{m}.{k}.weight"] = torch.randn(H, H, device="cuda", generator=g) / H ** 0.5
        Wt[f"{m}.{k}.bias"] = torch.randn(H, device="cuda", generator=g)
for save in (False, True):
    outs = ops.mlp_pair_fwd(X, Wt, maskp=maskp, save_hidden=save)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        outs = ops.mlp_pair_fwd(X, Wt, maskp=maskp, save_hidden=save)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"mlp_pair save={save}: {ms:.3f} ms  {8*H*H*P*B/ms/1e9:.1f} TF")
