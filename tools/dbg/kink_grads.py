"""debug: per-parameter gradient error ratios (GPU vs torch-f32, both against float64) on the ill-conditioned network of
test_backward_kinkfree_h128[8,8,8], for several arithmetic configurations of the training path"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import ppgn as op
from discrete_dppm_graphs_amd import ops
from test_gpu_ppgn import rand_case, run_gpu
B, N, H, L = 4, 96, 128, 2
rng = np.random.default_rng(7)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=5)
for k in list(W):
    if k.endswith("m1.0.bias") or k.endswith("m2.0.bias") or k == "after_conv.0.bias":
        W[k] = W[k] + 8.0
Rr = rng.normal(size=(B, N, N)).astype(np.float32) * mask
def oracle(dt):
    Wg = {k: v.to(dt).clone().requires_grad_(True) for k, v in W.items()}
    out = op.powerful_forward(Wg, A, mask, sigma, L, dtype=dt).squeeze(-1)
    (out * torch.from_numpy(Rr).to(dt)).sum().backward()
    return out.detach().numpy(), {k: v.grad.numpy() for k, v in Wg.items()}
l64, g64 = oracle(torch.float64)
l32, g32 = oracle(torch.float32)
gmax = max(np.abs(v).max() for v in g64.values())
for tag, env in [("all exact f32", ("DPPM_DISABLE_W6", "DPPM_DISABLE_PLANE6", "DPPM_WGRAD_F32")), ("round-1 (wgrad split)", ("DPPM_DISABLE_W6", "DPPM_DISABLE_PLANE6")),
                 ("+plane6", ("DPPM_DISABLE_W6",)), ("+w6 cols", ("DPPM_DISABLE_W6MLP",)), ("all on the bf16 pipe", ())]:
    with ops.debug_switch(**{k[5:]: 1 for k in env}):
        logits, grads = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=Rr)
    worst = []
    for n in g64:
        scale = max(np.abs(g64[n]).max(), 2e-2 * gmax)
        eg, e32 = np.abs(grads[n] - g64[n]).max(), np.abs(g32[n] - g64[n]).max()
        worst.append(((eg - 2e-4 * scale) / max(e32, 1e-30), n, eg, e32))
    worst.sort(reverse=True)
    print("%-24s logits err/f32err %.2f | worst params: %s" % (tag, np.abs(logits - l64).max() / np.abs(l32 - l64).max(),
          "; ".join("%s %.2fx" % (n, r) for r, n, _, _ in worst[:4])))
