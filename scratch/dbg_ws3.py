# kink-free H=128 gradient case: GPU and the f32 oracle against the f64 oracle (run with DPPM_DISABLE_WSTAT unset / =1)
import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_ppgn import run_gpu, rand_case
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
B, N, H, L = 4, 96, 128, 2
shift = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
rng = np.random.default_rng(7)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=5)
for k in list(W):
    if k.endswith("m1.0.bias") or k.endswith("m2.0.bias") or k == "after_conv.0.bias":
        W[k] = W[k] + shift
R = rng.normal(size=(B, N, N)).astype(np.float32) * mask
def run(dt):
    Wg = {k: v.to(dt).clone().requires_grad_(True) for k, v in W.items()}
    ref = op.powerful_forward(Wg, A, mask, sigma, L, dtype=dt).squeeze(-1)
    (ref * torch.from_numpy(R).to(dt)).sum().backward()
    return ref.detach().numpy(), {k: v.grad.numpy() for k, v in Wg.items()}
l32, g32 = run(torch.float32)
l64, g64 = run(torch.float64)
logits, grads = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=R)
print("logit err gpu", np.abs(logits - l64).max(), "oracle32", np.abs(l32 - l64).max())
for n in g64:
    s = max(np.abs(g64[n]).max(), 1e-3)
    eg = np.abs(grads[n] - g64[n]).max() / s; eo = np.abs(g32[n] - g64[n]).max() / s
    print(f"  {n:28s} gpu {eg:.2e} oracle32 {eo:.2e} ratio {eg/max(eo,1e-9):.2f}")
