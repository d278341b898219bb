# per-parameter gradient errors of the H=128 case, for the weight-stationary and the generic paths (run twice: DPPM_DISABLE_WSTAT unset / =1)
import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_ppgn import run_gpu, rand_case
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
B, N, H, L = 2, 60, 128, 2
rng = np.random.default_rng(B * 1000 + N)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=N)
Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
ref = op.powerful_forward(Wg, A, mask, sigma, L).squeeze(-1)
R = rng.normal(size=(B, N, N)).astype(np.float32) * mask
(ref * torch.from_numpy(R)).sum().backward()
logits, grads = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=R)
print("logit err", np.abs(logits - ref.detach().numpy()).max())
for n, v in Wg.items():
    r = v.grad.numpy(); d = np.abs(grads[n] - r); s = max(np.abs(r).max(), 1e-3)
    bad = int((d > 1e-4 * s).sum())
    print(f"  {n:28s} err {d.max():.3e} scale {s:.3e} rel {d.max()/s:.2e} n_bad(1e-4) {bad}/{d.size}")
np.savez(sys.argv[1], **{k: v for k, v in grads.items()})
