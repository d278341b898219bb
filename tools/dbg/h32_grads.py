import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
from test_gpu_ppgn import rand_case, run_gpu
B, N, L, H = 2, 200, 6, 32
rng = np.random.default_rng(6100 + N + L)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=N + 13 * L)
R = rng.normal(size=A.shape).astype(np.float32) * mask
lg, g = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=R)
lg2, g2 = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=R)
with ops.debug_switch(DISABLE_WIDE32=1):
    lg0, g0 = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=R)
Wg = {k: v.clone().double().requires_grad_(True) for k, v in W.items()}
ref = op.powerful_forward(Wg, A, mask, sigma, L, dtype=torch.float64).squeeze(-1)
(ref * torch.from_numpy(R).double()).sum().backward()
gmax = max(np.abs(v).max() for v in g0.values())
for k in g0:
    sc = max(np.abs(g0[k]).max(), 2e-2 * gmax)
    r64 = Wg[k].grad.numpy()
    print("%-26s wide-vs-old %.2e  wide-vs-f64 %.2e  old-vs-f64 %.2e  run-to-run %.1e" % (k, np.abs(g[k]-g0[k]).max()/sc, np.abs(g[k]-r64).max()/sc, np.abs(g0[k]-r64).max()/sc, np.abs(g[k]-g2[k]).max()/sc))
