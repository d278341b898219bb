import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_ppgn import run_gpu, rand_case
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
for (B,N,H,L) in [(1,129,32,1)]:
    rng = np.random.default_rng(B * 1000 + N)
    A, mask = rand_case(rng, B, N)
    sigma = rng.uniform(0, 0.5, size=B)
    W = op.init_weights(L, H, seed=N)
    R = rng.normal(size=(B, N, N)).astype(np.float32) * mask
    def run(dt):
        Wg = {k: v.clone().to(dt).requires_grad_(True) for k, v in W.items()}
        torch.set_default_dtype(dt)
        # oracle casts inputs to float32 internally; emulate f64 by monkeypatching
        ref = op.powerful_forward(Wg, A, mask, sigma, L).squeeze(-1)
        (ref * torch.from_numpy(R).to(dt)).sum().backward()
        torch.set_default_dtype(torch.float32)
        return ref.detach().numpy(), {k: v.grad.numpy() for k, v in Wg.items()}
    ref32, g32 = run(torch.float32)
    logits, grads = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=R)
    for n in g32:
        r = g32[n]; e = np.abs(grads[n]-r).max()
        print(f"  {n:28s} err {e:.3e} scale {np.abs(r).max():.3e} rel {e/np.abs(r).max():.2e}")
