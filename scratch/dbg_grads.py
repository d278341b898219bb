import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_ppgn import run_gpu, rand_case
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
for (B,N,H,L) in [(2,8,64,3),(1,129,32,1),(3,33,16,2),(2,20,32,2)]:
    rng = np.random.default_rng(B * 1000 + N)
    A, mask = rand_case(rng, B, N)
    sigma = rng.uniform(0, 0.5, size=B)
    W = op.init_weights(L, H, seed=N)
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    ref = op.powerful_forward(Wg, A, mask, sigma, L).squeeze(-1)
    R = rng.normal(size=(B, N, N)).astype(np.float32) * mask
    (ref * torch.from_numpy(R)).sum().backward()
    logits, grads = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=R)
    print("case", B,N,H,L, "logit err", np.abs(logits-ref.detach().numpy()).max())
    for n, v in Wg.items():
        r = v.grad.numpy(); e = np.abs(grads[n]-r).max()
        print(f"  {n:28s} err {e:.3e} scale {np.abs(r).max():.3e} {'<<<' if e > 1e-3*max(np.abs(r).max(),1e-3) else ''}")
