import sys, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from tests.test_gpu_ppgn import run_gpu, rand_case
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
def fwd(W, A, mask, sigma, L, dt):
    A = torch.as_tensor(A, dtype=dt); mask = torch.as_tensor(mask, dtype=dt)
    B, N, _ = A.shape
    mask4 = mask[..., None]; u = A[..., None]
    sig = torch.as_tensor(np.asarray(sigma, dtype=np.float64).astype(np.float32)).to(dt)
    nl = torch.diag_embed(sig.view(B, 1).expand(B, N))[..., None]
    u = torch.cat([u, nl], dim=-1); outs = [u]
    u = F.linear(u, W["in_lin.0.weight"], W["in_lin.0.bias"])
    for l in range(L):
        u = op.powerful_layer(u, mask4, W, l); u = op.masked_instance_norm2d(u, mask4); u = u * mask4; outs.append(u)
    out = torch.cat(outs, dim=-1)
    out = F.linear(out, W["layer_cat_lin.0.weight"], W["layer_cat_lin.0.bias"])
    out = out + F.leaky_relu(F.linear(out, W["after_conv.0.weight"], W["after_conv.0.bias"]), 0.01)
    return F.linear(out, W["final_lin.0.weight"], W["final_lin.0.bias"])
for (B,N,H,L) in [(2,60,128,2)]:
    rng = np.random.default_rng(B * 1000 + N)
    A, mask = rand_case(rng, B, N)
    sigma = rng.uniform(0, 0.5, size=B)
    W = op.init_weights(L, H, seed=N)
    R = rng.normal(size=(B, N, N)).astype(np.float32) * mask
    def run(dt):
        Wg = {k: v.clone().to(dt).requires_grad_(True) for k, v in W.items()}
        ref = fwd(Wg, A, mask, sigma, L, dt).squeeze(-1)
        (ref * torch.from_numpy(R).to(dt)).sum().backward()
        return ref.detach().numpy(), {k: v.grad.numpy() for k, v in Wg.items()}
    ref32, g32 = run(torch.float32)
    ref64, g64 = run(torch.float64)
    logits, grads = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=R)
    print("logits: gpu-vs-64", np.abs(logits-ref64).max(), " cpu32-vs-64", np.abs(ref32-ref64).max())
    for n in g32:
        s = np.abs(g64[n]).max()
        print(f"  {n:28s} gpu-vs-f64 {np.abs(grads[n]-g64[n]).max()/s:.2e}   cpu32-vs-f64 {np.abs(g32[n]-g64[n]).max()/s:.2e}")
    # torch-on-GPU fp32 (rocBLAS) for comparison of conditioning
    Wc = {k: v.clone().cuda().requires_grad_(True) for k, v in W.items()}
    A_c, m_c = torch.as_tensor(A).cuda(), torch.as_tensor(mask).cuda()
    def fwd_c(W, A, mask, sigma, L):
        B, N, _ = A.shape
        mask4 = mask[..., None]; u = A[..., None]
        sig = torch.as_tensor(np.asarray(sigma, dtype=np.float64).astype(np.float32)).cuda()
        nl = torch.diag_embed(sig.view(B, 1).expand(B, N))[..., None]
        u = torch.cat([u, nl], dim=-1); outs = [u]
        u = F.linear(u, W["in_lin.0.weight"], W["in_lin.0.bias"])
        for l in range(L):
            u = op.powerful_layer(u, mask4, W, l); u = op.masked_instance_norm2d(u, mask4); u = u * mask4; outs.append(u)
        out = torch.cat(outs, dim=-1)
        out = F.linear(out, W["layer_cat_lin.0.weight"], W["layer_cat_lin.0.bias"])
        out = out + F.leaky_relu(F.linear(out, W["after_conv.0.weight"], W["after_conv.0.bias"]), 0.01)
        return F.linear(out, W["final_lin.0.weight"], W["final_lin.0.bias"])
    rc = fwd_c(Wc, A_c, m_c, sigma, L).squeeze(-1)
    (rc * torch.from_numpy(R).cuda()).sum().backward()
    for n in g32:
        s = np.abs(g64[n]).max()
        print(f"  {n:28s} torchgpu-vs-f64 {np.abs(Wc[n].grad.cpu().numpy()-g64[n]).max()/s:.2e}")
