import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_ppgn import rand_case, dev
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
B,N,H,L=2,60,128,2
rng = np.random.default_rng(B * 1000 + N)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=N)
tensors = {k: dev(v.numpy()) for k, v in W.items()}
pack = ops.ParamPack(tensors, L, H)
ws = ops.ppgn_workspace(B, N, H, L, True, "cuda"); ws.fill_(float('nan'))
sig = dev(sigma.astype(np.float32))
logits = ops.ppgn_forward(pack, dev(A), dev(mask), sig, ws, training=True)
torch.cuda.synchronize()
r64 = lambda n: (n + 63)//64*64
P = N*N; CT = 2+L*H; S = B*H*P
off = r64(B*P) + r64(B) + r64(B*CT*P) + r64(S)
R = ws[off:off+S].cpu().numpy().reshape(B,H,P); off += r64(S)
act = ws[off:off+S].cpu().numpy().reshape(B,H,P)
tag = "nows" if os.environ.get("DPPM_DISABLE_WSTAT") else "ws"
np.save(f"gpurun_out/R_{tag}.npy", R); np.save(f"gpurun_out/act_{tag}.npy", act)
print(tag, "nan in R", np.isnan(R).sum(), "nan in act", np.isnan(act).sum())
