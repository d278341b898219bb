"""debug: which part of a position's channel vector was wrong when the layer-0 MLP output o1 of the in_lin-fused kernel differs from
the unfused path?  For each bad position the script replaces, in float64 on the host, the in_lin inputs (A, S) of a SUBSET of the 128
channels by those of another candidate position and looks for the subset / source that reproduces the GPU's value.  Subsets: one
staging lane's 4 channels (32 of them), one staging wave's 32 channels."""
import os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
from test_gpu_ppgn import rand_case, dev
B, N, L, H = 5, 64, 1, 128
rng = np.random.default_rng(1200 + N + 3)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=N + 3 * L)
tensors = {k: dev(v.detach().numpy()) for k, v in W.items()}
pack = ops.ParamPack(tensors, L, H)
sig = dev(np.asarray(sigma, np.float32))
ld = -(-N // 4) * 4; P = N * ld; S = B * H * P; CT = 2 + L * H
al = lambda n: -(-n // 64) * 64
off_tmp1 = al(B * P) + al(B) + al(B * CT * P) + 3 * al(S) + al(L * B * H) + al(S)
def run():
    ws = ops.ppgn_workspace(B, N, H, L, False, "cuda"); ws.fill_(float("nan"))
    if os.environ.get("O1_POISON"):
        import ctypes as C
        from discrete_dppm_graphs_amd import _lib
        rc = _lib.lib.dppm_test_fill_lds(int(os.environ["O1_POISON"], 0), C.c_void_p(torch.cuda.current_stream().cuda_stream)); assert rc == 0, rc
    ops.ppgn_forward(pack, dev(A), dev(mask), sig, ws, training=False, precision="bf16x3")
    torch.cuda.synchronize()
    return ws[off_tmp1:off_tmp1 + S].cpu().numpy().reshape(B, H, P)
got = run()
with ops.debug_switch(DISABLE_INLINFUSE=1):
    ref = run()
print("nan in got", int(np.isnan(got).sum()), "nan in ref", int(np.isnan(ref).sum()))
d = np.abs(got - ref); sc = np.abs(ref).max()
badpos = np.argwhere((d > 1e-4 * sc).any(axis=1))          # (graph, position)
print("bad positions", len(badpos), "in tiles", sorted(set((int(b) * (P // 64) + int(p) // 64) for b, p in badpos))[:40])
w = {k: v.double().numpy() for k, v in W.items()}
Win, bin_ = w["in_lin.0.weight"], w["in_lin.0.bias"]
W1, b1, W2, b2 = w["convs.0.m1.0.weight"], w["convs.0.m1.0.bias"], w["convs.0.m1.2.weight"], w["convs.0.m1.2.bias"]
def mlp(u):
    h = W1 @ u + b1; h = np.where(h > 0, h, 0.01 * h); return W2 @ h + b2
Sdiag = np.zeros((B, N, ld));
for b in range(B): Sdiag[b, np.arange(N), np.arange(N)] = np.float32(sigma[b])
Af = np.zeros((B, N, ld)); Af[:, :, :N] = A
Af = Af.reshape(B, P); Sf = Sdiag.reshape(B, P)
mk = np.zeros((B, N, ld)); mk[:, :, :N] = mask; mk = mk.reshape(B, P)
def u0(a, s): return Win[:, 0] * a + Win[:, 1] * s + bin_
import collections
tally = collections.Counter()
shown = 0
for b, p in badpos:
    g = got[b, :, p].astype(np.float64); r = ref[b, :, p]
    base = u0(Af[b, p], Sf[b, p])
    best = None
    for wv in range(4):
        for half in range(2):
            for pr in range(2):
                idx = np.array([64 * half + 16 * wv + 4 * c + 2 * pr + k for c in range(4) for k in range(2)])
                for what, a_, s_ in (("A term dropped", 0.0, Sf[b, p]), ("S term dropped", Af[b, p], 0.0), ("both dropped", 0.0, 0.0), ("A=1", 1.0, Sf[b, p])):
                    u = base.copy(); u[idx] = u0(a_, s_)[idx]
                    e = np.abs(mlp(u) * mk[b, p] - g).max()
                    if best is None or e < best[0]: best = (e, wv, half, pr, what)
    ok = best[0] < 2e-4
    tally[(ok,) + (best[1:] if ok else ())] += 1
    if shown < 10:
        shown += 1
        print("graph %d tile %d in-tile %d (A=%g S=%g) |got-ref| %.3e: wave %d half %d pair %d: %s, residual %.2e" % (b, b * (P // 64) + p // 64, p % 64, Af[b, p], Sf[b, p], np.abs(g - r).max(), best[1], best[2], best[3], best[4], best[0]))
print("explained by ONE wave-instruction (8 channels: one channel pair of the wave's four channel groups) having lost a term:", sum(v for k, v in tally.items() if k[0]), "of", len(badpos))
print(sorted(tally.items(), key=lambda kv: -kv[1])[:12])
