"""debug: which workspace regions differ between two runs of the same bf16x3 inference forward?  usage: ws_diff.py B N L runs"""
import os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
from test_gpu_ppgn import rand_case, dev
B, N, L, R = (int(x) for x in sys.argv[1:5])
H = 128
rng = np.random.default_rng(1200 + N + L)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=N + 3 * L)
tensors = {k: dev(v.detach().numpy()) for k, v in W.items()}
pack = ops.ParamPack(tensors, L, H)
sig = dev(np.asarray(sigma, np.float32))
ld = -(-N // 4) * 4; P = N * ld; S = B * H * P; CT = 2 + L * H
ntiles = B * (-(-P // 64))
regions = [("maskp", B * P), ("inv_norm", B), ("cat", B * CT * P), ("u0", S), ("R", S), ("act", S), ("rstd", L * B * H),
           ("tmp0(mm)", S), ("tmp1(o1)", S), ("tmp2(o2)", S), ("tmp3", S), ("himg", None), ("nstat", ntiles * H * 2), ("ncnt", ntiles), ("naff", (L + 1) * B * 2 * H)]
off = 0; bounds = []
for name, n in regions:
    if n is None:
        n = None
    bounds.append([name, off, n]); 
    if n is not None: off += -(-n // 64) * 64
def run():
    tensors = {k: dev(v.detach().numpy()) for k, v in W.items()}
    pack = ops.ParamPack(tensors, L, H)
    ws = ops.ppgn_workspace(B, N, H, L, False, "cuda")
    ws.fill_(float("nan"))
    out = ops.ppgn_forward(pack, dev(A), dev(mask), sig, ws, training=False, precision="bf16x3")
    torch.cuda.synchronize()
    wi = ws.view(torch.int32)
    sums = {}
    o = 0
    for name, lo, n in bounds:
        if n is None: continue
        sums[name] = int(wi[lo:lo + n].to(torch.int64).sum().item())
    sums["logits"] = int(out.view(torch.int32).to(torch.int64).sum().item())
    return sums
# himg has unknown size: regions after it are located from the end
total = ops.ppgn_workspace(B, N, H, L, False, "cuda").numel()
tail = [b for b in bounds if b[0] in ("nstat", "ncnt", "naff")]
end = total
for b in reversed(tail):
    b[1] = end - (-(-b[2] // 64) * 64); end = b[1]
import collections
base = run(); cnt = collections.Counter()
for r in range(R):
    s_ = run()
    bad = tuple(k for k in s_ if s_[k] != base[k])
    cnt[bad] += 1
    if bad: print("run", r, "regions differing from run 0:", bad, flush=True)
print("summary:", dict(cnt))
