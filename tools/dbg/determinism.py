"""debug: is the bf16x3 inference forward bit-identical from run to run?  usage: determinism.py B N L runs   (env switches apply)"""
import os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
from test_gpu_ppgn import rand_case, run_gpu
B, N, L, R = (int(x) for x in sys.argv[1:5])
H = 128
rng = np.random.default_rng(1200 + N + L)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=N + 3 * L)
import hashlib, collections
cnt = collections.Counter()
for r in range(R):
    out, _ = run_gpu(ops, W, A, mask, sigma, L, H, precision="bf16x3")
    cnt[hashlib.md5(out.tobytes()).hexdigest()[:8]] += 1
print(os.environ.get("TAG", ""), "distinct results over", R, "runs:", len(cnt), sorted(cnt.values(), reverse=True))
