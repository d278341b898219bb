"""debug: are forward + backward (training path) bit-identical from run to run?  usage: train_det.py B N L runs"""
import os, sys, numpy as np, torch, collections, hashlib
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from discrete_dppm_graphs_amd import ops
from oracle import ppgn as op
from test_gpu_ppgn import rand_case, run_gpu
B, N, L, R = (int(x) for x in sys.argv[1:5])
H = 128
rng = np.random.default_rng(77 + N + L)
A, mask = rand_case(rng, B, N)
sigma = rng.uniform(0, 0.5, size=B)
W = op.init_weights(L, H, seed=N + L)
Rr = rng.normal(size=(B, N, N)).astype(np.float32) * mask
cl, cg = collections.Counter(), collections.Counter()
for r in range(R):
    logits, grads = run_gpu(ops, W, A, mask, sigma, L, H, dlogits=Rr)
    cl[hashlib.md5(logits.tobytes()).hexdigest()[:8]] += 1
    h = hashlib.md5()
    for k in sorted(grads): h.update(grads[k].tobytes())
    cg[h.hexdigest()[:8]] += 1
print("training forward logits: distinct", len(cl), sorted(cl.values(), reverse=True), "| gradients: distinct", len(cg), sorted(cg.values(), reverse=True))
