import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_dppm_graphs_amd import engine, ops
from discrete_dppm_graphs_amd.model.ppgn import Powerful
N, H, L, B = 18, 32, 6, 32
torch.manual_seed(0)
model = Powerful(num_layers=L, hidden=H, hidden_final=H, n_nodes=N, normalization="instance", adj_out=True, input_features=2, dropout_p=1e-6).cuda().eval()
flags = torch.ones(B, N, device="cuda")
mask = ops.generate_mask(flags)
x = ops.init_xT(flags, seed=1)
a_t = torch.empty_like(x); logits = torch.empty_like(x)
def t(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("edgeflip      us", t(lambda: ops.edgeflip(x, flags, 0.3, seed=1, offset=3, want_target=False, out=a_t)))
print("infer         us", t(lambda: model.infer(a_t, mask, 0.3, out=logits)))
print("reverse       us", t(lambda: ops.reverse_step_simple(a_t, logits, flags, seed=1, offset=4, out=x)))
A, m, s = model._prepare(a_t, mask, 0.3)
ws = model._workspace(B, N, False, A.device); pack = model._pack()
print("ppgn_forward  us", t(lambda: ops.ppgn_forward(pack, A, m, s, ws, out=logits)))
print("_prepare      us", t(lambda: model._prepare(a_t, mask, 0.3)))
print("chain/level   us", t(lambda: engine.sample_chain_simple(model, flags, 16, 1), 20) / 16)
