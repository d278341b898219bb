import time, sys, numpy as np, torch
sys.path.insert(0, '.')
from discrete_dppm_graphs_amd import engine, ops
from discrete_dppm_graphs_amd.model.ppgn import Powerful
N, H, L, B = 20, 32, 6, 32
model = Powerful(num_layers=L, hidden=H, hidden_final=H, n_nodes=N, normalization="instance", adj_out=True, input_features=2, dropout_p=1e-6).cuda().eval()
flags = torch.ones(B, N, device="cuda")
mask = ops.generate_mask(flags)
x = ops.init_xT(flags, seed=1, graph_id0=0, offset=0)
a_t = torch.empty_like(x); logits = torch.empty_like(x)
sig = torch.full((B,), 0.2, device="cuda")
def timeit(name, f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:28s} cpu enqueue {1e6*(t1-t0)/n:8.1f} us   total {1e6*(t2-t0)/n:8.1f} us", flush=True)
timeit("infer", lambda: model.infer(a_t, mask, sig, out=logits))
timeit("edgeflip", lambda: ops.edgeflip(x, flags, 0.2, seed=1, graph_id0=0, offset=1, want_target=False, out=a_t))
timeit("reverse_step", lambda: ops.reverse_step_simple(a_t, logits, flags, seed=1, graph_id0=0, offset=2, out=x))
ws = model._workspace(B, N, False, x.device); pack = model._pack()
timeit("ops.ppgn_forward", lambda: ops.ppgn_forward(pack, a_t, mask, sig, ws, training=False, out=logits, precision="f32"))
# graph capture of the forward
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ops.ppgn_forward(pack, a_t, mask, sig, ws, training=False, out=logits, precision="f32")
torch.cuda.synchronize()
ref = logits.clone()
with torch.cuda.graph(g, stream=s):
    ops.ppgn_forward(pack, a_t, mask, sig, ws, training=False, out=logits, precision="f32")
logits.zero_(); g.replay(); torch.cuda.synchronize()
print("graph replay equal:", torch.equal(ref, logits))
timeit("graph replay", lambda: g.replay())
