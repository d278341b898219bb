"""Diagnostic (not product): per-phase s_memtime stamps of the fused small-graph kernel, layer 2, waves 0 and 1 of workgroup 0.
Loads scratch/sgdiag/libdppm_hip_diag.so (dppm_smallgraph.hip built with -DSG_STAMP) in place of the product library."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import discrete_dppm_graphs_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "scratch", "sgdiag", "libdppm_hip_diag.so")
L.lib = L._load()
from discrete_dppm_graphs_amd import engine, ops
from discrete_dppm_graphs_amd.model.ppgn import Powerful
N, H, Ln, B = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 32, 6, 32
torch.manual_seed(0)
model = Powerful(num_layers=Ln, hidden=H, hidden_final=H, n_nodes=N, normalization="instance", adj_out=True, input_features=2, dropout_p=1e-6).cuda().eval()
flags = torch.ones(B, N, device="cuda")
for _ in range(3):
    engine.sample_chain_simple(model, flags, 16, 1, 0)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
lib = L.lib
lib.dppm_sg_read_stamps.restype = ctypes.c_int
assert lib.dppm_sg_read_stamps(out) == 0
names = ["mlp", "barrier+weights", "contraction", "barrier", "m4", "totals1", "totals2", "norm+readout", "commit"]
for w in range(2):
    t = [out[16 * w + i] for i in range(10)]
    print("wave", w, "layer cycles", t[9] - t[0], " ".join(f"{n}={t[i + 1] - t[i]}" for i, n in enumerate(names)))
