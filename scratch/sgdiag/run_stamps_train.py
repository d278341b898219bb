"""Diagnostic: phase stamps of the TRAINING form of the small-graph kernel (sg_forward_kernel<true, false, 0>), layer 2."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import discrete_dppm_graphs_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "scratch", "sgdiag", "libdppm_hip_diag.so")
L.lib = L._load()
from discrete_dppm_graphs_amd import ops
from discrete_dppm_graphs_amd.model.ppgn import Powerful
N, H, Ln, B = 20, 32, 6, 32
torch.manual_seed(0)
model = Powerful(num_layers=Ln, hidden=H, hidden_final=H, n_nodes=N, normalization="instance", adj_out=True, input_features=2, dropout_p=1e-6).cuda()
A = (torch.rand(B, N, N, device="cuda") < 0.3).float(); A = torch.triu(A, 1); A = A + A.transpose(1, 2)
mask = torch.ones(B, N, N, device="cuda"); sigma = torch.full((B,), 0.2, device="cuda")
ws = model._workspace(B, N, True, A.device, cached=True)
train = len(sys.argv) < 2 or sys.argv[1] != "infer"
for _ in range(3):
    ops.ppgn_forward(model._pack(), A, mask, sigma, ws, training=train, precision="f32")
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
L.lib.dppm_sg_read_stamps.restype = ctypes.c_int
assert L.lib.dppm_sg_read_stamps(out) == 0
names = ["mlp", "barrier+weights", "contraction", "barrier", "m4", "totals1", "totals2", "norm+readout", "commit"]
for w in range(2):
    t = [out[16 * w + i] for i in range(10)]
    k = [out[16 * w + i] for i in range(10, 14)]
    print("  whole kernel: setup", k[1] - k[0], "layers", k[2] - k[1], "head", k[3] - k[2])
    print("train" if train else "infer-f32", "wave", w, "layer cycles", t[9] - t[0], " ".join(f"{n}={t[i + 1] - t[i]}" for i, n in enumerate(names)))
