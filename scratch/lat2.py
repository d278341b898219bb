import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_dppm_graphs_amd import engine, ops
from discrete_dppm_graphs_amd.model.ppgn import Powerful
N, H, L, B = 18, 32, 6, 32
torch.manual_seed(0)
model = Powerful(num_layers=L, hidden=H, hidden_final=H, n_nodes=N, normalization="instance", adj_out=True, input_features=2, dropout_p=1e-6).cuda().eval()
model.use_hip_graphs = False
flags = torch.ones(B, N, device="cuda")
for _ in range(3): engine.sample_chain_simple(model, flags, 16, 1)
torch.cuda.synchronize()
