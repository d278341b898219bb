"""tools/chain_vlb_n20.py: community_small geometry (N = 20, 6x32, 16 levels), the VLB sampler's chain, graphs/s at 32 and 256 graphs per call.
DPPM_DISABLE_SGCHAIN=1 in the environment gives the level-by-level loop for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from discrete_dppm_graphs_amd import engine
from discrete_dppm_graphs_amd.model.ppgn import Powerful
torch.manual_seed(0)
model = Powerful(num_layers=6, hidden=32, hidden_final=32, n_nodes=20, normalization="instance", adj_out=True, input_features=2, dropout_p=1e-6).cuda().eval()
for B in (32, 256):
    flags = torch.ones(B, 20, device="cuda")
    for _ in range(3):
        engine.sample_chain_vlb(model, flags, 16, 1, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(20):
        engine.sample_chain_vlb(model, flags, 16, 1 + it, 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"vlb chain B={B}: {B / dt:.0f} graphs/s, {dt * 1e3 / 16:.4f} ms per level", "(level by level)" if os.environ.get("DPPM_DISABLE_SGCHAIN") == "1" else "(one launch)")
