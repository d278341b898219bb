"""Golden vectors for the PPGN noise-conditioning variants, produced by running the REFERENCE's own
modules (model/ppgn_indnoise.py, model/ppgn_indnoise_multdim.py, model/ppgn_indnoise_sincos.py) in this container:

    python tests/golden/gen_golden_variants.py          # needs /root/reference; writes tests/golden/g6_variant_*.npz

Each file holds the seeded state_dict, the inputs (A_t, mask, one scalar sigma as the samplers pass it) and the logits of one
batched eval() call.  Only data is committed; the stubs for absent non-arithmetic imports come from gen_golden.py.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  (installs the stubs and puts the reference on sys.path)

from model.ppgn_indnoise import Powerful_Ind  # noqa: E402
from model.ppgn_indnoise_multdim import Powerful_Ind_Mult  # noqa: E402
from model.ppgn_indnoise_sincos import Powerful_sincos  # noqa: E402

KINDS = {"ind": (Powerful_Ind, 2), "ind_mult": (Powerful_Ind_Mult, 2), "sincos": (Powerful_sincos, 1)}


def gen(kind, tag, N, H, L, B, seed, noise_mlp):
    cls, in_feat = KINDS[kind]
    rng = np.random.default_rng(seed)
    A = gg.make_graphs(rng, B, N, max(3, N // 2))
    flags = (A.sum(-1) > 1e-5).astype(np.float32)
    mask = flags[:, :, None] * flags[:, None, :]
    torch.manual_seed(seed)
    model = cls(use_norm_layers=False, name=kind, channel_num_list=[], feature_nums=[], gnn_hidden_num_list=[], num_layers=L,
                input_features=in_feat, hidden=H, hidden_final=H, dropout_p=1e-6, simplified=False, n_nodes=N,
                normalization="instance", cat_output=True, adj_out=True, output_features=1, residual=False, project_first=False,
                node_out=False, noise_mlp=noise_mlp, config=gg.CFG)
    model.eval()
    with torch.no_grad():   # away from the default init so that the conditioning path is not negligible
        for n, p in model.named_parameters():
            if n.startswith("times.") or n.startswith("time_mlp."):
                p.mul_(2.0)
    sigma = float(rng.uniform(0.05, 0.45))
    with torch.no_grad():
        logits = model(None, torch.from_numpy(A), torch.from_numpy(mask), sigma).squeeze(-1)
    out = dict(gg.state_np(model))
    out.update(A=A, mask=mask, sigma=np.float64(sigma), logits=logits.numpy(), N=N, H=H, L=L, noise_mlp=int(noise_mlp))
    np.savez_compressed(os.path.join(gg.OUT, f"g6_variant_{kind}_{tag}.npz"), **out)
    print("G6", kind, tag, "|logits|max", float(logits.abs().max()))


if __name__ == "__main__":
    torch.set_num_threads(1)
    gen("ind", "n12_h16_l2", N=12, H=16, L=2, B=3, seed=301, noise_mlp=False)
    gen("ind", "n12_h16_l2_mlp", N=12, H=16, L=2, B=3, seed=302, noise_mlp=True)
    gen("ind_mult", "n9_h32_l3_mlp", N=9, H=32, L=3, B=2, seed=303, noise_mlp=True)
    gen("sincos", "n12_h16_l2", N=12, H=16, L=2, B=3, seed=304, noise_mlp=True)
    gen("sincos", "n9_h32_l3", N=9, H=32, L=3, B=2, seed=305, noise_mlp=True)
