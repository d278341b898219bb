"""GPU parity of the PPGN noise-conditioning variants (Powerful_Ind / _Ind_Mult / _sincos) through the
C-ABI: against logits produced by the reference's own modules (tests/golden/g6_variant_*.npz, strict state_dict load: the key
names and shapes are the reference's) and, at H = 128 (weight-stationary / split-bf16 kernels), against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def build(kind, L, H, N, noise_mlp):
    from discrete_dppm_graphs_amd.model.ppgn_indnoise import Powerful_Ind, Powerful_sincos
    from discrete_dppm_graphs_amd.model.ppgn_indnoise_multdim import Powerful_Ind_Mult
    cls = {"ind": Powerful_Ind, "ind_mult": Powerful_Ind_Mult, "sincos": Powerful_sincos}[kind]
    return cls(num_layers=L, input_features=1 if kind == "sincos" else 2, hidden=H, hidden_final=H, n_nodes=N,
               normalization="instance", adj_out=True, noise_mlp=noise_mlp, dropout_p=1e-6)


def close(got, ref, tol):
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(np.asarray(got, np.float64) - ref).max()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("kind,tag", [("ind", "n12_h16_l2"), ("ind", "n12_h16_l2_mlp"), ("ind_mult", "n9_h32_l3_mlp"),
                                      ("sincos", "n12_h16_l2"), ("sincos", "n9_h32_l3")])
def test_variant_golden(kind, tag):
    g = np.load(os.path.join(G, f"g6_variant_{kind}_{tag}.npz"))
    L, H, N = int(g["L"]), int(g["H"]), int(g["N"])
    model = build(kind, L, H, N, bool(g["noise_mlp"]))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    with torch.no_grad():
        out = model(None, torch.from_numpy(g["A"]).cuda(), torch.from_numpy(g["mask"]).cuda(), float(g["sigma"]))
    assert out.shape == g["logits"].shape + (1,)
    close(out.squeeze(-1).cpu().numpy(), g["logits"], 2e-4)
    with pytest.raises(NotImplementedError):      # training these variants is not provided: loud, not silent
        model(None, torch.from_numpy(g["A"]).cuda(), torch.from_numpy(g["mask"]).cuda(), float(g["sigma"]))


@pytest.mark.parametrize("precision,tol", [("f32", 2e-4), ("bf16x3", 1e-3)])
@pytest.mark.parametrize("kind", ["ind", "ind_mult", "sincos"])
def test_variant_h128_vs_oracle(kind, precision, tol):
    from oracle import ppgn as op
    L, H, N, B = 2, 128, 40, 2
    torch.manual_seed(11)
    model = build(kind, L, H, N, True)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("times.") or n.startswith("time_mlp."):
                p.mul_(2.0)
    W = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(5)
    A = np.zeros((B, N, N), np.float32)
    for b in range(B):
        n = N - 7 * b
        up = np.triu((rng.random((n, n)) < 0.3).astype(np.float32), 1)
        for i in range(n - 1):
            up[i, i + 1] = 1
        A[b, :n, :n] = up + up.T
    flags = (A.sum(-1) > 1e-5).astype(np.float32)
    mask = flags[:, :, None] * flags[:, None, :]
    with torch.no_grad():
        if kind == "sincos":
            ref = op.powerful_sincos_forward(W, A, mask, 0.21, L, H)
        else:
            ref = op.powerful_ind_forward(W, A, mask, 0.21, L, noise_mlp=True, mult=kind == "ind_mult")
    model = model.cuda().eval()
    model.precision = precision
    with torch.no_grad():
        out = model(None, torch.from_numpy(A).cuda(), torch.from_numpy(mask).cuda(), 0.21)
    close(out.squeeze(-1).cpu().numpy(), ref.squeeze(-1).numpy(), tol)


def test_factory_builds_variants():
    from discrete_dppm_graphs_amd.utils.easydict import EasyDict
    from discrete_dppm_graphs_amd.utils.loading_utils import get_score_model
    for name, cls in (("ppgn_ind", "Powerful_Ind"), ("ppgn_ind_mult", "Powerful_Ind_Mult"), ("ppgn_ind_sincos", "Powerful_sincos")):
        cfg = EasyDict(dict(model=dict(name=name), num_layers=2, hidden=16, hidden_final=16, dropout=1e-6, n_nodes=10,
                            normalization="instance", noise_mlp=True, dev=torch.device("cuda")))
        m = get_score_model(cfg)
        assert type(m).__name__ == cls and next(m.parameters()).is_cuda
