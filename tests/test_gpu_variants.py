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
    # the training forward (autograd on: exact-f32 kernels + saved activations) returns the same logits
    out_t = model(None, torch.from_numpy(g["A"]).cuda(), torch.from_numpy(g["mask"]).cuda(), float(g["sigma"]))
    assert out_t.requires_grad
    close(out_t.detach().squeeze(-1).cpu().numpy(), g["logits"], 2e-4)


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


def rand_graphs(rng, B, N):
    A = np.zeros((B, N, N), np.float32)
    for b in range(B):
        n = N - 3 * b
        up = np.triu((rng.random((n, n)) < 0.3).astype(np.float32), 1)
        for i in range(n - 1):
            up[i, i + 1] = 1
        A[b, :n, :n] = up + up.T
    flags = (A.sum(-1) > 1e-5).astype(np.float32)
    return A, flags[:, :, None] * flags[:, None, :]


def oracle_grads(kind, W, A, mask, sigma, L, H, R, noise_mlp):
    """float64 autograd through the oracle restatement; sigma: one float (sincos) or [B]."""
    from oracle import ppgn as op
    W64 = {k: v.double().requires_grad_(True) for k, v in W.items()}
    if kind == "sincos":
        ref = op.powerful_sincos_forward(W64, A, mask, sigma, L, H, dtype=torch.float64)
    elif kind == "ppgn":
        ref = op.powerful_forward(W64, A, mask, sigma, L, noise_mlp=noise_mlp, dtype=torch.float64)
    else:
        ref = op.powerful_ind_forward(W64, A, mask, sigma, L, noise_mlp=noise_mlp, mult=kind == "ind_mult", dtype=torch.float64)
    ref = ref.squeeze(-1)
    (ref * torch.from_numpy(R).double()).sum().backward()
    return ref.detach().numpy(), {k: v.grad.numpy() for k, v in W64.items() if v.grad is not None}


def check_grads(model, gref, tol=3e-3):
    named = dict(model.named_parameters())
    names = model.trainable_param_names()
    assert set(names) == set(gref), sorted(set(names) ^ set(gref))     # exactly the parameters the reference trains
    gmax = max(np.abs(g).max() for g in gref.values())
    for n in names:
        g = named[n].grad
        assert g is not None, n
        scale = max(np.abs(gref[n]).max(), 2e-2 * gmax)
        err = np.abs(g.cpu().numpy().astype(np.float64) - gref[n]).max()
        assert err <= tol * scale, f"{n}: {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("kind,noise_mlp,H,per_graph_sigma", [("ind", False, 16, True), ("ind", True, 16, True), ("ind_mult", True, 32, True),
                                                              ("ind", True, 128, True), ("sincos", True, 16, False), ("sincos", True, 128, False),
                                                              ("ppgn", True, 16, True), ("ppgn", True, 128, True)])
def test_variant_gradients_vs_oracle(kind, noise_mlp, H, per_graph_sigma):
    """Training path of the conditioning variants (and of Powerful with a learned noise_mlp): gradients of sum(logits * R) for every
    trained parameter - including the embedding layers behind d/d layer_cond and d/d sigma of dppm_ppgn_backward - against float64
    autograd through the oracle.  Tolerance as in test_forward_backward_vs_oracle (LeakyReLU kink flips, small cases)."""
    L, N, B = 2, 14, 2
    torch.manual_seed(3 + H)
    if kind == "ppgn":
        from discrete_dppm_graphs_amd.model.ppgn import Powerful
        model = Powerful(num_layers=L, input_features=2, hidden=H, hidden_final=H, n_nodes=N, normalization="instance", adj_out=True,
                         noise_mlp=True, dropout_p=1e-6)
    else:
        model = build(kind, L, H, N, noise_mlp)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("times.") or n.startswith("time_mlp."):
                p.mul_(2.0)
    W = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(17)
    A, mask = rand_graphs(rng, B, N)
    R = rng.normal(size=A.shape).astype(np.float32) * mask
    sigma = np.asarray([0.12, 0.37]) if per_graph_sigma else 0.21
    ref, gref = oracle_grads(kind, W, A, mask, sigma, L, H, R, noise_mlp)
    model = model.cuda().train()
    out = model(None, torch.from_numpy(A).cuda(), torch.from_numpy(mask).cuda(), torch.as_tensor(sigma) if per_graph_sigma else sigma)
    close(out.detach().squeeze(-1).cpu().numpy(), ref, 2e-4)
    (out.squeeze(-1) * torch.from_numpy(R).cuda()).sum().backward()
    check_grads(model, gref)


def test_sincos_mixed_sigma_batch():
    """Powerful_sincos with different noise levels in one batch (how the engine's train step calls it) = per-graph calls."""
    L, H, N, B = 2, 16, 12, 3
    torch.manual_seed(5)
    model = build("sincos", L, H, N, True).cuda().train()
    rng = np.random.default_rng(2)
    A, mask = rand_graphs(rng, B, N)
    A, mask = torch.from_numpy(A).cuda(), torch.from_numpy(mask).cuda()
    sig = [0.1, 0.1, 0.4]
    out = model(None, A, mask, torch.tensor(sig))
    for b in range(B):
        with torch.no_grad():
            model.precision = "f32"
            one = model(None, A[b:b + 1], mask[b:b + 1], sig[b])
        close(out[b:b + 1].detach().cpu().numpy(), one.cpu().numpy().astype(np.float64), 1e-5)


@pytest.mark.parametrize("kind", ["ind", "sincos", "ppgn_noise_mlp"])
def test_engine_step_trains_variant(kind):
    """engine.train_step_simple (flat bucket + fused Adam) drives the variants through forward_backward: one step equals
    torch.optim.Adam on the autograd gradients of the same batch, embedding layers included."""
    from discrete_dppm_graphs_amd import engine, ops
    L, H, N, B = 2, 16, 12, 4
    torch.manual_seed(9)
    if kind == "ppgn_noise_mlp":
        from discrete_dppm_graphs_amd.model.ppgn import Powerful
        mk = lambda: Powerful(num_layers=L, input_features=2, hidden=H, hidden_final=H, n_nodes=N, normalization="instance", adj_out=True,
                              noise_mlp=True, dropout_p=1e-6)
    else:
        mk = lambda: build(kind, L, H, N, True)
    model = mk().cuda().train()
    twin = mk().cuda().train()
    twin.load_state_dict(model.state_dict())
    rng = np.random.default_rng(4)
    A, _ = rand_graphs(rng, B, N)
    adj = torch.from_numpy(A).cuda()
    sigmas = rng.uniform(0.05, 0.5, size=B)
    state = engine.FlatState(model, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    loss = engine.train_step_simple(model, state, adj, sigmas, seed=7, offset=0)
    # the same step by hand on the twin
    flags = engine.node_flags(adj)
    A_t, target = ops.edgeflip(adj, flags, sigmas, seed=7, graph_id0=0, offset=0)
    mask = ops.generate_mask(flags)
    named = dict(twin.named_parameters())
    params = [named[n] for n in twin.trainable_param_names()]
    opt = torch.optim.Adam(params, lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    logits = twin(None, A_t, mask, torch.as_tensor(sigmas))[..., 0]
    l2, dlogits = ops.loss_bce(logits.detach(), target, torch.as_tensor(sigmas, device="cuda"), mask, B_global=B)
    logits.backward(dlogits)
    opt.step()
    assert abs(float(loss) - float(l2)) <= 1e-6 * max(1.0, abs(float(l2)))
    for n in twin.trainable_param_names():
        a, b = dict(model.named_parameters())[n].detach().cpu().numpy(), named[n].detach().cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-4, err_msg=n)    # one Adam step moves each entry by <= lr = 1e-2
    untouched = [n for n, _ in model.named_parameters() if n not in set(model.trainable_param_names())]
    assert all(n.startswith("feature_extractors.") or n.startswith("time_mlp.") or n.startswith("times.0.") for n in untouched), untouched


def test_factory_builds_variants():
    from discrete_dppm_graphs_amd.utils.easydict import EasyDict
    from discrete_dppm_graphs_amd.utils.loading_utils import get_score_model
    for name, cls in (("ppgn_ind", "Powerful_Ind"), ("ppgn_ind_mult", "Powerful_Ind_Mult"), ("ppgn_ind_sincos", "Powerful_sincos")):
        cfg = EasyDict(dict(model=dict(name=name), num_layers=2, hidden=16, hidden_final=16, dropout=1e-6, n_nodes=10,
                            normalization="instance", noise_mlp=True, dev=torch.device("cuda")))
        m = get_score_model(cfg)
        assert type(m).__name__ == cls and next(m.parameters()).is_cuda
