"""CPU restatement of the PPGN ("Powerful") denoiser (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional torch-CPU float32 port of reference model/ppgn.py:257-337 (Powerful.forward_cat, the configuration
get_score_model builds: input_features=2, cat_output, adj_out, not simplified, no residual, not project_first)
with PowerfulLayer.forward (:47-71) and masked_instance_norm2D (util/model_helper.py:176-191).  Differences
from the reference are only organisational: weights come in as a dict keyed by the reference's state_dict
names, the batch is processed in one call, and `noiselevel` may be a per-graph vector (the reference takes
one python float and is called once per graph by fit, ppgn_simple.py:84-88).  Autograd provides gradients.
Pinned against tests/golden/g1_ppgn_*.npz (logits, loss, d/dparams produced by the reference itself).
"""
import numpy as np
import torch
import torch.nn.functional as F

SLOPE = 0.01


def _lin(x, w, b=None):
    return F.linear(x, w, b)


def masked_instance_norm2d(x, mask, eps=1e-5):
    """util/model_helper.py:176-191.  x [B,N,N,C], mask [B,N,N,1]."""
    mask = mask.expand_as(x)
    cnt = torch.sum(mask, dim=[1, 2])
    mean = torch.sum(x * mask, dim=[1, 2]) / cnt
    var_term = ((x - mean[:, None, None, :]) * mask) ** 2
    var = torch.sum(var_term, dim=[1, 2]) / cnt
    out = (x - mean[:, None, None, :]) / torch.sqrt(var[:, None, None, :] + eps)
    return out * mask


def powerful_layer(x, mask4, W, l):
    """model/ppgn.py:47-71.  x [B,N,N,H]; mask4 [B,N,N,1]."""
    pre = f"convs.{l}."
    norm = mask4[:, 0].squeeze(-1).sum(-1).sqrt().view(mask4.size(0), 1, 1, 1)
    m = mask4.unsqueeze(1).squeeze(-1)   # B,1,N,N

    def mlp(name):
        h = _lin(x, W[pre + name + ".0.weight"], W[pre + name + ".0.bias"])
        h = F.leaky_relu(h, SLOPE)
        return _lin(h, W[pre + name + ".2.weight"], W[pre + name + ".2.bias"])

    out1 = mlp("m1").permute(0, 3, 1, 2) * m
    out2 = mlp("m2").permute(0, 3, 1, 2) * m
    out = (out1 @ out2) / norm
    out = torch.cat((out.permute(0, 2, 3, 1), x), dim=3)
    return _lin(out, W[pre + "m4.0.weight"], W[pre + "m4.0.bias"])


def powerful_forward(W, A, mask, noiselevel, num_layers, noise_mlp=False, return_intermediates=False, dtype=torch.float32):
    """W: dict name -> tensor (reference state_dict names); A, mask [B,N,N]; noiselevel float or [B].
    Returns logits [B,N,N,1] (model/ppgn.py:257-337).  dtype=torch.float64 (with float64 weights) evaluates the same graph in
    double precision: the yardstick the tests use to separate rounding noise from errors."""
    A = torch.as_tensor(A, dtype=dtype)
    mask = torch.as_tensor(mask, dtype=dtype)
    B, N, _ = A.shape
    mask4 = mask[..., None]
    u = A[..., None]
    sig = torch.as_tensor(np.asarray(noiselevel, dtype=np.float64).reshape(-1).astype(np.float32)).to(dtype)
    if sig.numel() == 1:
        sig = sig.expand(B)
    if noise_mlp:   # ppgn.py:273-274: Linear(1,4) -> GELU -> Linear(4,1) on the scalar
        s = sig.view(B, 1)
        s = _lin(F.gelu(_lin(s, W["time_mlp.0.weight"], W["time_mlp.0.bias"])),
                 W["time_mlp.2.weight"], W["time_mlp.2.bias"])
        sig = s.view(B)
    # ppgn.py:283-293: sigma on the diagonal of a second input channel (padded diagonal entries included)
    nl = torch.diag_embed(sig.view(B, 1).expand(B, N))[..., None]
    u = torch.cat([u, nl], dim=-1)
    outs = [u]
    u = _lin(u, W["in_lin.0.weight"], W["in_lin.0.bias"])
    inter = []
    for l in range(num_layers):
        u = powerful_layer(u, mask4, W, l)
        u = masked_instance_norm2d(u, mask4)
        u = u * mask4
        outs.append(u)
        inter.append(u)
    out = torch.cat(outs, dim=-1)
    out = _lin(out, W["layer_cat_lin.0.weight"], W["layer_cat_lin.0.bias"])
    out = out + F.leaky_relu(_lin(out, W["after_conv.0.weight"], W["after_conv.0.bias"]), SLOPE)
    # F.dropout(p=1e-6) is the identity in eval(); parity is asserted in eval()
    out = _lin(out, W["final_lin.0.weight"], W["final_lin.0.bias"])
    if return_intermediates:
        return out, inter
    return out


def used_param_names(num_layers):
    names = ["in_lin.0.weight", "in_lin.0.bias"]
    for l in range(num_layers):
        for m in ("m1", "m2"):
            for k in ("0", "2"):
                names += [f"convs.{l}.{m}.{k}.weight", f"convs.{l}.{m}.{k}.bias"]
        names += [f"convs.{l}.m4.0.weight", f"convs.{l}.m4.0.bias"]
    names += ["layer_cat_lin.0.weight", "layer_cat_lin.0.bias", "after_conv.0.weight", "after_conv.0.bias",
              "final_lin.0.weight", "final_lin.0.bias"]
    return names


def init_weights(num_layers, hidden, seed, noise_mlp=False):
    """Random weights with the reference's shapes and torch's default nn.Linear init scale."""
    g = torch.Generator().manual_seed(seed)
    H = hidden

    def lin(o, i):
        bound = 1.0 / np.sqrt(i)
        return ((torch.rand(o, i, generator=g) * 2 - 1) * bound, (torch.rand(o, generator=g) * 2 - 1) * bound)

    W = {}
    W["in_lin.0.weight"], W["in_lin.0.bias"] = lin(H, 2)
    for l in range(num_layers):
        for m in ("m1", "m2"):
            W[f"convs.{l}.{m}.0.weight"], W[f"convs.{l}.{m}.0.bias"] = lin(H, H)
            W[f"convs.{l}.{m}.2.weight"], W[f"convs.{l}.{m}.2.bias"] = lin(H, H)
        W[f"convs.{l}.m4.0.weight"], W[f"convs.{l}.m4.0.bias"] = lin(H, 2 * H)
    W["layer_cat_lin.0.weight"], W["layer_cat_lin.0.bias"] = lin(H, num_layers * H + 2)
    W["after_conv.0.weight"], W["after_conv.0.bias"] = lin(H, H)
    W["final_lin.0.weight"], W["final_lin.0.bias"] = lin(1, H)
    if noise_mlp:
        W["time_mlp.0.weight"], W["time_mlp.0.bias"] = lin(4, 1)
        W["time_mlp.2.weight"], W["time_mlp.2.bias"] = lin(1, 4)
    return W


# ---------------------------------------------------------------------------------------------------------
# Noise-conditioning variants.  Pinned against tests/golden/g6_variant_*.npz, produced by the
# reference's own Powerful_Ind / Powerful_Ind_Mult / Powerful_sincos (tests/golden/gen_golden_variants.py).
# ---------------------------------------------------------------------------------------------------------
def _sig_vec(noiselevel, B, dtype):
    sig = torch.as_tensor(np.asarray(noiselevel, dtype=np.float64).reshape(-1).astype(np.float32)).to(dtype)
    return sig.expand(B) if sig.numel() == 1 else sig


def _readout(W, outs):
    out = torch.cat(outs, dim=-1)
    out = _lin(out, W["layer_cat_lin.0.weight"], W["layer_cat_lin.0.bias"])
    out = out + F.leaky_relu(_lin(out, W["after_conv.0.weight"], W["after_conv.0.bias"]), SLOPE)
    return _lin(out, W["final_lin.0.weight"], W["final_lin.0.bias"])


def powerful_ind_forward(W, A, mask, noiselevel, num_layers, noise_mlp=False, mult=False, dtype=torch.float32):
    """model/ppgn_indnoise.py:271-351 (Powerful_Ind) and model/ppgn_indnoise_multdim.py:169-244 (Powerful_Ind_Mult, mult=True:
    its time_mlp ends in a second GELU, :112-117).  Trunk width hidden-1; before every layer the learned scalar
    times[l+1](noise) goes on the diagonal of an extra channel (padded diagonal entries included)."""
    A = torch.as_tensor(A, dtype=dtype)
    mask = torch.as_tensor(mask, dtype=dtype)
    B, N, _ = A.shape
    mask4 = mask[..., None]
    cond = _sig_vec(noiselevel, B, dtype).view(B, 1)

    def diag(v):   # [B,1] -> [B,N,N,1] with v_b on the whole diagonal
        return torch.diag_embed(v.expand(B, N))[..., None]

    if noise_mlp:   # ppgn_indnoise.py:279-287
        cond = _lin(F.gelu(_lin(cond, W["time_mlp.0.weight"], W["time_mlp.0.bias"])), W["time_mlp.2.weight"], W["time_mlp.2.bias"])
        if mult:
            cond = F.gelu(cond)
        nin = _lin(cond, W["times.0.0.weight"], W["times.0.0.bias"])
    else:
        nin = cond
    u = torch.cat([A[..., None], diag(nin)], dim=-1)
    outs = [u]
    u = _lin(u, W["in_lin.0.weight"], W["in_lin.0.bias"])
    for l in range(num_layers):
        nl = _lin(cond, W[f"times.{l + 1}.0.weight"], W[f"times.{l + 1}.0.bias"])
        u = torch.cat([u, diag(nl)], dim=-1)
        u = powerful_layer(u, mask4, W, l)
        u = masked_instance_norm2d(u, mask4) * mask4
        outs.append(u)
    return _readout(W, outs)


def sinusoidal_pos_emb(t, dim):
    """model/ppgn_indnoise_sincos.py:89-100."""
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=t.dtype) * -(np.log(10000) / (half - 1)))
    e = t[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def sincos_embeddings(W, noiselevel, B, num_layers, hidden, dtype=torch.float32):
    """-> [e_0 [B,1], e_1 .. e_L [B,H]]: act(times[l](time_mlp(sigma)))  (ppgn_indnoise_sincos.py:198-203, :215-217)."""
    t = _sig_vec(noiselevel, B, dtype)
    pre = sinusoidal_pos_emb(t, hidden)
    pre = F.leaky_relu(_lin(pre, W["time_mlp.1.weight"], W["time_mlp.1.bias"]), SLOPE)
    pre = F.leaky_relu(_lin(pre, W["time_mlp.3.weight"], W["time_mlp.3.bias"]), SLOPE)
    return [F.leaky_relu(_lin(pre, W[f"times.{l}.0.weight"], W[f"times.{l}.0.bias"]), SLOPE) for l in range(num_layers + 1)]


def powerful_sincos_forward(W, A, mask, noiselevel, num_layers, hidden, dtype=torch.float32):
    """model/ppgn_indnoise_sincos.py:189-262 (Powerful_sincos; the reference only runs with noise_mlp=True).  A learned
    embedding of sigma is ADDED to every channel at every position before in_lin and before each layer."""
    A = torch.as_tensor(A, dtype=dtype)
    mask = torch.as_tensor(mask, dtype=dtype)
    B, N, _ = A.shape
    mask4 = mask[..., None]
    emb = sincos_embeddings(W, noiselevel, B, num_layers, hidden, dtype)
    u = A[..., None] + emb[0][:, None, None, :]
    outs = [u]
    u = _lin(u, W["in_lin.0.weight"], W["in_lin.0.bias"])
    for l in range(num_layers):
        u = u + emb[l + 1][:, None, None, :]
        u = powerful_layer(u, mask4, W, l)
        u = masked_instance_norm2d(u, mask4) * mask4
        outs.append(u)
    return _readout(W, outs)
