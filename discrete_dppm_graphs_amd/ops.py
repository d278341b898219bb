"""Torch-tensor front end of the C-ABI (include/dppm_hip.h).  torch is used only as the owner of device memory and
of the current HIP stream; every function validates its operands on the host (shape, dtype, device, contiguity)
before a raw pointer reaches a kernel, and raises RuntimeError on a non-zero status.  No CPU path exists."""
import contextlib
import ctypes as C

import numpy as np
import torch

from ._lib import PRECISIONS, PpgnLayer, PpgnParams, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, shape=None, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the DDPM hot path runs on the GPU only (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _rng_args(U, seed, graph_id0, offset, shape):
    if U is not None:
        _chk(U, "U", shape)
    return _ptr(U), C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint32(int(graph_id0)), C.c_uint32(int(offset))


def _f32c(t):
    return t.to(torch.float32).contiguous()


def _prec(precision):
    """'f32' | 'bf16x3' | 'bf16' | 'f32x6' -> DPPM_PREC_* (include/dppm_hip.h)"""
    try:
        return PRECISIONS[precision]
    except KeyError:
        raise ValueError(f"precision {precision!r}: expected one of {sorted(PRECISIONS)}") from None


def generate_mask(flags):
    flags = _chk(_f32c(flags), "flags")
    B, N = flags.shape
    mask = torch.empty(B, N, N, device=flags.device, dtype=torch.float32)
    check(lib.dppm_generate_mask(_ptr(flags), B, N, _ptr(mask), _stream()), "dppm_generate_mask")
    return mask


def _sigma_args(sigma, B, device):
    """python float -> scalar path; sequence / array / tensor of B doubles -> device array."""
    if isinstance(sigma, (float, int, np.floating)):
        return None, float(sigma)
    s = torch.as_tensor(np.asarray(sigma, dtype=np.float64) if not isinstance(sigma, torch.Tensor) else sigma)
    s = s.to(device=device, dtype=torch.float64).contiguous().view(-1)
    if s.numel() != B:
        raise ValueError(f"sigma: {s.numel()} values for a batch of {B} graphs")
    return s, 0.0


def edgeflip(A0, flags, sigma, U=None, seed=0, graph_id0=0, offset=0, want_target=True, out=None):
    """q(A_t|A_0): returns (A_t, target).  sigma: python float (all graphs) or B values (one per graph)."""
    A0 = _chk(A0, "A0")
    B, N, _ = A0.shape
    flags = _chk(flags, "flags", (B, N))
    sig_dev, sig_scalar = _sigma_args(sigma, B, A0.device)
    A_t = out if out is not None else torch.empty_like(A0)
    _chk(A_t, "out", (B, N, N))
    target = torch.empty_like(A0) if want_target else None
    u, sd, g0, off = _rng_args(U, seed, graph_id0, offset, (B, N, N))
    check(lib.dppm_edgeflip_fwd(_ptr(A0), _ptr(flags), _ptr(sig_dev), C.c_double(sig_scalar), u, sd, g0, off, B, N,
                                _ptr(A_t), _ptr(target), _stream()), "dppm_edgeflip_fwd")
    return A_t, target


def init_xT(flags, U=None, seed=0, graph_id0=0, offset=0):
    flags = _chk(flags, "flags")
    B, N = flags.shape
    out = torch.empty(B, N, N, device=flags.device, dtype=torch.float32)
    u, sd, g0, off = _rng_args(U, seed, graph_id0, offset, (B, N, N))
    check(lib.dppm_init_xT(_ptr(flags), u, sd, g0, off, B, N, _ptr(out), _stream()), "dppm_init_xT")
    return out


def reverse_step_simple(A_t, logits, flags, U=None, seed=0, graph_id0=0, offset=0, is_probs=False, out=None):
    A_t = _chk(A_t, "A_t")
    B, N, _ = A_t.shape
    logits = _chk(logits, "logits", (B, N, N))
    flags = _chk(flags, "flags", (B, N))
    out = out if out is not None else torch.empty_like(A_t)
    _chk(out, "out", (B, N, N))
    u, sd, g0, off = _rng_args(U, seed, graph_id0, offset, (B, N, N))
    check(lib.dppm_reverse_step_simple(_ptr(A_t), _ptr(logits), int(bool(is_probs)), _ptr(flags), u, sd, g0, off, B, N,
                                       _ptr(out), _stream()), "dppm_reverse_step_simple")
    return out


def reverse_step_vlb(A_t, logits, flags, sigmatilde_t, beta_t, sigmatilde_t1, U=None, seed=0, graph_id0=0, offset=0,
                     is_probs=False, out=None):
    A_t = _chk(A_t, "A_t")
    B, N, _ = A_t.shape
    logits = _chk(logits, "logits", (B, N, N))
    flags = _chk(flags, "flags", (B, N))
    out = out if out is not None else torch.empty_like(A_t)
    _chk(out, "out", (B, N, N))
    u, sd, g0, off = _rng_args(U, seed, graph_id0, offset, (B, N, N))
    check(lib.dppm_reverse_step_vlb(_ptr(A_t), _ptr(logits), int(bool(is_probs)), _ptr(flags), C.c_double(sigmatilde_t),
                                    C.c_double(beta_t), C.c_double(sigmatilde_t1), u, sd, g0, off, B, N, _ptr(out),
                                    _stream()), "dppm_reverse_step_vlb")
    return out


def _loss_ws(B, N, device):
    n = lib.dppm_loss_workspace_bytes(B, N)
    return torch.empty(max(n, 8) // 8, device=device, dtype=torch.float64), n


def loss_bce(score, target, sigmas, mask, B_global=None, want_grad=True):
    """returns (loss [float64 0-d tensor on device], dscore or None)."""
    score = _chk(score, "score")
    B, N, _ = score.shape
    target = _chk(target, "target", (B, N, N))
    mask = _chk(mask, "mask", (B, N, N))
    sig_dev, _ = _sigma_args(list(sigmas) if not isinstance(sigmas, torch.Tensor) else sigmas, B, score.device)
    ws, nbytes = _loss_ws(B, N, score.device)
    loss = torch.empty((), device=score.device, dtype=torch.float64)
    dscore = torch.empty_like(score) if want_grad else None
    check(lib.dppm_loss_bce_fwd_bwd(_ptr(score), _ptr(target), _ptr(sig_dev), _ptr(mask), B, N, int(B_global or B), _ptr(loss),
                                    _ptr(dscore), _ptr(ws), C.c_size_t(nbytes), _stream()), "dppm_loss_bce_fwd_bwd")
    return loss, dscore


def loss_vlb(score, A_t, A_0, sched, mask, want_grad=True):
    """sched: [B,3] doubles (sigmatilde_t, beta_t, sigmatilde_{t-1}).  returns (loss float32 0-d, dscore or None)."""
    score = _chk(score, "score")
    B, N, _ = score.shape
    A_t = _chk(A_t, "A_t", (B, N, N))
    A_0 = _chk(A_0, "A_0", (B, N, N))
    mask = _chk(mask, "mask", (B, N, N))
    sched = torch.as_tensor(np.asarray(sched, dtype=np.float64) if not isinstance(sched, torch.Tensor) else sched)
    sched = _chk(sched.to(device=score.device, dtype=torch.float64).contiguous(), "sched", (B, 3), torch.float64)
    ws, nbytes = _loss_ws(B, N, score.device)
    loss = torch.empty((), device=score.device, dtype=torch.float32)
    dscore = torch.empty_like(score) if want_grad else None
    check(lib.dppm_loss_vlb_fwd_bwd(_ptr(score), _ptr(A_t), _ptr(A_0), _ptr(sched), _ptr(mask), B, N, _ptr(loss), _ptr(dscore),
                                    _ptr(ws), C.c_size_t(nbytes), _stream()), "dppm_loss_vlb_fwd_bwd")
    return loss, dscore


# ------------------------------------------------------------------------------------------------- PPGN
LAYER_FIELDS = (("m1_w0", "m1.0.weight"), ("m1_b0", "m1.0.bias"), ("m1_w1", "m1.2.weight"), ("m1_b1", "m1.2.bias"),
                ("m2_w0", "m2.0.weight"), ("m2_b0", "m2.0.bias"), ("m2_w1", "m2.2.weight"), ("m2_b1", "m2.2.bias"),
                ("m4_w", "m4.0.weight"), ("m4_b", "m4.0.bias"))
TOP_FIELDS = (("in_w", "in_lin.0.weight"), ("in_b", "in_lin.0.bias"), ("cat_w", "layer_cat_lin.0.weight"),
              ("cat_b", "layer_cat_lin.0.bias"), ("after_w", "after_conv.0.weight"), ("after_b", "after_conv.0.bias"),
              ("final_w", "final_lin.0.weight"), ("final_b", "final_lin.0.bias"))


def used_param_names(num_layers):
    names = [n for _, n in TOP_FIELDS[:2]]
    for l in range(num_layers):
        names += [f"convs.{l}.{n}" for _, n in LAYER_FIELDS]
    names += [n for _, n in TOP_FIELDS[2:]]
    return names


def expected_shapes(num_layers, H):
    sh = {"in_lin.0.weight": (H, 2), "in_lin.0.bias": (H,), "layer_cat_lin.0.weight": (H, num_layers * H + 2),
          "layer_cat_lin.0.bias": (H,), "after_conv.0.weight": (H, H), "after_conv.0.bias": (H,),
          "final_lin.0.weight": (1, H), "final_lin.0.bias": (1,)}
    for l in range(num_layers):
        for m in ("m1", "m2"):
            for k in ("0", "2"):
                sh[f"convs.{l}.{m}.{k}.weight"] = (H, H)
                sh[f"convs.{l}.{m}.{k}.bias"] = (H,)
        sh[f"convs.{l}.m4.0.weight"] = (H, 2 * H)
        sh[f"convs.{l}.m4.0.bias"] = (H,)
    return sh


class ParamPack:
    """Host-side dppm_ppgn_params built from a {state_dict name: cuda tensor} mapping.  Keeps the tensors alive."""

    def __init__(self, tensors, num_layers, hidden, layer_cond=None):
        shapes = expected_shapes(num_layers, hidden)
        self.tensors = {}
        for name, shape in shapes.items():
            if name not in tensors:
                raise KeyError(f"missing PPGN parameter {name}")
            self.tensors[name] = _chk(tensors[name], name, shape)
        self.layers = (PpgnLayer * num_layers)()
        for l in range(num_layers):
            for field, suffix in LAYER_FIELDS:
                setattr(self.layers[l], field, self.tensors[f"convs.{l}.{suffix}"].data_ptr())
        self.struct = PpgnParams()
        self.struct.num_layers = num_layers
        self.struct.hidden = hidden
        for field, name in TOP_FIELDS:
            setattr(self.struct, field, self.tensors[name].data_ptr())
        self.struct.layers_host = C.cast(self.layers, C.POINTER(PpgnLayer))
        self.layer_cond = None
        self.set_layer_cond(layer_cond)

    def set_layer_cond(self, layer_cond):
        """[L, B] float32 on the device (Powerful_Ind conditioning, include/dppm_hip.h) or None."""
        if layer_cond is not None:
            if layer_cond.dim() != 2 or layer_cond.shape[0] != self.struct.num_layers:
                raise ValueError(f"layer_cond: shape {tuple(layer_cond.shape)}, expected [{self.struct.num_layers}, B]")
            _chk(layer_cond, "layer_cond")
        self.layer_cond = layer_cond
        self.struct.layer_cond = layer_cond.data_ptr() if layer_cond is not None else None

    def ref(self):
        return C.byref(self.struct)


def ppgn_workspace(B, N, H, L, training, device):
    nbytes = lib.dppm_ppgn_workspace_bytes(B, N, H, L, int(bool(training)))
    if nbytes == 0:
        raise ValueError(f"invalid PPGN geometry B={B} N={N} H={H} L={L}")
    return torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)


def ppgn_forward(pack, A, mask, sigma, workspace, training=False, out=None, precision="f32"):
    A = _chk(A, "A")
    B, N, _ = A.shape
    mask = _chk(mask, "mask", (B, N, N))
    sigma = _chk(sigma, "sigma", (B,))
    if pack.layer_cond is not None and pack.layer_cond.shape[1] != B:
        raise ValueError(f"layer_cond holds {pack.layer_cond.shape[1]} graphs, the batch has {B}")
    _chk(workspace, "workspace")
    logits = out if out is not None else torch.empty(B, N, N, device=A.device, dtype=torch.float32)
    _chk(logits, "out", (B, N, N))
    check(lib.dppm_ppgn_forward(pack.ref(), _ptr(A), _ptr(mask), _ptr(sigma), B, N, _ptr(logits), _ptr(workspace),
                                C.c_size_t(workspace.numel() * 4), int(bool(training)), _prec(precision), _stream()),
          "dppm_ppgn_forward")
    return logits


def chain_supported(pack, N):
    """Whether a whole sampling chain of this network at N nodes runs as one launch (hidden 32, N <= 20: include/dppm_hip.h)."""
    return bool(lib.dppm_ppgn_chain_supported(pack.ref(), int(N)))


def _chain(fn, name, width, pack, flags, mask, level_table, seed, graph_id0, out, precision):
    flags = _chk(flags, "flags")
    B, N = flags.shape
    mask = _chk(mask, "mask", (B, N, N))
    level_table = _chk(level_table, "level_table")
    if level_table.dim() != 2 or level_table.shape[1] != width:
        raise ValueError(f"level_table must be [T, {width}], got {tuple(level_table.shape)}")
    x = out if out is not None else torch.empty(B, N, N, device=flags.device, dtype=torch.float32)
    _chk(x, "out", (B, N, N))
    check(fn(pack.ref(), _ptr(flags), _ptr(mask), _ptr(level_table), int(level_table.shape[0]), C.c_uint64(int(seed) & (2 ** 64 - 1)),
             C.c_uint32(int(graph_id0)), B, N, _ptr(x), _prec(precision), _stream()), name)
    return x


def chain_simple(pack, flags, mask, level_table, seed, graph_id0=0, out=None, precision="f32"):
    """x_T, then T x (re-noise, denoise, reverse step) in one launch; level_table [T, 3] float32 on the device
    (engine.level_table_simple).  Bit-identical to init_xT / edgeflip / ppgn_forward / reverse_step_simple level by level."""
    return _chain(lib.dppm_ppgn_chain_simple, "dppm_ppgn_chain_simple", 3, pack, flags, mask, level_table, seed, graph_id0, out, precision)


def chain_vlb(pack, flags, mask, level_table, seed, graph_id0=0, out=None, precision="f32"):
    """x_T, then T x (denoise, posterior reverse step) in one launch; level_table [T, 7] float32 on the device
    (engine.level_table_vlb).  Bit-identical to init_xT / ppgn_forward / reverse_step_vlb level by level."""
    return _chain(lib.dppm_ppgn_chain_vlb, "dppm_ppgn_chain_vlb", 7, pack, flags, mask, level_table, seed, graph_id0, out, precision)


def ppgn_backward(pack, grad_pack, A, mask, sigma, dlogits, workspace, dsigma=None):
    """Parameter gradients through `grad_pack`; dsigma ([B], optional) receives d/d sigma; when `pack` carries layer_cond,
    `grad_pack.layer_cond` ([L, B]) receives its gradient."""
    A = _chk(A, "A")
    B, N, _ = A.shape
    dlogits = _chk(dlogits, "dlogits", (B, N, N))
    if dsigma is not None:
        _chk(dsigma, "dsigma", (B,))
    if pack.layer_cond is not None and (grad_pack.layer_cond is None or grad_pack.layer_cond.shape != pack.layer_cond.shape):
        raise ValueError("ppgn_backward: grad_pack needs a layer_cond buffer of the parameters' layer_cond shape")
    check(lib.dppm_ppgn_backward(pack.ref(), grad_pack.ref(), _ptr(A), _ptr(mask), _ptr(sigma), B, N, _ptr(dlogits),
                                 _ptr(dsigma), _ptr(workspace), C.c_size_t(workspace.numel() * 4), _stream()),
          "dppm_ppgn_backward")


def plane_matmul(A, Bm, n, trans_a=False, trans_b=False, scale=None, mask=None, scale_div=1, precision="f32"):
    """A, Bm: [batch, n, ld] planes (ld = n rounded up to 4)."""
    A = _chk(A, "A")
    batch, n_, ld = A.shape
    if n_ != n:
        raise ValueError("plane_matmul: A must be [batch, n, ld]")
    Bm = _chk(Bm, "Bm", (batch, n, ld))
    out = torch.empty_like(A)
    check(lib.dppm_plane_matmul(_ptr(A), _ptr(Bm), _ptr(out), batch, n, ld, int(trans_a), int(trans_b), _ptr(scale), _ptr(mask),
                                int(scale_div), _prec(precision), _stream()), "dppm_plane_matmul")
    return out


def linear_fwd(W, X, bias=None, act=False, maskp=None, trans_w=False, out=None, accumulate=False):
    """X: [B, K, P] planes; W: [M, K] (or [K, M] with trans_w) -> Y [B, M, P]."""
    X = _chk(X, "X")
    B, K, P = X.shape
    if W.dim() != 2 or not W.is_cuda or W.dtype != torch.float32 or W.stride(1) != 1:
        raise ValueError("W: expected a 2-d float32 cuda tensor with unit column stride")
    M = W.shape[1] if trans_w else W.shape[0]
    if (W.shape[0] if trans_w else W.shape[1]) != K:
        raise ValueError(f"W: shape {tuple(W.shape)} does not match K={K}")
    Y = out if out is not None else torch.empty(B, M, P, device=X.device, dtype=torch.float32)
    _chk(Y, "out", (B, M, P))
    check(lib.dppm_linear_fwd(_ptr(W), int(W.stride(0)), int(trans_w), _ptr(X), C.c_long(K * P), K, _ptr(bias), int(act),
                              _ptr(maskp), _ptr(Y), C.c_long(M * P), M, B, P, int(accumulate), _stream()), "dppm_linear_fwd")
    return Y


def mlp_pair_fwd(X, weights, maskp=None, save_hidden=False, precision="f32"):
    """X: [B, H, P]; weights: dict with the ten `m1.0.weight` ... `m4.0.bias`-style keys of one layer (m4 unused).
    -> (o1, o2, hid1 or None, hid2 or None)"""
    X = _chk(X, "X")
    B, H, P = X.shape
    layer = PpgnLayer()
    keep = []
    for field, suffix in LAYER_FIELDS:
        if field.startswith("m4"):
            continue
        t = _chk(weights[suffix], suffix, (H, H) if suffix.endswith("weight") else (H,))
        keep.append(t)
        setattr(layer, field, t.data_ptr())
    o1, o2 = torch.empty_like(X), torch.empty_like(X)
    need_hid = save_hidden or H != 128
    h1 = torch.empty_like(X) if need_hid else None
    h2 = torch.empty_like(X) if need_hid else None
    check(lib.dppm_mlp_pair_fwd(_ptr(X), C.c_long(H * P), C.byref(layer), H, _ptr(maskp), _ptr(o1), _ptr(o2), _ptr(h1), _ptr(h2), B, P,
                                _prec(precision), _stream()), "dppm_mlp_pair_fwd")
    return o1, o2, h1, h2


def linear_wgrad(dY, X, want_bias=True):
    """dY: [B, M, P], X: [B, Nw, P] -> (dW [M, Nw], db [M] or None)."""
    dY = _chk(dY, "dY")
    B, M, P = dY.shape
    X = _chk(X, "X")
    Nw = X.shape[1]
    if X.shape[0] != B or X.shape[2] != P:
        raise ValueError("X: batch / plane size mismatch")
    nbytes = lib.dppm_linear_wgrad_workspace_bytes(M, Nw, B, P)
    ws = torch.empty((nbytes + 3) // 4, device=dY.device, dtype=torch.float32)
    dW = torch.empty(M, Nw, device=dY.device, dtype=torch.float32)
    db = torch.empty(M, device=dY.device, dtype=torch.float32) if want_bias else None
    check(lib.dppm_linear_wgrad(_ptr(dY), C.c_long(M * P), _ptr(X), C.c_long(Nw * P), M, Nw, B, P, _ptr(dW), Nw, _ptr(db), _ptr(ws),
                                C.c_size_t(nbytes), _stream()), "dppm_linear_wgrad")
    return dW, db


def inorm_fwd(y, maskp):
    """y: [B, C, P], maskp: [B, P] -> (u, mean [B,C], rstd [B,C])."""
    y = _chk(y, "y")
    B, Cc, P = y.shape
    maskp = _chk(maskp, "maskp", (B, P))
    u = torch.empty_like(y)
    mean = torch.empty(B, Cc, device=y.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    check(lib.dppm_inorm_fwd(_ptr(y), C.c_long(Cc * P), _ptr(maskp), _ptr(u), C.c_long(Cc * P), B, Cc, P, _ptr(mean), _ptr(rstd),
                             _stream()), "dppm_inorm_fwd")
    return u, mean, rstd


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, n, (param.numel(),))
    check(lib.dppm_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), C.c_size_t(param.numel()), lr, beta1,
                             beta2, eps, weight_decay, int(step), grad_scale, _stream()), "dppm_adam_step")


# ------------------------------------------------------------------------------------------------- graph statistics
def graph_stats(adj, bins=100):
    """adj [B, N, N] on the GPU -> (deg_hist [B, N], clus_hist [B, bins]) float32 counts (include/dppm_hip.h)."""
    adj = _chk(_f32c(adj), "adj")
    B, N, N2 = adj.shape
    if N != N2:
        raise ValueError(f"adj must be [B, N, N], got {tuple(adj.shape)}")
    edges = torch.as_tensor(np.linspace(0.0, 1.0, bins + 1), device=adj.device)      # numpy.histogram's float64 edges
    deg = torch.empty(B, N, device=adj.device, dtype=torch.float32)
    clus = torch.empty(B, bins, device=adj.device, dtype=torch.float32)
    check(lib.dppm_graph_stats(_ptr(adj), B, N, int(bins), _ptr(edges), _ptr(deg), _ptr(clus), _stream()), "dppm_graph_stats")
    return deg, clus


def mmd_gaussian_emd(h1, h2, sigma=1.0, distance_scaling=1.0):
    """h1 [n1, bins], h2 [n2, bins] (counts) -> float64 tensor {mmd, k11, k22, k12} on the device."""
    h1 = _chk(_f32c(h1), "h1")
    h2 = _chk(_f32c(h2), "h2")
    if h1.dim() != 2 or h2.dim() != 2 or h1.shape[1] != h2.shape[1]:
        raise ValueError("mmd_gaussian_emd: histograms must be [n, bins] with equal bins")
    n1, bins = h1.shape
    n2 = h2.shape[0]
    nbytes = lib.dppm_mmd_workspace_bytes(n1, n2)
    ws = torch.empty(max(nbytes, 8) // 8, device=h1.device, dtype=torch.float64)
    out = torch.empty(4, device=h1.device, dtype=torch.float64)
    check(lib.dppm_mmd_gaussian_emd(_ptr(h1), n1, _ptr(h2), n2, bins, C.c_double(sigma), C.c_double(distance_scaling), _ptr(out),
                                    _ptr(ws), C.c_size_t(nbytes), _stream()), "dppm_mmd_gaussian_emd")
    return out


PROF_CLASSES = ("linear_gemm", "plane_matmul", "wgrad_gemm", "inorm", "wstat_mlp2", "wstat_sum2", "wstat_cols", "head")


def prof_begin(max_records=200000):
    check(lib.dppm_prof_begin(int(max_records)), "dppm_prof_begin")


PROF_ARITH = ("f32", "bf16x3", "bf16", "f32x6", "none")   # the arithmetic a timed launch ran in (include/dppm_hip_debug.h)


def prof_end():
    """-> {(class, arith): {"ms": device time, "work": FLOPs (bytes for inorm), "bytes": algorithmic HBM bytes, "launches": n}} for
    every (kernel class, arithmetic) pair that saw a launch, and the truncated flag."""
    nc, na = len(PROF_CLASSES), len(PROF_ARITH)
    n = nc * na
    ms, work, nbytes, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_long * n)()
    rc = lib.dppm_prof_end(ms, work, nbytes, cnt, nc, na)
    if rc < 0:
        check(rc, "dppm_prof_end")
    out = {}
    for i, k in enumerate(PROF_CLASSES):
        for j, a in enumerate(PROF_ARITH):
            if cnt[i * na + j]:
                out[(k, a)] = {"ms": ms[i * na + j], "work": work[i * na + j], "bytes": nbytes[i * na + j], "launches": cnt[i * na + j]}
    return out, rc == 1


def debug_set(name, value):
    """Experiment switch of the library (include/dppm_hip_debug.h); returns the previous value."""
    old = C.c_int(0)
    check(lib.dppm_debug_get(name.encode(), C.byref(old)), "dppm_debug_get")
    check(lib.dppm_debug_set(name.encode(), int(value)), "dppm_debug_set")
    return old.value


@contextlib.contextmanager
def debug_switch(**switches):
    """with ops.debug_switch(DISABLE_HEAD3=1): ...   -- tests and A/B measurements only."""
    old = {k: debug_set(k, v) for k, v in switches.items()}
    try:
        yield
    finally:
        for k, v in old.items():
            debug_set(k, v)
