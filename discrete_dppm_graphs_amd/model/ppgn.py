"""`Powerful` -- the PPGN denoiser of the reference (model/ppgn.py:124-337) as a thin nn.Module over the HIP kernels.

Drop-in contract
  * constructor keywords exactly as model/ppgn.py:125-150 (unused ones accepted and ignored);
  * `forward(node_features, A, mask, noiselevel)` in that positional order, returning float32 logits [B, N, N, 1];
  * `state_dict()` carries the reference's key names, including the parameters the configuration never executes
    (`time_mlp.*`, `feature_extractors.{l}.lin{1,2,3}.0.*`), so reference checkpoints load with strict=True or False;
  * `get_out_dim()`, `train()/eval()`.
Extensions: `noiselevel` may be a length-B sequence / tensor (one sigma per graph) so a whole training batch runs in
one call, and `forward_backward` runs forward + loss gradient + backward without autograd (used by `fit`).
`model.precision` ('f32' | 'bf16x3' | 'bf16'; config key `precision`, environment DPPM_PRECISION, default 'bf16x3')
selects the matrix arithmetic of the H = 128 kernels for inference / sampling (include/dppm_hip.h DPPM_PREC_*;
'bf16x3' holds the float32 parity bar, tests/test_gpu_ppgn.py).  Training forward + backward always run exact float32,
and tensors stay float32 in every mode.

Only the configuration that utils/loading_utils.py:98-119 builds is implemented in HIP (input_features=2, instance
norm, cat_output, adj_out, layer_after_conv, no residual / project_first / node_out); anything else raises.
All arithmetic happens in libdppm_hip.so; there is no PyTorch fallback.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops

SLOPE = 0.01


def default_precision(config=None, key="precision", env="DPPM_PRECISION", fallback="bf16x3"):
    """config.<key> if the YAML sets it, else $<env>, else `fallback`."""
    p = None
    if config is not None:
        p = config.get(key) if isinstance(config, dict) else getattr(config, key, None)
    p = p or os.environ.get(env) or fallback
    ops._prec(p)   # validates
    return p


class _Seq(nn.Sequential):
    """nn.Sequential used purely as a parameter container so state_dict keys read `<name>.0.weight` etc."""


class PowerfulLayer(nn.Module):
    """Parameter container mirroring model/ppgn.py:10-44 (m1, m2: Linear-LeakyReLU-Linear; m4: Linear(2H, H))."""

    def __init__(self, in_feat, out_feat, num_layers=2):
        super().__init__()
        self.in_feat, self.out_feat = in_feat, out_feat

        def mlp():
            mods = []
            for i in range(num_layers * 2 - 1):
                mods.append(nn.Linear(in_feat if i == 0 else out_feat, out_feat) if i % 2 == 0 else nn.LeakyReLU(SLOPE))
            return _Seq(*mods)

        self.m1 = mlp()
        self.m2 = mlp()
        self.m4 = _Seq(nn.Linear(in_feat + out_feat, out_feat, bias=True))


class FeatureExtractor(nn.Module):
    """Constructed by the reference (model/ppgn.py:75-93) but never executed when adj_out=True; kept for state_dict parity."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.lin1 = _Seq(nn.Linear(in_features, out_features, bias=True))
        self.lin2 = _Seq(nn.Linear(in_features, out_features, bias=False))
        self.lin3 = _Seq(nn.Linear(out_features, out_features, bias=False))


class _PowerfulFn(torch.autograd.Function):
    """autograd bridge: forward/backward are single C-ABI calls; the workspace filled by forward is kept for backward."""

    @staticmethod
    def forward(ctx, module, training, A, mask, sigma, *params):
        pack = module._pack()
        B, N, _ = A.shape
        ws = module._workspace(B, N, training, A.device)
        logits = ops.ppgn_forward(pack, A, mask, sigma, ws, training=training, precision=module.precision)
        if training:
            ctx.module, ctx.ws, ctx.pack = module, ws, pack
            ctx.save_for_backward(A, mask, sigma)
        return logits

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dlogits):
        A, mask, sigma = ctx.saved_tensors
        module = ctx.module
        names = module.used_param_names()
        grads = {n: torch.empty_like(p) for n, p in zip(names, module._used_params())}
        gpack = ops.ParamPack(grads, module.num_layers, module.hidden)
        ops.ppgn_backward(ctx.pack, gpack, A, mask, sigma, dlogits.contiguous(), ctx.ws)
        ctx.ws = None
        return (None, None, None, None, None) + tuple(grads[n] for n in names)


class _PackedFn(torch.autograd.Function):
    """autograd bridge over explicit kernel-layout tensors (ops.used_param_names order) that are themselves results of
    differentiable torch ops on a module's parameters (zero padding, effective biases), plus d/d sigma and d/d layer_cond.
    Every training forward owns its workspace (several forwards may precede one backward)."""

    @staticmethod
    def forward(ctx, L, H, A, mask, sigma, layer_cond, *tensors):
        names = ops.used_param_names(L)
        tens = {n: t.detach().contiguous() for n, t in zip(names, tensors)}
        cond = layer_cond.detach().contiguous() if layer_cond is not None else None
        pack = ops.ParamPack(tens, L, H, layer_cond=cond)
        B, N, _ = A.shape
        ws = ops.ppgn_workspace(B, N, H, L, True, A.device)
        sig = sigma.detach().contiguous()
        logits = ops.ppgn_forward(pack, A, mask, sig, ws, training=True)
        ctx.pack, ctx.ws, ctx.geom, ctx.names = pack, ws, (L, H), names
        ctx.save_for_backward(A, mask, sig)
        return logits

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dlogits):
        A, mask, sig = ctx.saved_tensors
        L, H = ctx.geom
        pack = ctx.pack
        grads = {n: torch.empty_like(pack.tensors[n]) for n in ctx.names}
        dcond = torch.empty_like(pack.layer_cond) if pack.layer_cond is not None else None
        dsig = torch.empty_like(sig) if ctx.needs_input_grad[4] else None
        ops.ppgn_backward(pack, ops.ParamPack(grads, L, H, layer_cond=dcond), A, mask, sig, dlogits.contiguous(), ctx.ws, dsigma=dsig)
        ctx.ws = ctx.pack = None
        return (None, None, None, None, dsig, dcond) + tuple(grads[n] for n in ctx.names)


def autograd_forward_backward(model, A, mask, noiselevel, loss_grad_fn, grad_views):
    """forward + loss gradient + backward for the engine (engine.FlatState) when part of the model is ordinary torch code
    (learned noise embeddings, padded / effective parameter tensors): the graph is model._train_logits, the parameter
    gradients are copied into `grad_views` ({name: view of the flat gradient bucket})."""
    with torch.enable_grad():
        logits = model._train_logits(A, mask, noiselevel)
    loss, dlogits = loss_grad_fn(logits.detach())
    names = list(grad_views)
    sd = dict(model.named_parameters())
    gs = torch.autograd.grad(logits, [sd[n] for n in names], dlogits, allow_unused=True)
    for n, g in zip(names, gs):
        if g is None:
            grad_views[n].zero_()
        else:
            grad_views[n].copy_(g)
    return loss, logits.detach()


class Powerful(nn.Module):
    def __init__(self, use_norm_layers=False, name="ppgn", channel_num_list=None, feature_nums=None,
                 gnn_hidden_num_list=None, num_layers=6, input_features=2, hidden=32, hidden_final=32,
                 dropout_p=0.000001, simplified=False, n_nodes=30, config=None, normalization="none",
                 cat_output=True, adj_out=False, output_features=1, residual=False,
                 activation=None, spectral_norm=None, project_first=False, node_out=False, noise_mlp=False):
        super().__init__()
        if normalization not in ("none", "instance", "layer", "batch"):
            raise ValueError(f"unknown normalization {normalization!r}")          # model/ppgn.py:314
        unsupported = []
        if normalization != "instance": unsupported.append(f"normalization={normalization!r}")
        if not cat_output: unsupported.append("cat_output=False")
        if not adj_out: unsupported.append("adj_out=False")
        if simplified: unsupported.append("simplified=True")
        if residual: unsupported.append("residual=True")
        if project_first: unsupported.append("project_first=True")
        if node_out: unsupported.append("node_out=True")
        if input_features != 2: unsupported.append(f"input_features={input_features}")
        if output_features != 1: unsupported.append(f"output_features={output_features}")
        if hidden != hidden_final: unsupported.append("hidden != hidden_final")
        if hidden % 4: unsupported.append("hidden not a multiple of 4")
        if spectral_norm is not None: unsupported.append("spectral_norm")
        if unsupported:
            raise NotImplementedError("the HIP PPGN implements the configuration of utils/loading_utils.py:98-119 only; "
                                      "unsupported: " + ", ".join(unsupported))
        self.num_layers, self.hidden, self.hidden_final = num_layers, hidden, hidden_final
        self.cat_output, self.normalization, self.adj_out = cat_output, normalization, adj_out
        self.layer_after_conv = not simplified
        self.dropout_p = dropout_p
        self.residual, self.project_first, self.node_out = residual, project_first, node_out
        self.output_features = self.node_output_features = output_features
        self.noise_mlp = noise_mlp
        self.config = config
        self.precision = default_precision(config)
        # same registration order as the reference so that seeded initialisation consumes the RNG identically
        self.time_mlp = nn.Sequential(nn.Linear(1, 4), nn.GELU(), nn.Linear(4, 1))
        self.in_lin = _Seq(nn.Linear(input_features, hidden))
        self.layer_cat_lin = _Seq(nn.Linear(hidden * num_layers + input_features, hidden))
        self.convs = nn.ModuleList([PowerfulLayer(hidden, hidden, 2) for _ in range(num_layers)])
        self.bns = nn.ModuleList([])
        self.feature_extractors = nn.ModuleList([FeatureExtractor(hidden, hidden_final) for _ in range(num_layers)])
        self.after_conv = _Seq(nn.Linear(hidden_final, hidden_final))
        self.final_lin = _Seq(nn.Linear(hidden_final, output_features))
        self._pack_cache = None
        self._ws_cache = {}

    # ------------------------------------------------------------------ reference surface
    def get_out_dim(self):
        return self.output_features

    def forward(self, node_features, A, mask, noiselevel):
        return self.forward_cat(A, node_features, mask, noiselevel)

    def forward_cat(self, A, node_features, mask, noiselevel):
        del node_features   # ignored by the reference as well (model/ppgn.py:294)
        A, mask, sigma = self._prepare(A, mask, noiselevel)
        params = self._used_params()
        # grad mode is off inside Function.forward, so the decision to keep the backward workspace is taken here
        training = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if training and sigma.requires_grad:   # learned noise_mlp: d logits / d sigma flows back into time_mlp
            logits = _PackedFn.apply(self.num_layers, self.hidden, A, mask, sigma, None, *params)
        else:
            logits = _PowerfulFn.apply(self, training, A, mask, sigma, *params)
        return logits.unsqueeze(-1)

    # ------------------------------------------------------------------ engine surface (no autograd)
    @torch.no_grad()
    def forward_backward(self, A, mask, noiselevel, loss_grad_fn, grad_pack):
        """logits = f(A); (loss, dlogits) = loss_grad_fn(logits); parameter gradients are written through `grad_pack`
        (an ops.ParamPack over caller-owned buffers, e.g. views of one flat gradient bucket).  Returns (loss, logits)."""
        if isinstance(grad_pack, dict):   # learned noise_mlp: time_mlp is torch code in front of the kernels
            return autograd_forward_backward(self, A, mask, noiselevel, loss_grad_fn, grad_pack)
        A, mask, sigma = self._prepare(A, mask, noiselevel)
        B, N, _ = A.shape
        ws = self._workspace(B, N, True, A.device, cached=True)
        pack = self._pack()
        logits = ops.ppgn_forward(pack, A, mask, sigma, ws, training=True)
        loss, dlogits = loss_grad_fn(logits)
        ops.ppgn_backward(pack, grad_pack, A, mask, sigma, dlogits, ws)
        return loss, logits

    @torch.no_grad()
    def infer(self, A, mask, noiselevel, out=None):
        """Inference forward returning logits [B, N, N] (no trailing singleton), reusing one cached workspace."""
        A, mask, sigma = self._prepare(A, mask, noiselevel)
        B, N, _ = A.shape
        return ops.ppgn_forward(self._pack(), A, mask, sigma, self._workspace(B, N, False, A.device), training=False, out=out,
                                precision=self.precision)

    def chain_supported(self, N):
        """A whole sampling chain as one launch (include/dppm_hip.h dppm_ppgn_chain_*): small networks only, and not with a
        learned noise embedding (time_mlp is torch code in front of the kernels)."""
        return not self.noise_mlp and ops.chain_supported(self._pack(), N)

    @torch.no_grad()
    def chain_simple(self, flags, mask, level_table, seed, graph_id0=0, out=None):
        return ops.chain_simple(self._pack(), flags, mask, level_table, seed, graph_id0, out=out, precision=self.precision)

    @torch.no_grad()
    def chain_vlb(self, flags, mask, level_table, seed, graph_id0=0, out=None):
        return ops.chain_vlb(self._pack(), flags, mask, level_table, seed, graph_id0, out=out, precision=self.precision)

    # ------------------------------------------------------------------ helpers
    def used_param_names(self):
        """Names the kernels read, in C-ABI order."""
        return ops.used_param_names(self.num_layers)

    def trainable_param_names(self):
        """What an optimiser updates: the kernels' parameters, plus time_mlp when it produces sigma (model/ppgn.py:273-274)."""
        extra = [f"time_mlp.{i}.{k}" for i in (0, 2) for k in ("weight", "bias")] if self.noise_mlp else []
        return self.used_param_names() + extra

    def make_grad_pack(self, grad_views):
        """The object forward_backward writes gradients through: a C-ABI pointer pack, or the views themselves when autograd is
        in the loop."""
        return dict(grad_views) if self.noise_mlp else ops.ParamPack(grad_views, self.num_layers, self.hidden)

    def _train_logits(self, A, mask, noiselevel):
        return self.forward_cat(A, None, mask, noiselevel)[..., 0]

    def _used_params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in self.used_param_names()]

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() re-home the parameters: drop the cached pointer pack
        self._pack_cache = None
        return super()._apply(fn, *args, **kwargs)

    def _pack(self):
        """Host-side dppm_ppgn_params over the current parameter storage.  Cached; the cache is dropped by _apply (device /
        dtype moves), by engine.FlatState (re-homing into the flat bucket) and whenever the first or last parameter's address
        changes.  Code that swaps `p.data` of a middle parameter by hand must set `model._pack_cache = None`."""
        if self._pack_cache is not None:
            key, pack, first, last = self._pack_cache
            if first.data_ptr() == key[0] and last.data_ptr() == key[-1]:
                return pack
        params = self._used_params()
        key = tuple(p.data_ptr() for p in params)
        tensors = {}
        for n, p in zip(self.used_param_names(), params):
            if not p.is_cuda:
                raise RuntimeError(f"parameter {n} is on {p.device}: the PPGN hot path needs the GPU (no CPU fallback); "
                                   "call model.to('cuda')")
            if not p.data.is_contiguous():
                raise RuntimeError(f"parameter {n} must be contiguous")
            tensors[n] = p.data
        self._pack_cache = (key, ops.ParamPack(tensors, self.num_layers, self.hidden), params[0], params[-1])
        return self._pack_cache[1]

    def _workspace(self, B, N, training, device, cached=None):
        # autograd callers may run several forwards before one backward (the reference's fit() does B batch-1 calls), so
        # each training forward owns its workspace; inference and the engine path reuse a cached one.
        if cached is None:
            cached = not training
        if not cached:
            return ops.ppgn_workspace(B, N, self.hidden, self.num_layers, True, device)
        key = (B, N, bool(training), str(device))
        if key not in self._ws_cache:
            self._ws_cache.clear()          # one geometry at a time: workspaces are large
            self._ws_cache[key] = ops.ppgn_workspace(B, N, self.hidden, self.num_layers, training, device)
        return self._ws_cache[key]

    def _prepare(self, A, mask, noiselevel):
        if A.dim() == 4 and A.shape[-1] == 1:
            A = A[..., 0]
        if A.dim() != 3 or A.shape[-1] != A.shape[-2]:
            raise ValueError(f"A must be [B, N, N], got {tuple(A.shape)}")
        if not A.is_cuda:
            raise RuntimeError("Powerful.forward: A is not on the GPU; the PPGN hot path has no CPU fallback")
        B, N, _ = A.shape
        A = A.to(torch.float32).contiguous()
        mask = mask.to(device=A.device, dtype=torch.float32)
        if mask.dim() == 2:
            mask = mask.unsqueeze(0)
        mask = mask.expand(B, N, N).contiguous()
        if isinstance(noiselevel, torch.Tensor):
            sigma = noiselevel.detach().to(device=A.device, dtype=torch.float32).reshape(-1)
        else:
            # torch.full([1], noiselevel) in the reference rounds the python/numpy double to float32 (model/ppgn.py:283)
            sigma = torch.as_tensor(np.asarray(noiselevel, dtype=np.float64).reshape(-1).astype(np.float32), device=A.device)
        if sigma.numel() == 1:
            sigma = sigma.expand(B)
        if sigma.numel() != B:
            raise ValueError(f"noiselevel: {sigma.numel()} values for {B} graphs")
        sigma = sigma.contiguous()
        if self.noise_mlp:   # model/ppgn.py:273-274; under autograd the result keeps its graph (dsigma comes back from the C-ABI)
            sigma = self.time_mlp(sigma.view(B, 1)).view(B).contiguous()
        if self.training and self.dropout_p > 1e-4:
            raise NotImplementedError(f"dropout_p={self.dropout_p}: only the reference's p=1e-6 (treated as identity) is supported")
        return A, mask, sigma
