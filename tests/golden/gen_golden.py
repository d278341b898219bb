#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE.

Run once in the build container (the reference lives read-only at /root/reference and never
travels; only the .npz data written here is committed):

    python tests/golden/gen_golden.py

What is exercised, unmodified, from /root/reference (CPU, torch fp32):
  G1  model/ppgn.py            Powerful.forward  (+ autograd grads through ppgn_simple.loss_func_bce)
  G2  utils/graph_utils.py     discretenoise, discretenoise_single, gen_list_of_data_single,
                               mask_adjs, generate_mask
  G3  ppgn_simple.py           loss_func_bce ;  ppgn_vlb.py loss_func_kld (values + d/dlogits)
  G4  sample_ppgn_simple.py / sample_ppgn_vlb.py   sample_main (closures gen_init_data, take_step,
                               run_sample executed as shipped, with module-level I/O names patched)
  G5  ppgn_vlb.py              sigma_lin

torch.bernoulli is replaced by `(U < p)` with U drawn from a recorded numpy stream so that every
random decision is explicit in the fixture (torch.bernoulli(p) is exactly "u < p" for u~U[0,1)).
Modules the image lacks (graph_tool, easydict, wandb, pyemd) and the plotting module are stubbed with
empty modules: none of them is on the arithmetic path.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------------
# stubs for absent, non-arithmetic imports
# --------------------------------------------------------------------------------------------
class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    gt = mod("graph_tool")
    gt.all = mod("graph_tool.all")
    mod("wandb")
    mod("pyemd", emd=None)
    mod("easydict", EasyDict=_EasyDict)
    noop = lambda *a, **k: None
    sys.path.insert(0, REF)
    import utils  # the reference package (real)

    class _AnyNoop(types.ModuleType):      # every plotting helper resolves to a no-op
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return noop

    vis = _AnyNoop("utils.visual_utils")
    sys.modules["utils.visual_utils"] = vis
    utils.visual_utils = vis


_install_stubs()

from model.ppgn import Powerful  # noqa: E402
from utils import graph_utils as ref_gu  # noqa: E402
import ppgn_simple as ref_simple  # noqa: E402
import ppgn_vlb as ref_vlb  # noqa: E402
import sample_ppgn_simple as ref_ssimple  # noqa: E402
import sample_ppgn_vlb as ref_svlb  # noqa: E402

CFG = SimpleNamespace(dev=torch.device("cpu"), noisetype="switched")


# --------------------------------------------------------------------------------------------
# explicit uniform stream in place of torch.bernoulli
# --------------------------------------------------------------------------------------------
class UniformStream:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.draws = []   # list of (U, p) per call
        self._orig = torch.bernoulli

    def bernoulli(self, p, *a, **k):
        u = self.rng.random(tuple(p.shape), dtype=np.float32)
        self.draws.append((u, p.detach().cpu().numpy().copy()))
        return (torch.from_numpy(u) < p).to(p.dtype)

    def __enter__(self):
        torch.bernoulli = self.bernoulli
        return self

    def __exit__(self, *exc):
        torch.bernoulli = self._orig


def make_graphs(rng, B, N, n_min, p_edge=0.3):
    """Symmetric 0/1 adjacencies, zero diagonal, active nodes first, every active node has degree>=1
    (so flags = adj.sum(-1) > 1e-5 recovers the node count, as the reference derives them)."""
    A = np.zeros((B, N, N), np.float32)
    for b in range(B):
        n = int(rng.integers(n_min, N + 1))
        up = np.triu((rng.random((n, n)) < p_edge).astype(np.float32), 1)
        for i in range(n - 1):      # a path guarantees degree >= 1
            up[i, i + 1] = 1.0
        A[b, :n, :n] = up + up.T
    return A


def make_model(N, H, L, seed, noise_mlp=False):
    torch.manual_seed(seed)
    m = Powerful(use_norm_layers=False, name="ppgn", channel_num_list=[], feature_nums=[],
                 gnn_hidden_num_list=[], num_layers=L, input_features=2, hidden=H, hidden_final=H,
                 dropout_p=1e-6, simplified=False, n_nodes=N, normalization="instance",
                 cat_output=True, adj_out=True, output_features=1, residual=False,
                 project_first=False, node_out=False, noise_mlp=noise_mlp, config=CFG)
    return m


def state_np(model):
    return {"w::" + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


# --------------------------------------------------------------------------------------------
# G1: PPGN logits + gradients, driven exactly as ppgn_simple.fit drives the model (:84-91)
# --------------------------------------------------------------------------------------------
def gen_ppgn(tag, N, H, L, B, seed):
    rng = np.random.default_rng(seed)
    A0 = make_graphs(rng, B, N, max(3, N // 2))
    model = make_model(N, H, L, seed)
    model.eval()   # dropout p=1e-6 is live in train(); parity is asserted in eval()
    sigmas = list(rng.uniform(0.0, 0.5, size=B))
    adj = torch.from_numpy(A0)
    flags = adj.sum(-1).gt(1e-5).to(torch.float32)
    with UniformStream(seed + 1) as us:
        _, A_t, flags_out, targets = ref_gu.gen_list_of_data_single(
            torch.zeros(B, N, 1), adj, flags, sigmas, CFG)
    A_t_ch = A_t.chunk(B, dim=0)
    fl_ch = flags_out.chunk(B, dim=0)
    score, masks = [], []
    for i, s in enumerate(sigmas):
        mask = ref_gu.generate_mask(fl_ch[i])
        score.append(model(A=A_t_ch[i].unsqueeze(0), node_features=A_t_ch[i], mask=mask, noiselevel=s))
        masks.append(mask)
    score = torch.cat(score, 0).squeeze(-1)
    score.retain_grad()
    masktens = torch.cat(masks, 0)
    loss = ref_simple.loss_func_bce(score, torch.stack(targets), sigmas, CFG, masktens)
    loss.backward()
    out = dict(state_np(model))
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["g::" + k] = p.grad.detach().numpy()
    # one batched call with a single scalar sigma (what the samplers do, sample_ppgn_simple.py:83)
    with torch.no_grad():
        logits_batched = model(None, torch.stack(list(A_t_ch)), masktens, float(sigmas[0])).squeeze(-1)
    out.update(A0=A0, A_t=torch.stack(list(A_t_ch)).numpy(), flags=flags.numpy(), mask=masktens.numpy(),
               sigmas=np.asarray(sigmas, np.float64), targets=torch.stack(targets).numpy(),
               U_flip=np.stack([u for u, _ in us.draws]),
               logits=score.detach().numpy(), dlogits=score.grad.numpy(),
               loss=np.float64(loss.item()), logits_batched_sigma0=logits_batched.numpy(),
               N=N, H=H, L=L)
    np.savez_compressed(os.path.join(OUT, f"g1_ppgn_{tag}.npz"), **out)
    print("G1", tag, "loss", loss.item(), "|logits|max", score.abs().max().item())


# --------------------------------------------------------------------------------------------
# G2: forward noising
# --------------------------------------------------------------------------------------------
def gen_flip(seed=7):
    rng = np.random.default_rng(seed)
    B, N = 5, 12
    A0 = make_graphs(rng, B, N, 4)
    adj = torch.from_numpy(A0)
    flags = adj.sum(-1).gt(1e-5).to(torch.float32)
    out = dict(A0=A0, flags=flags.numpy())
    # batched, one scalar sigma (sampler re-noising, sample_ppgn_simple.py:51)
    for k, s in enumerate([0.0, 0.0625, 0.25, 0.5]):
        with UniformStream(seed + 10 + k) as us:
            a_t, tgt = ref_gu.discretenoise(adj.clone(), flags, s, CFG)
        out[f"batched_{k}_sigma"] = np.float64(s)
        out[f"batched_{k}_U"] = us.draws[0][0]
        out[f"batched_{k}_At"] = a_t.numpy()
        out[f"batched_{k}_target"] = tgt.numpy()
    # per graph sigma (training, graph_utils.py:96-138)
    sigmas = list(rng.uniform(0, 0.5, size=B))
    with UniformStream(seed + 20) as us:
        _, a_t, fl, tg = ref_gu.gen_list_of_data_single(torch.zeros(B, N, 1), adj.clone(), flags, sigmas, CFG)
    out.update(single_sigmas=np.asarray(sigmas, np.float64), single_U=np.stack([u for u, _ in us.draws]),
               single_At=a_t.numpy().reshape(B, N, N), single_target=torch.stack(tg).numpy(),
               mask=ref_gu.generate_mask(flags).numpy(),
               mask_adjs_in=rng.random((B, N, N), dtype=np.float32))
    out["mask_adjs_out"] = ref_gu.mask_adjs(torch.from_numpy(out["mask_adjs_in"]), flags).numpy()
    np.savez_compressed(os.path.join(OUT, "g2_flip.npz"), **out)
    print("G2 flip ok")


# --------------------------------------------------------------------------------------------
# G3: losses
# --------------------------------------------------------------------------------------------
def gen_losses(seed=11):
    rng = np.random.default_rng(seed)
    B, N, T = 6, 10, 16
    A0 = make_graphs(rng, B, N, 4)
    adj = torch.from_numpy(A0)
    flags = adj.sum(-1).gt(1e-5).to(torch.float32)
    mask = ref_gu.generate_mask(flags)
    out = dict(A0=A0, flags=flags.numpy(), mask=mask.numpy(), T=T)
    # --- simple / BCE (ppgn_simple.py:25-34)
    sigmas = list(rng.uniform(0, 0.5, size=B))
    with UniformStream(seed + 1):
        _, a_t, _, tg = ref_gu.gen_list_of_data_single(torch.zeros(B, N, 1), adj.clone(), flags, sigmas, CFG)
    score = torch.from_numpy(rng.normal(0, 2.0, size=(B, N, N)).astype(np.float32)).requires_grad_(True)
    loss = ref_simple.loss_func_bce(score, torch.stack(tg), sigmas, CFG, mask)
    loss.backward()
    out.update(bce_sigmas=np.asarray(sigmas, np.float64), bce_score=score.detach().numpy(),
               bce_target=torch.stack(tg).numpy(), bce_loss=np.float64(loss.item()),
               bce_dscore=score.grad.numpy())
    # --- VLB / KL (ppgn_vlb.py:79-125), schedule exactly as fit builds it (:161-168)
    ind = np.array([1, 2, T // 2, T - 1, T, 3], dtype=np.int64)   # includes t=1 and t=T edge cases
    sigma_line = torch.linspace(0, 1 / 2, T + 1).tolist()
    sigma_list = [sigma_line[i] for i in ind]
    sig_list = ref_vlb.sigma_lin(sigma_line)
    nontild = [sig_list[i] for i in ind]
    with UniformStream(seed + 2):
        _, a_t, _, tg = ref_gu.gen_list_of_data_single(torch.zeros(B, N, 1), adj.clone(), flags, sigma_list, CFG)
    a_t_st = torch.stack(a_t.chunk(B, dim=0))
    leaf = torch.from_numpy(rng.normal(0, 2.0, size=(B, N, N)).astype(np.float32)).requires_grad_(True)
    score_in = leaf.clone()   # the reference overwrites score_list in place (:111)
    loss = ref_vlb.loss_func_kld(score_in, a_t_st, adj, torch.stack(tg), sigma_list, ind, nontild, CFG, mask)
    loss.backward()
    out.update(vlb_ind=ind, vlb_sigma_list=np.asarray(sigma_list, np.float64),
               vlb_nontild=np.asarray(nontild, np.float64), vlb_At=a_t_st.numpy(),
               vlb_score=leaf.detach().numpy(), vlb_loss=np.float32(loss.item()),
               vlb_dscore=leaf.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "g3_losses.npz"), **out)
    print("G3 bce", out["bce_loss"], "vlb", out["vlb_loss"])


# --------------------------------------------------------------------------------------------
# G4: full reverse chains through the reference's sample_main (closures run as shipped)
# --------------------------------------------------------------------------------------------
def gen_chain(kind, seed, B=4, N=10, T=16, H=8, L=2):
    mod = ref_ssimple if kind == "simple" else ref_svlb
    rng = np.random.default_rng(seed)
    pool = make_graphs(rng, 7, N, 4)                 # stands in for the training graphs
    model = make_model(N, H, L, seed)
    model.eval()
    rec = dict(model_A=[], model_logits=[], model_sigma=[], final=None)

    class RecModel(torch.nn.Module):
        def forward(self, feat, A, mask, noise):
            out = model(feat, A, mask, noise)
            rec["model_A"].append(A.numpy().copy())
            rec["model_logits"].append(out.squeeze(-1).numpy().copy())
            rec["model_sigma"].append(float(noise))
            return out

    def fake_graphs_to_tensor(config, graph_list):
        adjs = torch.from_numpy(np.stack(graph_list))
        return adjs, torch.zeros(adjs.size(0), N, 1)

    def rec_adjs_to_graphs(adjs, node_flags=None):
        rec["final"] = np.asarray(adjs).copy()
        return [None] * len(adjs)

    config = _EasyDict(dev=CFG.dev, noisetype="switched", samplesize=B, save_dir="/tmp",
                       dataset=dict(max_node_num=N, name="synthetic"), test=dict(batch_size=B))
    saved = {k: getattr(mod, k) for k in ("load_data", "graphs_to_tensor", "prepare_test_model_train",
                                          "load_model", "adjs_to_graphs", "plot_graphs_list", "eval_graph_list")}
    mod.load_data = lambda config, get_graph_list=False: ([g for g in pool], [g for g in pool])
    mod.graphs_to_tensor = fake_graphs_to_tensor
    mod.prepare_test_model_train = lambda config, link: [("f", [0], [])]
    mod.load_model = lambda *a: RecModel()
    mod.adjs_to_graphs = rec_adjs_to_graphs
    mod.plot_graphs_list = lambda *a, **k: None
    mod.eval_graph_list = lambda *a, **k: {}
    np.random.seed(seed)
    rand_idx = np.random.randint(0, len(pool), B)    # what gen_init_data will draw (:25 / :45)
    np.random.seed(seed)
    try:
        with UniformStream(seed + 5) as us:
            mod.sample_main(config, "unused", 0, T)
    finally:
        for k, v in saved.items():
            setattr(mod, k, v)
    base = pool[rand_idx]
    flags = (base.sum(-1) > 1e-5).astype(np.float32)
    out = dict(state_np(model))
    out.update(kind=kind, T=T, N=N, H=H, L=L, flags=flags,
               U=np.stack([u for u, _ in us.draws]), P=np.stack([p for _, p in us.draws]),
               model_A=np.stack(rec["model_A"]), model_logits=np.stack(rec["model_logits"]),
               model_sigma=np.asarray(rec["model_sigma"], np.float64), final=rec["final"])
    np.savez_compressed(os.path.join(OUT, f"g4_chain_{kind}.npz"), **out)
    print("G4", kind, "steps", len(rec["model_A"]), "bernoulli calls", len(us.draws),
          "final edges", rec["final"].sum() / 2)


# --------------------------------------------------------------------------------------------
# G5: sigma_lin
# --------------------------------------------------------------------------------------------
def gen_sigma_lin():
    out = {}
    for T in (8, 16, 32, 64, 128):
        line = torch.linspace(0, 1 / 2, T + 1).tolist()
        out[f"line_{T}"] = np.asarray(line, np.float64)
        out[f"beta_{T}"] = np.asarray(ref_vlb.sigma_lin(line), np.float64)
        assert ref_vlb.sigma_lin(line) == ref_svlb.sigma_lin(line)
    np.savez_compressed(os.path.join(OUT, "g5_sigma_lin.npz"), **out)
    print("G5 ok")


if __name__ == "__main__":
    torch.set_num_threads(1)
    gen_ppgn("n6_h8_l2", N=6, H=8, L=2, B=4, seed=101)
    gen_ppgn("n18_h32_l6", N=18, H=32, L=6, B=6, seed=102)
    gen_ppgn("n33_h16_l3", N=33, H=16, L=3, B=3, seed=103)
    gen_flip()
    gen_losses()
    gen_chain("simple", seed=201)
    gen_chain("vlb", seed=202)
    gen_sigma_lin()
