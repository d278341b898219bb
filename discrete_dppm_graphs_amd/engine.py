"""Device-resident training step and reverse sampling chains built on the C-ABI ops.

This is the part of the reference's `fit` (ppgn_simple.py:65-94, ppgn_vlb.py:158-199) and `run_sample`
(sample_ppgn_simple.py:70-88, sample_ppgn_vlb.py:111-131) that runs per batch, restated so that
  * the whole batch goes through ONE denoiser call (the reference loops over graphs with batch 1),
  * masks, noising, loss and gradient are fused kernels (the reference builds masks with Python loops on the CPU),
  * nothing synchronises with the host inside a step (the reference calls .item() every step),
  * parameters / gradients / Adam moments live in flat buffers, so data-parallel training needs exactly one
    all-reduce per step (RCCL over xGMI) and one fused optimiser kernel.
Random numbers are Philox draws keyed by (seed, global graph id, call offset): results are independent of how the
graphs are sharded over GPUs.
"""
import numpy as np
import torch

from . import ops


# ---------------------------------------------------------------------------------------------- schedules
def sigma_lin(sigma_list):
    """beta_t from the cumulative flip probabilities (ppgn_vlb.py:27-34 == sample_ppgn_vlb.py:28-35)."""
    sigmas = []
    for g, sigma in enumerate(sigma_list):
        if sigma < 1.0e-5:
            sigmas.append(0.0)
            continue
        sigmas.append(((1 - sigma) - (1 - sigma_list[g - 1])) / (1 - 2 * (1 - sigma_list[g - 1])))
    return sigmas


def schedule_simple(noise_num):
    """Descending sigmas the simple sampler walks (sample_ppgn_simple.py:45-47,81-82)."""
    s = torch.linspace(0, 1 / 2, noise_num).tolist()
    s.sort()
    return [s[len(s) - c - 1] for c in range(len(s))]


def schedule_vlb(noise_num):
    """(sigmatilde_t, beta_t, sigmatilde_{t-1}) for t = T..1 (sample_ppgn_vlb.py:67-70,122-126)."""
    s = torch.linspace(0, 1 / 2, noise_num + 1).tolist()
    s.sort()
    beta = sigma_lin(s)
    n = len(s)
    return [(s[n - c - 1], beta[n - c - 1], s[n - c - 2]) for c in range(n - 1)]


# ---------------------------------------------------------------------------------------------- flat parameters
class FlatState:
    """Trained parameters of a `Powerful` (or a conditioning variant) re-homed into one contiguous float32 buffer (each nn.Parameter becomes a view,
    names and shapes unchanged), plus a flat gradient bucket and Adam moments.  Unused parameters (time_mlp,
    feature_extractors) never receive gradients in the reference either and are left alone, exactly as
    torch.optim.Adam skips parameters whose .grad is None."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        self.names = model.trainable_param_names()
        named = dict(model.named_parameters())
        params = [named[n] for n in self.names]
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatState: move the model to the GPU first (no CPU fallback)")
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]     # keep every view 16-byte aligned
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).tolist()
        n = self.offsets[-1]
        self.flat = torch.zeros(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        grads = {}
        for name, p, off in zip(self.names, params, self.offsets):
            view = self.flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            grads[name] = self.grad[off:off + p.numel()].view_as(p)
        self.grad_views = grads
        self.grad_pack = model.make_grad_pack(grads)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        model._pack_cache = None

    def numel(self):
        return self.flat.numel()

    def all_reduce(self, group=None):
        """Sum the flat gradient bucket over the data-parallel ranks (one collective per step)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    def adam_step(self, lr=None):
        self.step_count += 1
        ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.lr if lr is None else lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, self.step_count)

    def state_dict(self):
        """Optimizer state as saved in the checkpoint's 'optimizer_state_dict' (flat moments + hyper-parameters)."""
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'lr': self.lr,
                'betas': list(self.betas), 'eps': self.eps, 'weight_decay': self.weight_decay, 'names': list(self.names),
                'offsets': list(self.offsets)}

    def load_state_dict(self, sd):
        """Restore what state_dict() saved.  The moments must belong to the same parameter layout (names / offsets are
        checked when present); a reference checkpoint's torch.optim.Adam state is not accepted here - its per-parameter
        tensors are keyed by index over ALL parameters, not by name."""
        if 'exp_avg' not in sd or 'exp_avg_sq' not in sd:
            raise ValueError("FlatState.load_state_dict: not a flat optimizer state (a torch.optim.Adam state_dict cannot be resumed)")
        if 'names' in sd and list(sd['names']) != list(self.names):
            raise ValueError("FlatState.load_state_dict: parameter names differ from this model's")
        if sd['exp_avg'].numel() != self.exp_avg.numel():
            raise ValueError("FlatState.load_state_dict: moment size %d != %d" % (sd['exp_avg'].numel(), self.exp_avg.numel()))
        self.exp_avg.copy_(sd['exp_avg'].to(self.exp_avg.device))
        self.exp_avg_sq.copy_(sd['exp_avg_sq'].to(self.exp_avg_sq.device))
        self.step_count = int(sd['step'])
        self.lr = float(sd.get('lr', self.lr))
        self.betas = tuple(sd.get('betas', self.betas))
        self.eps = float(sd.get('eps', self.eps))
        self.weight_decay = float(sd.get('weight_decay', self.weight_decay))

    def publish_grads(self):
        """Expose the flat bucket as .grad on the parameters (for callers that want to inspect or clip)."""
        named = dict(self.model.named_parameters())
        for name in self.names:
            named[name].grad = self.grad_views[name]


# ---------------------------------------------------------------------------------------------- training steps
def node_flags(adj):
    """train_node_flag_b = adj.sum(-1).gt(1e-5) (ppgn_simple.py:70)."""
    return adj.sum(-1).gt(1e-5).to(torch.float32)


def empty_step(state, like, dtype, group=None, lr=None, train=True):
    """The step of a data-parallel rank whose slice of the global batch is empty (a tail batch with fewer graphs than
    ranks): a zero gradient contribution that still joins the all-reduce and takes the same Adam step as every other
    rank.  Returns a zero loss on `like`'s device."""
    if train:
        state.grad.zero_()
        state.all_reduce(group)
        state.adam_step(lr)
    return torch.zeros((), device=like.device, dtype=dtype)


def _f64_dev(values, device):
    """B doubles on the device: a float64 device tensor is taken as it is (the per-epoch draw table of ppgn_simple._epoch_loop:
    no host-to-device copy per step), host values are uploaded."""
    if isinstance(values, torch.Tensor):
        return values.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.asarray(values, dtype=np.float64), device=device)


def train_step_simple(model, state, adj, sigmas, seed, offset, graph_id0=0, B_global=None, group=None, lr=None):
    """One optimiser step of the simple (BCE) objective on a device-resident batch `adj` [B,N,N].
    sigmas: B doubles, host values or a float64 device tensor (the reference draws np.random.uniform(0, .5, B),
    ppgn_simple.py:67).  Returns the loss as a 0-d float64 device tensor (this rank's share of the global mean; sum over
    ranks for the global value)."""
    B = adj.shape[0]
    if B == 0:
        return empty_step(state, adj, torch.float64, group, lr)
    flags = node_flags(adj)
    sig_dev = _f64_dev(sigmas, adj.device)
    A_t, target = ops.edgeflip(adj, flags, sig_dev, seed=seed, graph_id0=graph_id0, offset=offset)
    mask = ops.generate_mask(flags)

    def loss_grad(logits):
        return ops.loss_bce(logits, target, sig_dev, mask, B_global=B_global or B)

    loss, _ = model.forward_backward(A_t, mask, sig_dev, loss_grad, state.grad_pack)
    state.all_reduce(group)
    state.adam_step(lr)
    return loss


def vlb_batch_schedule(T, t_idx):
    """Per-graph (sigmatilde_t, beta_t, sigmatilde_{t-1}=st-st/t) as fit builds them (ppgn_vlb.py:161-168, :85-87)."""
    line = torch.linspace(0, 1 / 2, T + 1).tolist()
    beta = sigma_lin(line)
    st = np.asarray([line[i] for i in t_idx], dtype=np.float64)
    bt = np.asarray([beta[i] for i in t_idx], dtype=np.float64)
    st1 = st - st / np.asarray(t_idx, dtype=np.float64)
    return np.stack([st, bt, st1], 1)


def train_step_vlb(model, state, adj, t_idx, T, seed, offset, graph_id0=0, group=None, lr=None, sched=None):
    """One optimiser step of the VLB objective; t_idx: B integers in [1, T] (ppgn_vlb.py:161), or `sched`: the [B,3]
    float64 device table vlb_batch_schedule makes of them (ppgn_simple._epoch_loop uploads one per epoch)."""
    if adj.shape[0] == 0:
        return empty_step(state, adj, torch.float32, group, lr)
    flags = node_flags(adj)
    sched_dev = _f64_dev(vlb_batch_schedule(T, t_idx) if sched is None else sched, adj.device)
    sig_dev = sched_dev[:, 0].contiguous()
    A_t, _ = ops.edgeflip(adj, flags, sig_dev, seed=seed, graph_id0=graph_id0, offset=offset, want_target=False)
    mask = ops.generate_mask(flags)

    def loss_grad(logits):
        return ops.loss_vlb(logits, A_t, adj, sched_dev, mask)

    loss, _ = model.forward_backward(A_t, mask, sig_dev, loss_grad, state.grad_pack)
    state.all_reduce(group)
    state.adam_step(lr)
    return loss


# ---------------------------------------------------------------------------------------------- sampling chains
def chain_width(model, N, B, n_batches, cus=256):
    """How many of the reference's sampling batches (test.batch_size graphs each, run_sample's `while` loop,
    sample_ppgn_simple.py:72-88) one chain call may carry.  Where the denoiser runs one workgroup per graph (N <= 20, hidden 32:
    csrc/dppm_smallgraph.hip) a call's latency does not depend on the batch until every CU holds a graph, so consecutive batches
    ride one call; graph ids are global, hence the samples are the same either way."""
    small = N <= 20 and getattr(model, "hidden", None) == 32 and type(model).__name__ == "Powerful"
    return max(1, min(n_batches, cus // max(B, 1))) if small else 1


_SIGMA_ROWS = {}


def _sigma_rows(levels, like):
    """[T, B] float32 on the device, row c = the noise level of chain step c for every graph; the values are the float32
    roundings model._prepare would make.  A schedule is a pure function of the number of levels, so the table is uploaded once
    per (schedule, batch size, device) and kept: the chains of a sampling call copy nothing from the host."""
    key = (tuple(levels), like.shape[0], str(like.device))
    rows = _SIGMA_ROWS.get(key)
    if rows is None:
        if len(_SIGMA_ROWS) > 64:
            _SIGMA_ROWS.clear()
        col = torch.as_tensor(np.asarray(levels, dtype=np.float64).astype(np.float32), device=like.device)
        rows = _SIGMA_ROWS[key] = col.view(-1, 1).expand(-1, like.shape[0]).contiguous()
    return rows


_LEVEL_TABLES = {}


def level_table_simple(levels, device):
    """[T, 3] float32 on the device for ops.chain_simple: per level the network's noise input float32(sigma) (model._prepare's
    rounding) and the keep / flip probabilities float32(1 - sigma), float32(sigma) (dppm_edgeflip_fwd's roundings from double).
    Uploaded once per (schedule, device)."""
    key = (tuple(levels), str(device))
    tab = _LEVEL_TABLES.get(key)
    if tab is None:
        if len(_LEVEL_TABLES) > 64:
            _LEVEL_TABLES.clear()
        s = np.asarray(levels, dtype=np.float64)
        host = np.stack([s.astype(np.float32), (1.0 - s).astype(np.float32), s.astype(np.float32)], 1)
        tab = _LEVEL_TABLES[key] = torch.as_tensor(np.ascontiguousarray(host), device=device)
    return tab


def level_table_vlb(sched, device):
    """[T, 7] float32 on the device for ops.chain_vlb: per level float32(sigmatilde_t) for the network, then the six constants
    dppm_reverse_step_vlb rounds from (sigmatilde_t, beta_t, sigmatilde_{t-1}): 1 - b, b, st1, 1 - st1, st, 1 - st."""
    key = ("vlb", tuple(sched), str(device))
    tab = _LEVEL_TABLES.get(key)
    if tab is None:
        if len(_LEVEL_TABLES) > 64:
            _LEVEL_TABLES.clear()
        st, bt, st1 = (np.asarray([row[k] for row in sched], dtype=np.float64) for k in range(3))
        host = np.stack([st, 1.0 - bt, bt, st1, 1.0 - st1, st, 1.0 - st], 1).astype(np.float32)
        tab = _LEVEL_TABLES[key] = torch.as_tensor(np.ascontiguousarray(host), device=device)
    return tab


def _one_launch(model, flags, record):
    return record is None and getattr(model, "chain_supported", None) is not None and model.chain_supported(flags.shape[1])


@torch.no_grad()
def sample_chain_simple(model, flags, noise_num, seed, graph_id0=0, record=None):
    """x_T ~ Bern(1/2), then for sigma descending: re-noise x0_hat to sigma, predict flips, sample, XOR
    (sample_ppgn_simple.py:70-88).  flags [B,N] float32 on the GPU.  Returns the final adjacency [B,N,N].
    Small networks (hidden 32, N <= 20) run the whole chain as ONE launch -- same draws, same arithmetic, same bits as the
    level-by-level loop below, which stays for every other network and whenever `record` wants the intermediate tensors."""
    mask = ops.generate_mask(flags)
    if _one_launch(model, flags, record):
        return model.chain_simple(flags, mask, level_table_simple(schedule_simple(noise_num), flags.device), seed, graph_id0)
    x = ops.init_xT(flags, seed=seed, graph_id0=graph_id0, offset=0)
    a_t = torch.empty_like(x)
    logits = torch.empty_like(x)
    sched = schedule_simple(noise_num)
    sig_rows = _sigma_rows(sched, x)
    for c, sigma in enumerate(sched):
        ops.edgeflip(x, flags, sigma, seed=seed, graph_id0=graph_id0, offset=1 + 2 * c, want_target=False, out=a_t)
        model.infer(a_t, mask, sig_rows[c], out=logits)
        ops.reverse_step_simple(a_t, logits, flags, seed=seed, graph_id0=graph_id0, offset=2 + 2 * c, out=x)
        if record is not None:
            record(c, a_t, logits, x)
    return x


@torch.no_grad()
def sample_chain_vlb(model, flags, noise_num, seed, graph_id0=0, record=None):
    """x_T ~ Bern(1/2), then t = T..1: predict flips, mix posteriors, sample A_{t-1} (sample_ppgn_vlb.py:111-131).
    Small networks: one launch for the whole chain, as in sample_chain_simple."""
    mask = ops.generate_mask(flags)
    if _one_launch(model, flags, record):
        return model.chain_vlb(flags, mask, level_table_vlb(schedule_vlb(noise_num), flags.device), seed, graph_id0)
    x = ops.init_xT(flags, seed=seed, graph_id0=graph_id0, offset=0)
    y = torch.empty_like(x)
    logits = torch.empty_like(x)
    sched = schedule_vlb(noise_num)
    sig_rows = _sigma_rows([st for st, _, _ in sched], x)
    for c, (st, beta, st1) in enumerate(sched):
        model.infer(x, mask, sig_rows[c], out=logits)
        ops.reverse_step_vlb(x, logits, flags, st, beta, st1, seed=seed, graph_id0=graph_id0, offset=1 + c, out=y)
        if record is not None:
            record(c, x, logits, y)
        x, y = y, x
    return x
