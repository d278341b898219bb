"""Mirror of the reference's utils/graph_utils.py for the discrete ("switched") noise path, backed by the HIP kernels.

Same names, argument order and return values as the reference; every function also accepts the explicit uniform
draws (`U=`) that torch.bernoulli would have consumed, which is how the parity tests compare bit for bit.  Without `U`
the kernels draw Philox numbers keyed by (seed, graph id, call counter); `set_rng(seed)` resets the counter, as
torch.manual_seed does for the reference (utils/arg_helper.py:99).
The deprecated "balanced"/"density" corruptions (graph_utils.py:167-233,249-296) are not provided: the reference's own
configs forbid them (`noisetype: switched  # deprecated, do not change this`).
"""
import numpy as np
import torch

from .. import ops

_RNG = {"seed": 0, "offset": 0}


def set_rng(seed, offset=0):
    _RNG["seed"], _RNG["offset"] = int(seed), int(offset)


def _next_offset():
    o = _RNG["offset"]
    _RNG["offset"] = o + 1
    return o


def _check_noisetype(config):
    nt = getattr(config, "noisetype", "switched") if config is not None else "switched"
    if nt != "switched":
        raise NotImplementedError(f"noisetype={nt!r}: only the 'switched' edge-flip process is implemented "
                                  "(the reference marks every other value deprecated)")


def _dev(config, like=None):
    dev = getattr(config, "dev", None) if config is not None else None
    if dev is None and like is not None:
        dev = like.device
    dev = torch.device(dev if dev is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError(f"config.dev={dev}: the edge-flip kernels run on the GPU only (no CPU fallback)")
    return dev


def mask_adjs(adjs, node_flags):
    """reference utils/graph_utils.py:7-19 (plain elementwise torch, any device)."""
    if len(adjs.shape) == 4:
        node_flags = node_flags.unsqueeze(1)
    adjs = adjs * node_flags.unsqueeze(-1)
    adjs = adjs * node_flags.unsqueeze(-2)
    return adjs


def generate_mask(node_flags):
    """reference utils/graph_utils.py:312-319: [B, N] flags -> [B, N, N] mask (one kernel instead of a Python triple
    loop).  The result stays on the device of `node_flags` when that is a GPU; CPU flags are moved to the GPU."""
    if not node_flags.is_cuda:
        node_flags = node_flags.cuda()
    return ops.generate_mask(node_flags.to(torch.float32).contiguous())


def discretenoise(train_adj_b, node_flags, sigma, config, U=None, graph_id0=0):
    """reference utils/graph_utils.py:155-165: batched q(A_t|A_0) with one sigma -> (A_t, flip indicator)."""
    _check_noisetype(config)
    dev = _dev(config, train_adj_b)
    A0 = train_adj_b.to(device=dev, dtype=torch.float32).contiguous()
    flags = node_flags.to(device=dev, dtype=torch.float32).contiguous()
    if U is not None:
        U = torch.as_tensor(U, dtype=torch.float32, device=dev).contiguous()
    return ops.edgeflip(A0, flags, float(sigma), U=U, seed=_RNG["seed"], graph_id0=graph_id0, offset=_next_offset())


def discretenoise_single(train_adj_b, node_flags, sigma, config, U=None, graph_id=0):
    """reference utils/graph_utils.py:236-246: one graph [N, N] with its own sigma."""
    A_t, tgt = discretenoise(train_adj_b.unsqueeze(0), node_flags.unsqueeze(0), sigma, config,
                             U=None if U is None else torch.as_tensor(U).unsqueeze(0), graph_id0=graph_id)
    return A_t[0], tgt[0]


def gen_list_of_data_single(train_x_b, train_adj_b, train_node_flag_b, sigma_list, config, U=None, graph_id0=0):
    """reference utils/graph_utils.py:96-138: graph i is noised with sigma_list[i].  One kernel launch for the whole
    batch instead of B launches x 9 ops.  Returns (x, A_t as [(B*N), N], flags, list of B targets) like the reference."""
    assert isinstance(sigma_list, list)
    _check_noisetype(config)
    dev = _dev(config, train_adj_b)
    B = len(sigma_list)
    A0 = train_adj_b.to(device=dev, dtype=torch.float32).contiguous()
    flags = train_node_flag_b.to(device=dev, dtype=torch.float32).contiguous()
    if U is not None:
        U = torch.as_tensor(U, dtype=torch.float32, device=dev).contiguous()
    A_t, tgt = ops.edgeflip(A0[:B], flags[:B], np.asarray(sigma_list, dtype=np.float64), U=U, seed=_RNG["seed"],
                            graph_id0=graph_id0, offset=_next_offset())
    N = A_t.shape[-1]
    train_x_b = train_x_b.repeat(len(sigma_list), 1, 1)      # as the reference does (:136)
    return train_x_b, A_t.reshape(B * N, N), train_node_flag_b.repeat(1, 1), list(tgt.unbind(0))


def pad_adjs(ori_adj, node_number):
    """reference utils/graph_utils.py:298-308."""
    a = ori_adj
    ori_len = a.shape[-1]
    if ori_len == node_number:
        return a
    if ori_len > node_number:
        raise ValueError(f'ori_len {ori_len} > node_number {node_number}')
    a = np.concatenate([a, np.zeros([ori_len, node_number - ori_len])], axis=-1)
    a = np.concatenate([a, np.zeros([node_number - ori_len, node_number])], axis=0)
    return a
