"""Device-resident dataset: the step in front of both hot loops.

The reference wraps the padded adjacencies in DataLoader(TensorDataset(adjs, x), batch_size, shuffle=True)
(utils/arg_helper.py:130-134) and moves every batch to the device inside the training loop (ppgn_simple.py:68-69); its
samplers re-read the node counts of random training graphs on the host for every sampling batch
(sample_ppgn_vlb.py:44-49).  Here the whole padded set is uploaded ONCE and a batch is an index_select on the device.
Everything random on the host side -- the shuffle order, the per-graph noise draws of an epoch, the node-count pool
indices of a sampling call -- comes from numpy generators keyed by (seed, stream, counter), never from numpy's global
state, so every data-parallel rank computes the same global batch and results do not depend on the number of GPUs.
"""
import numpy as np
import torch

# stream ids of the host generators (second word of the SeedSequence key)
STREAM_SHUFFLE, STREAM_NOISE, STREAM_POOL = 1, 2, 3


def host_rng(seed, stream, counter=0):
    """numpy Generator keyed by (seed, stream, counter): identical on every rank, independent of numpy's global state."""
    return np.random.default_rng([int(seed) & 0xFFFFFFFF, int(stream), int(counter)])


def upload(array, device, dtype=None):
    """The one place host arrays of this module cross to the device (counted, so tests can assert it happens per epoch /
    per sampling call and not per step)."""
    upload.calls += 1
    t = torch.as_tensor(np.ascontiguousarray(array))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


upload.calls = 0


class ResidentLoader:
    """Iterable of (adj_b, x_b) batches with DataLoader's surface (len(), iteration, .batch_size, .dataset) over tensors that
    live on `device`.  One pass = one epoch: its shuffle permutation is drawn from host_rng(seed, STREAM_SHUFFLE, epoch)
    and uploaded once; every batch is an index_select on the device (no host-to-device copy per step)."""

    def __init__(self, adjs, x, batch_size, shuffle=True, seed=0, device=None, drop_last=False):
        if adjs.shape[0] != x.shape[0]:
            raise ValueError(f"ResidentLoader: {adjs.shape[0]} adjacencies but {x.shape[0]} feature rows")
        self.device = torch.device(device) if device is not None else adjs.device
        self.adjs = adjs.to(self.device).contiguous()          # the one upload of the dataset
        self.x = x.to(self.device).contiguous()
        self.batch_size, self.shuffle, self.seed, self.drop_last = int(batch_size), bool(shuffle), int(seed), bool(drop_last)
        self.epoch = 0
        self.dataset = torch.utils.data.TensorDataset(self.adjs, self.x)

    def __len__(self):
        G = self.adjs.shape[0]
        return G // self.batch_size if self.drop_last else -(-G // self.batch_size)

    def batch_sizes(self):
        """Sizes of the batches one pass yields, in order."""
        G, bs = self.adjs.shape[0], self.batch_size
        sizes = [bs] * (G // bs)
        if G % bs and not self.drop_last:
            sizes.append(G % bs)
        return sizes

    def order(self, epoch):
        """Host permutation of pass number `epoch` (the same on every rank)."""
        G = self.adjs.shape[0]
        return host_rng(self.seed, STREAM_SHUFFLE, epoch).permutation(G) if self.shuffle else np.arange(G)

    def __iter__(self):
        perm = upload(self.order(self.epoch), self.device, torch.int64)
        self.epoch += 1
        lo = 0
        for n in self.batch_sizes():
            idx = perm[lo:lo + n]
            lo += n
            yield self.adjs.index_select(0, idx), self.x.index_select(0, idx)

    def pool(self):
        """All adjacencies [G, N, N], resident (what sample_testing draws node counts from)."""
        return self.adjs


def batch_sizes_of(dl):
    """Batch sizes of one pass over `dl` (a ResidentLoader, or a torch DataLoader over a sized dataset)."""
    if hasattr(dl, "batch_sizes"):
        return dl.batch_sizes()
    G, bs = len(dl.dataset), dl.batch_size
    sizes = [bs] * (G // bs)
    if G % bs and not getattr(dl, "drop_last", False):
        sizes.append(G % bs)
    return sizes
