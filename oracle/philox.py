"""Philox4x32-10 counter-based RNG, numpy restatement (TEST INFRASTRUCTURE, see oracle/__init__.py).

No counterpart in the reference (it draws from torch's global generator, utils/graph_utils.py:159);
this is the published Philox4x32-10 of Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3"
(SC'11) with the standard constants.  The HIP kernels (csrc/philox.h) implement the same function;
known-answer vectors from the Random123 distribution are checked in tests/test_oracle_golden.py.

Draw contract used by every kernel of this project:
    u(seed, graph_id, offset, i, j) = (philox(key=(seed_lo, seed_hi), ctr=(i, j, graph_id, offset))[0] >> 8) * 2^-24
with i < j the upper-triangle pair, graph_id the GLOBAL index of the graph in the job (so results do
not depend on batch size or GPU count) and offset the index of the Bernoulli call in the chain.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: 4 broadcastable uint32 arrays, key: 2 uint32 scalars/arrays -> tuple of 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(*ctr)]
    k0 = np.asarray(key[0], dtype=np.uint32)
    k1 = np.asarray(key[1], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return c0, c1, c2, c3


def uniform_pairs(seed, graph_ids, offset, N):
    """Uniforms in [0,1) for every (i, j) of an N x N grid, for each graph id -> float32 [B, N, N].

    Only the strict upper triangle is consumed by the kernels; the full grid is returned so that it can be
    fed to the explicit-U entry points in place of what torch.bernoulli would have drawn."""
    graph_ids = np.asarray(graph_ids, dtype=np.uint32).reshape(-1, 1, 1)
    i = np.arange(N, dtype=np.uint32).reshape(1, N, 1)
    j = np.arange(N, dtype=np.uint32).reshape(1, 1, N)
    seed = int(seed)
    key = (np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    r0 = philox4x32_10((i, j, graph_ids, np.uint32(offset)), key)[0]
    return ((r0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
