"""dppm_loss_bce_fwd_bwd_noalloc (include/dppm_hip.h): the BCE loss for callers that cannot allocate -- its partial sums live in a
file-scope device array of the library.  Called through the C-ABI and compared bit for bit with dppm_loss_bce_fwd_bwd (the entry
point with a caller workspace) on two calls in a row, the second on a smaller input, plus the size limit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(rng, B, N):
    score = torch.as_tensor(rng.normal(size=(B, N, N)).astype(np.float32) * 3).cuda()
    target = torch.as_tensor((rng.random((B, N, N)) < 0.3).astype(np.float32)).cuda()
    n = rng.integers(max(2, N // 2), N + 1, size=B)
    flags = (np.arange(N)[None, :] < n[:, None]).astype(np.float32)
    mask = torch.as_tensor(flags[:, :, None] * flags[:, None, :]).cuda()
    sigma = torch.as_tensor(rng.uniform(0, 0.5, size=B)).cuda()
    return score, target, sigma, mask


def _both(lib, check, score, target, sigma, mask, B_global):
    B, N, _ = score.shape
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    nbytes = lib.dppm_loss_workspace_bytes(B, N)
    ws = torch.empty(max(nbytes, 8) // 8, device="cuda", dtype=torch.float64)
    loss_a = torch.full((), float("nan"), device="cuda", dtype=torch.float64)
    loss_b = torch.full((), float("nan"), device="cuda", dtype=torch.float64)
    d_a = torch.full_like(score, float("nan"))
    d_b = torch.full_like(score, float("nan"))
    check(lib.dppm_loss_bce_fwd_bwd(p(score), p(target), p(sigma), p(mask), B, N, B_global, p(loss_a), p(d_a), p(ws), C.c_size_t(nbytes), st),
          "dppm_loss_bce_fwd_bwd")
    check(lib.dppm_loss_bce_fwd_bwd_noalloc(p(score), p(target), p(sigma), p(mask), B, N, B_global, p(loss_b), p(d_b), st),
          "dppm_loss_bce_fwd_bwd_noalloc")
    torch.cuda.synchronize()
    return loss_a, d_a, loss_b, d_b


def test_noalloc_loss_equals_workspace_loss_twice_second_smaller():
    from discrete_dppm_graphs_amd._lib import check, lib
    rng = np.random.default_rng(77)
    for B, N, Bg in ((48, 37, 64), (5, 12, 5)):           # the second call is smaller: stale partials of the first must not leak in
        score, target, sigma, mask = _case(rng, B, N)
        loss_a, d_a, loss_b, d_b = _both(lib, check, score, target, sigma, mask, Bg)
        assert torch.isfinite(loss_b) and torch.equal(loss_a, loss_b)
        assert torch.equal(d_a, d_b)
    # the loss-only form (dscore = NULL) and the oracle's value
    from oracle import diffusion as od
    score, target, sigma, mask = _case(rng, 3, 9)
    loss = torch.full((), float("nan"), device="cuda", dtype=torch.float64)
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.dppm_loss_bce_fwd_bwd_noalloc(p(score), p(target), p(sigma), p(mask), 3, 9, 3, p(loss), C.c_void_p(0),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dppm_loss_bce_fwd_bwd_noalloc")
    ref = od.loss_bce(score.cpu(), target.cpu(), sigma.cpu().numpy(), mask.cpu())
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)


def test_noalloc_loss_rejects_oversize_batches():
    from discrete_dppm_graphs_amd._lib import lib
    t = torch.zeros(1, device="cuda")
    d = torch.zeros(1, device="cuda", dtype=torch.float64)
    p = lambda x: C.c_void_p(x.data_ptr())
    rc = lib.dppm_loss_bce_fwd_bwd_noalloc(p(t), p(t), p(d), p(t), 4097, 1, 4097, p(d), C.c_void_p(0), C.c_void_p(0))
    assert rc == -1 and b"4096" in lib.dppm_last_error()
