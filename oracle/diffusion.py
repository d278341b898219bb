"""CPU restatement of the discrete edge-flip diffusion ops (TEST INFRASTRUCTURE, see oracle/__init__.py).

Every function takes the uniforms U that torch.bernoulli would have drawn as an explicit argument
(bernoulli(p) == (u < p)), so that results are a pure function of the inputs.  Arithmetic is torch-CPU
float32 in exactly the reference's operation order, so the HIP kernels can be compared bit for bit.
Pinned against tests/golden/g2_flip.npz, g3_losses.npz, g4_chain_*.npz, g5_sigma_lin.npz.
"""
import numpy as np
import torch

F32 = torch.float32


def _t(x, dtype=F32):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), dtype=dtype)


def _per_graph(vals, B):
    """python/np float64 scalar(s) -> float32 tensor [B,1,1]; rounding happens once, double -> float,
    exactly as torch.full_like(x, python_float) / torch.where(c, python_float, python_float) do."""
    v = np.asarray(vals, dtype=np.float64).reshape(-1)
    if v.size == 1:
        v = np.repeat(v, B)
    assert v.size == B
    return torch.from_numpy(v.astype(np.float32)).view(B, 1, 1), v


def generate_mask(flags):
    """reference utils/graph_utils.py:312-319 (Python triple loop) -> outer product of (flag >= 0.9)."""
    f = (_t(flags) >= 0.9).to(F32)
    return f.unsqueeze(-1) * f.unsqueeze(-2)


def generate_mask_loop(flags):
    """The same mask built the way the reference builds it (utils/graph_utils.py:312-319): one Python-level visit per
    (graph, i, j) with tensor-scalar comparisons, on the CPU.  Only bench.py's reference-faithful CPU baseline times this;
    tests check it equals generate_mask."""
    flags = _t(flags)
    B, N = flags.shape
    out = torch.zeros(B, N, N)
    for b, row in enumerate(flags):
        for i, fi in enumerate(row):
            for j, fj in enumerate(row):
                if fj >= 0.9 and fi >= 0.9:
                    out[b][i][j] = 1
    return out


def mask_adjs(adjs, flags):
    """reference utils/graph_utils.py:7-19 (3-D case)."""
    adjs, flags = _t(adjs), _t(flags)
    return adjs * flags.unsqueeze(-1) * flags.unsqueeze(-2)


def discretenoise(A0, flags, sigma, U):
    """q(A_t|A_0): reference utils/graph_utils.py:155-165 (batched, one sigma) and :236-246 (_single, one
    graph) / :96-138 (gen_list_of_data_single, a sigma per graph).  A0 [B,N,N], flags [B,N], sigma scalar or
    [B], U [B,N,N] -> (A_t, target) float32 [B,N,N]."""
    A0, flags, U = _t(A0), _t(flags), _t(U)
    B = A0.shape[0]
    _, s64 = _per_graph(sigma, B)
    p_keep = torch.from_numpy((1.0 - s64).astype(np.float32)).view(B, 1, 1)
    p_flip = torch.from_numpy(s64.astype(np.float32)).view(B, 1, 1)
    bern = torch.where(A0 > 0.5, p_keep, p_flip).expand_as(A0)
    upper = (U < bern).to(F32).triu(diagonal=1)
    lower = upper.transpose(-1, -2)
    target = torch.abs(-A0 + upper + lower)
    A_t = upper + lower
    return mask_adjs(A_t, flags), mask_adjs(target, flags)


def gen_init_data(flags, U):
    """x_T: reference sample_ppgn_simple.py:31-41 / sample_ppgn_vlb.py:51-62.  p = 1/2 where both nodes active."""
    flags, U = _t(flags), _t(U)
    act = (flags > 0.5).to(F32)
    bern = 0.5 * act.unsqueeze(-1) * act.unsqueeze(-2)
    upper = (U < bern).to(F32).triu(diagonal=1)
    return upper + upper.transpose(-1, -2)


def gen_init_prob_loop(flags):
    """The p = 1/2 matrix of gen_init_data built as the reference builds it (sample_ppgn_simple.py:32-37): a Python loop over
    every (graph, i, j) with tensor-scalar reads.  Timed by bench.py's reference-faithful CPU baseline only."""
    flags = _t(flags)
    B, N = flags.shape
    bern = torch.zeros(B, N, N)
    for k in range(B):
        for i in range(N):
            for j in range(N):
                if 1 / 2 < flags[k][i] and 1 / 2 < flags[k][j]:
                    bern[k, i, j] = 1 / 2
    return bern


def probs_from_logits(logits):
    return torch.sigmoid(_t(logits))


def reverse_step_simple(A_t, probs, flags, U):
    """reference sample_ppgn_simple.py:59-68, from the post-sigmoid flip probabilities `probs`:
    r=(r+r^T)/2 * mask; F=sym(triu(bern(r),1)); x0_hat = A_t XOR F."""
    A_t, r, U = _t(A_t), _t(probs), _t(U)
    mask = generate_mask(flags)
    r = (r + r.transpose(-2, -1)) / 2
    r = r * mask
    noise = (U < r).to(F32)
    noise = torch.triu(noise, diagonal=1) + torch.triu(noise, diagonal=1).transpose(-2, -1)
    inter = torch.where(noise > 0.5, A_t - 1, A_t)
    return torch.where(inter < -0.5, inter + 2, inter)


def take_step_simple(x0_hat, logits_fn, flags, sigma, U_noise, U_sample):
    """One full step of the simple sampler (sample_ppgn_simple.py:54-68): re-noise x0_hat to level sigma,
    predict, sample.  logits_fn(A_t) -> logits [B,N,N]."""
    A_t, _ = discretenoise(x0_hat, flags, sigma, U_noise)
    return reverse_step_simple(A_t, probs_from_logits(logits_fn(A_t)), flags, U_sample), A_t


def vlb_scalars(sigmatilde_t, sigma_t, sigmatilde_t1):
    """The six float32 constants of one VLB step, each rounded once from a double expression, as
    torch.where(cond, python_float, python_float) rounds them (sample_ppgn_vlb.py:91-100).
    order: [1-beta_t, beta_t, st1, 1-st1, st, 1-st]."""
    st, b, st1 = float(sigmatilde_t), float(sigma_t), float(sigmatilde_t1)
    return np.asarray([1 - b, b, st1, 1 - st1, st, 1 - st], dtype=np.float64).astype(np.float32)


def _posterior_mix(A_t, score_i, c):
    """p(A_{t-1}=1 | A_t) = (1-s)*post(x0=0) + s*post(x0=1); c = per-graph [B,6] float32 constants (vlb_scalars).
    Operation order of sample_ppgn_vlb.py:91-101 / ppgn_vlb.py:98-109."""
    B = A_t.shape[0]
    c = _t(c).view(B, 6, 1, 1)
    one_m_b, b, st1, one_m_st1, st, one_m_st = [c[:, k] for k in range(6)]
    on = A_t > 0.5
    mult1 = torch.where(on, one_m_b, b)
    div0 = torch.where(on, st, one_m_st)          # xor(A_t, 0) = A_t != 0
    p = (1 - score_i) * mult1 * st1 / div0
    div1 = torch.where(on, one_m_st, st)          # xor(A_t, 1) = A_t == 0
    p = p + score_i * mult1 * one_m_st1 / div1
    return p


def reverse_step_vlb(A_t, probs, flags, consts, U):
    """reference sample_ppgn_vlb.py:83-107 from post-sigmoid probabilities.  consts: [6] or [B,6] (vlb_scalars)."""
    A_t, r, U = _t(A_t), _t(probs), _t(U)
    B = A_t.shape[0]
    c = _t(consts).reshape(-1, 6)
    if c.shape[0] == 1:
        c = c.expand(B, 6)
    mask = generate_mask(flags)
    r = (r + r.transpose(-2, -1)) / 2
    score_i = torch.where(A_t > 0.5, 1 - r, r)
    p = _posterior_mix(A_t, score_i, c)
    init = (p + p.transpose(-2, -1)) / 2
    init = init * mask
    bern = (U < init).to(F32)
    return torch.triu(bern, diagonal=1) + torch.triu(bern, diagonal=1).transpose(-2, -1)


def sigma_lin(sigma_list):
    """reference ppgn_vlb.py:27-34 == sample_ppgn_vlb.py:28-35 (python doubles)."""
    out = []
    for g, s in enumerate(sigma_list):
        if s < 1.0e-5:
            out.append(0.0)
            continue
        out.append(((1 - s) - (1 - sigma_list[g - 1])) / (1 - 2 * (1 - sigma_list[g - 1])))
    return out


def schedule_simple(noise_num):
    """sample_ppgn_simple.py:45-47: ascending sigmas; the sampler walks them from the end."""
    s = torch.linspace(0, 1 / 2, noise_num).tolist()
    s.sort()
    return s


def schedule_vlb(noise_num):
    """sample_ppgn_vlb.py:67-70,122-126 -> list over steps (t = T..1) of (sigmatilde_t, beta_t, sigmatilde_{t-1})."""
    s = torch.linspace(0, 1 / 2, noise_num + 1).tolist()
    s.sort()
    beta = sigma_lin(s)
    n = len(s)
    return [(s[n - c - 1], beta[n - c - 1], s[n - c - 2]) for c in range(n - 1)]


def loss_bce(score, target, sigmas, mask):
    """reference ppgn_simple.py:25-34.  The weight tensor is float64 there (torch.tensor(list of np.float64)), so the
    loss is a float64 scalar; d loss / d score is float32."""
    score, target, mask = _t(score), _t(target), _t(mask)
    B = score.shape[0]
    lm = torch.nn.functional.binary_cross_entropy_with_logits(score, target, reduction="none")
    w = 1 - 2 * torch.tensor(np.asarray(sigmas, dtype=np.float64)).view(B, 1, 1) + 1.0 / B
    lm = lm * w
    lm = (lm + lm.transpose(-2, -1)) / 2
    lm = lm * mask
    return lm.mean()


def vlb_train_scalars(sigma_list, sigma_ind_list, sigma_nontild_list):
    """Per-graph constants of loss_func_kld (ppgn_vlb.py:85-87): sigmatilde_{t-1} = st - st / t."""
    out = []
    for st, t, b in zip(sigma_list, sigma_ind_list, sigma_nontild_list):
        st = float(st)
        st1 = st - st / float(t)
        out.append(vlb_scalars(st, float(b), st1))
    return np.stack(out)


def loss_vlb(score, A_t, A_0, consts, mask):
    """reference ppgn_vlb.py:79-125, vectorised over graphs.  consts [B,6] from vlb_train_scalars.  float32 sum."""
    score, A_t, A_0, mask = _t(score), _t(A_t), _t(A_0), _t(mask)
    B = score.shape[0]
    c = _t(consts).view(B, 6, 1, 1)
    one_m_b, b, st1, one_m_st1, st, one_m_st = [c[:, k] for k in range(6)]
    mult1 = torch.where(A_t > 0.5, one_m_b, b)
    mult2 = torch.where(A_0 > 0.5, one_m_st1, st1)
    xor = torch.logical_xor(A_t, A_0)
    div = torch.where(xor, st, one_m_st)
    q = mult1 * mult2 / div
    sg = torch.sigmoid(score)
    score_i = torch.where(A_t > 0.5, 1 - sg, sg)
    p = _posterior_mix(A_t, score_i, c.view(B, 6))
    p2 = torch.stack([p, 1 - p], -1)
    q2 = torch.stack([q, 1 - q], -1)
    lm = torch.nn.functional.kl_div(torch.log(p2), q2, reduction="none").sum(-1)
    lm = (lm + lm.transpose(-2, -1)) / 2
    lm = lm * mask
    lm = torch.triu(lm, diagonal=1) + torch.tril(lm, diagonal=-1)
    return lm.sum()
