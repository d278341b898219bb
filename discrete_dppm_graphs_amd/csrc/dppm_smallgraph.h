// Whole-graph-per-workgroup PPGN forward for small graphs (dppm_smallgraph.hip): N <= 20, hidden 32, inference.
#pragma once
#include "dppm_common.h"

namespace dppm {

constexpr int SG_MAX_NODES = 20;    // 32 channels x N x round_up(N, 4) floats per operand of the contraction must fit LDS twice
constexpr int SG_MAX_LAYERS = 16;   // layer pointer table travels in the kernel arguments

// Training: where dppm_ppgn_backward expects what the forward saves (pointers into the training workspace, dppm_ppgn.hip Layout):
// plane sets [B][32][P] (u0, R, act), per-layer plane sets at l * S (hid1, hid2, o1, o2, mm), CAT [B][2 + 32 L][P], rstd [L][B][32].
struct SgSave {
  float *maskp, *inv_norm, *cat, *u0, *hid1, *hid2, *o1, *o2, *mm, *rstd, *R, *act;
  size_t S;
};

bool sg_supported(const dppm_ppgn_params* pr, int N);
// logits[B][N][N] of dppm_ppgn_forward (exact float32).  save == nullptr: inference, no workspace is touched; otherwise the
// activations the backward pass reads are written as well.  Operands as in include/dppm_hip.h.
int sg_forward(const dppm_ppgn_params* pr, const float* A, const float* mask, const float* sigma, int B, int N, float* logits,
               const SgSave* save, hipStream_t st, bool split = false);

// A sampling chain, all T levels in one launch (x_T, then per level: simple = re-noise, forward, reverse step; vlb = forward,
// posterior reverse step; bit-identical to the per-level kernels of dppm_diffusion.hip around sg_forward).  level_tab: [T][3]
// (simple) or [T][7] (vlb) floats on the device, see include/dppm_hip.h.
int sg_chain(bool vlb, const dppm_ppgn_params* pr, const float* flags, const float* mask, const float* level_tab, int T, uint64_t seed,
             uint32_t graph_id0, int B, int N, float* x, hipStream_t st, bool split);

}  // namespace dppm
