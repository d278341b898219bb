// Whole-graph-per-workgroup PPGN forward for small graphs (dppm_smallgraph.hip): N <= 20, hidden 32, inference.
#pragma once
#include "dppm_common.h"

namespace dppm {

constexpr int SG_MAX_NODES = 20;    // 32 channels x N x round_up(N, 4) floats per operand of the contraction must fit LDS twice
constexpr int SG_MAX_LAYERS = 16;   // layer pointer table travels in the kernel arguments

bool sg_supported(const dppm_ppgn_params* pr, int N, int training);
// logits[B][N][N] of dppm_ppgn_forward (eval mode, exact float32); no workspace.  Operands as in include/dppm_hip.h.
int sg_forward(const dppm_ppgn_params* pr, const float* A, const float* mask, const float* sigma, int B, int N, float* logits,
               hipStream_t st);

}  // namespace dppm
