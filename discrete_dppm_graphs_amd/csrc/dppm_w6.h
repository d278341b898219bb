// Float32-grade (three bf16 pieces, six products) role-split linears of the H = 128 network for training (dppm_w6.hip).
#pragma once
#include "dppm_wstat.h"

namespace dppm {

// Y (+)= epilogue(W1 X + b1) for every job (bias, LeakyReLU, mask, accumulate as ws_lin); trans applies the transposed matrices (data
// gradients).  nsrc must be 1 (two matrices in three pieces do not fit a wave's registers); no side input, no folded norm.
bool w6_lin_ok(const WsArgs& a, int nsrc);
// job.Hsaved (nullable): the result is multiplied by LeakyReLU'(Hsaved) before it is stored (backward through an activation).
// prof_class: measurement class of the launch (default: the transposed / single-source linears).
int w6_lin_run(const WsArgs& a, int nsrc, bool trans, hipStream_t st, int prof_class = -1);

}  // namespace dppm
