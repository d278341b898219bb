// Role-split (staging waves / multiplying waves) forms of the split-bf16 H = 128 linears for inference (dppm_w4.hip).
#pragma once
#include "dppm_wstat.h"

namespace dppm {

// Two-source Linear Y = W1 X + W2 X2 + b1 of job 0 (m4 of model/ppgn.py:66-69), optionally with the fused instance norm
// (a.aff on X2, a.stat / a.cnt / a.maskstat for the result's tile statistics).  ns: 3 = bf16x3, 1 = bf16.
bool w4_lin2_ok(const WsArgs& a);
int w4_lin2_run(const WsArgs& a, int ns, hipStream_t st);

// Fused per-position MLP (Linear -> LeakyReLU -> Linear, * mask) for every job (model/ppgn.py:58-59), hidden tile in LDS;
// a.aff (nullable): the source's instance norm, applied while staging.
bool w4_mlp_ok(const WsArgs& a);
int w4_mlp_run(const WsArgs& a, int ns, hipStream_t st);

}  // namespace dppm
