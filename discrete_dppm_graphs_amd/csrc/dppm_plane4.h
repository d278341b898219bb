// Role-split persistent form of the split-bf16 per-channel contraction for 129 <= N <= 512 (dppm_plane4.hip).
#pragma once
#include "dppm_planemm.h"

namespace dppm {

// C[g] = A[g] * B[g] * scale[g / div] (* mask) for g < batch, planes [n][ld] (model/ppgn.py:62-64).  ns: 3 = bf16x3, 1 = bf16.
bool plane4_ok(const PlaneArgs& a, int batch);
int plane4_run(const PlaneArgs& a, int batch, int ns, hipStream_t st);

}  // namespace dppm
