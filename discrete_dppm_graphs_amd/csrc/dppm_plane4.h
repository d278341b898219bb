// Role-split persistent form of the split-bf16 per-channel contraction for 129 <= N <= 512 (dppm_plane4.hip).
#pragma once
#include "dppm_planemm.h"

namespace dppm {

// C[g] = op(A[g]) * op(B[g]) * scale[g / div] (* mask) for g < batch, planes [n][ld] (model/ppgn.py:62-64 and its backward forms).
// ns: 3 = bf16x3, 1 = bf16 (forward form only); 6 = three bf16 pieces, six products: float32-equivalent, ta / tb allowed (not both).
bool plane4_ok(const PlaneArgs& a, int batch, int ns, bool ta, bool tb);
int plane4_run(const PlaneArgs& a, int batch, int ns, bool ta, bool tb, hipStream_t st);

}  // namespace dppm
