/* Measurement and test hooks of libdppm_hip.so -- NOT part of the drop-in interface (include/dppm_hip.h), no counterpart in the reference.
 * bench.py, tests/ and tools/ are the only callers; a product caller never needs them.  Process-global state, one caller at a time.
 */
#ifndef DPPM_HIP_DEBUG_H
#define DPPM_HIP_DEBUG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Per-launch timing.  dppm_prof_begin arms HIP-event timing of the matrix-family and norm kernels (events are recorded on the stream
 * each kernel is launched on); dppm_prof_end waits for them and returns tables [nclass][narith] (row-major, host memory) of summed
 * device time [ms], summed algorithmic work (FLOPs; bytes for the norm class), summed algorithmic HBM bytes (every distinct operand
 * read once, every result written once; 0 for the generic GEMM class) and launch counts.
 *   class: 0 per-position linears on the generic GEMM, 1 per-channel N x N contraction, 2 weight-gradient GEMMs, 3 instance norm,
 *          4 / 5 / 6 weight-stationary kernels (fused MLP pair / skip-concat Linear / single Linear), 7 fused read-out head  (nclass >= 8)
 *   arith: the arithmetic the launch ran in: 0 exact f32 MFMA, 1 bf16x3, 2 bf16, 3 f32x6 (the DPPM_PREC_* values), 4 none  (narith >= 5)
 * so that no roofline is formed over launches of two different pipes.
 * dppm_prof_end returns 1 if more than max_records launches happened (totals then cover the first max_records only). */
int dppm_prof_begin(int max_records);
int dppm_prof_end(double* ms_host, double* work_host, double* bytes_host, long* launches_host, int nclass, int narith);

/* 2048 workgroups each overwrite 160 KiB of LDS with `pattern`, so that a following kernel that reads LDS it never wrote sees that
 * pattern (tests poison the LDS with NaN bits). */
int dppm_test_fill_lds(uint32_t pattern, void* stream);

/* Experiment switches.  Each selects an older kernel for the same arithmetic (DISABLE_W4, W3MLP, DISABLE_NORMFUSE, DISABLE_HEAD3,
 * DISABLE_INLINFUSE, DISABLE_PLANE4, DISABLE_PLANE6, DISABLE_W6, DISABLE_W6MLP, DISABLE_MLP32, DISABLE_LIN32, DISABLE_WGRAD32,
 * DISABLE_STRIPMM, DISABLE_WIDE32, DISABLE_MLP32S, DISABLE_SMALLGRAPH, DISABLE_SGCHAIN, DISABLE_WSTAT, WGRAD_F32, PLANE_STRIPS_ONLY) or a timing experiment with WRONG results
 * (DBG, DBG_P4, DBG_W4).  The library reads them once per process from the environment (DPPM_<NAME>; unset or 0 = off) into one
 * table; afterwards these two calls are the only way to change or read one.  Unknown names return DPPM_E_INVALID. */
int dppm_debug_set(const char* name, int value);
int dppm_debug_get(const char* name, int* value);

#ifdef __cplusplus
}
#endif
#endif /* DPPM_HIP_DEBUG_H */
