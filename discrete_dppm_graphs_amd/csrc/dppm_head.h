// Read-out head of the H = 128 PPGN in ONE pass over the layer outputs (model/ppgn.py:318-333), split-bf16 arithmetic:
//   R      = layer_cat_lin(cat([A, sigma I, u_1 .. u_L]))      K = 2 + 128 L
//   logits = final_lin(R + LeakyReLU(after_conv(R)))            (dropout p = 1e-6 is the identity in eval)
// The layer-by-layer form (four two-source launches that re-read their accumulator, after_conv, final_lin) moves
// 15 + 2 + 2 plane sets per level for 8 (+1/64) algorithmic ones; here every u_l tile is read once, R / act never leave the CU.
//
// Organisation: one 8-wave workgroup per CU walks super-tiles of 256 positions (4 tiles of 64).  Wave w owns output channels
// 16w .. 16w+15.  Per super-tile the sources u_0 .. u_{L-1} are visited in turn; for each, the wave's weight fragments (32
// registers, pre-split into bf16 hi / lo pieces by head_wimg_kernel and fetched from L2 one source ahead) multiply the four X
// tiles, which stream global -> registers -> bf16 hi / lo images in LDS exactly as in dppm_wsplit.h (two tiles in flight per
// thread, one barrier per tile).  The products are issued as D = W X, so a lane holds 4 consecutive CHANNELS of one position
// and the finished R tile drops into a hidden image with 8-byte writes (the fused-MLP trick); after_conv then runs from that
// image with its own fragments (weight phase L of the same pipeline), and final_lin is a dot product over channels:
// registers -> 2 shuffles -> LDS partials of the 8 waves -> 64 threads store the logits of a tile.
// The X stream of the next super-tile stays in flight across this epilogue.
#pragma once
#include "dppm_common.h"

namespace dppm {

constexpr int HD_STG = 4;                     // staging waves
constexpr int HD_THREADS = 64 * (HD_STG + 8); // 4 staging waves + 8 multiplying waves (168 registers each), one workgroup per CU
#ifndef HD_NT_VAL
#define HD_NT_VAL 2
#endif
constexpr int HD_NT = HD_NT_VAL;                       // 64-position tiles per super-tile
constexpr int HD_SP = HD_NT * 64;              // positions per super-tile (64-position tiles, WS_TP)
constexpr int HD_WFRAG = 8 * 2 * 4 * 64;       // uint4 per weight phase in the fragment image: [wave][hi, lo][k step][lane]

struct HeadArgs {
  const float* cat; long cat_bs;               // CAT [B][2 + 128 L][P]: planes A, sigma I, u_1 .. u_L
  const uint4* wimg;                           // [L + 1][HD_WFRAG]: layer_cat_lin slices 0 .. L-1, then after_conv
  const float* cat_w; int ldw;                 // [128][2 + 128 L]: columns 0, 1 multiply the [A, sigma I] planes
  const float* cat_b; const float* after_b; const float* final_w; const float* final_b;
  float* logits;                               // [B][N][N]
  int B, N, ld, P, L;
  const float* aff; const float* maskp;        // nullable: [L][B][2][128] (scale, shift) of each layer's instance norm, mask planes [B][P]
};

// logits = head(CAT) for H = 128 (dppm_head.hip).  a.wimg: scratch of head3_wimg_floats(L) floats, 16-byte aligned.  ns: 3 = bf16x3, 1 = bf16.
inline size_t head3_wimg_floats(int L) { return (size_t)(L + 1) * HD_WFRAG * 4; }
int head3_run(const HeadArgs& a, const float* after_w, int ns, hipStream_t st);

}  // namespace dppm
