/*
 * dppm_hip.h -- C-ABI of libdppm_hip.so: the MI355X (gfx950) implementation of the discrete
 * graph-DDPM hot path of kilianhae/discrete_DPPM_Graphs.
 *
 * The reference has no FFI for this path (it is eager PyTorch); each entry point below names the reference
 * Python function (file:line under the reference tree) whose device work it replaces.  The Python host
 * side (discrete_dppm_graphs_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; buffers are owned by the caller;
 *     no entry point allocates or frees device memory, synchronises the device, or keeps state between calls;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued on it;
 *   - return value: 0 = ok, <0 = error (DPPM_E_*); dppm_last_error() gives a thread-local message;
 *   - public tensors are dense row-major float32: adjacency / mask / logits / uniforms [B][N][N], flags [B][N];
 *   - {0,1}-valued tensors (A_0, A_t, flags, mask, targets) are float32 as in the reference;
 *   - activations inside the PPGN workspace are channel-first planes [B][C][N][ld], ld = N rounded up to 4.
 *
 * Random numbers: every sampling entry point takes either explicit uniforms U[B][N][N] (what torch.bernoulli
 * would have drawn; only i<j is read) or, when U == NULL, draws them itself from Philox4x32-10 with
 *   u(b,i,j) = (philox(key = seed, ctr = (i, j, graph_id0 + b, offset)).x >> 8) * 2^-24 .
 * graph_id0 is the GLOBAL index of the shard's first graph, so samples do not depend on batch or GPU count.
 */
#ifndef DPPM_HIP_H
#define DPPM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPPM_OK 0
#define DPPM_E_INVALID (-1)   /* bad argument (null pointer, size <= 0, workspace too small, ...) */
#define DPPM_E_LAUNCH (-2)    /* HIP reported a launch / runtime error */
#define DPPM_E_UNSUPPORTED (-3)

/* Matrix arithmetic of the per-position linears (H = 128, and since round 3 H = 32: the wide kernels at N > 20 and the one-launch forward at N <= 20)
 * and of the per-channel contraction in INFERENCE forwards (operands and results are float32 in HBM
 * in every mode; kernels that have no reduced form use DPPM_PREC_F32 silently; training forward + backward always run
 * float32-grade arithmetic (DPPM_PREC_F32, and DPPM_PREC_F32X6 where a kernel has it): through the backward of the instance
 * norms the 2^-16 products of the two-piece split are amplified to ~50x the float32 gradient error, measured in
 * tests/test_gpu_ppgn.py::test_backward_kinkfree_h128):
 *   DPPM_PREC_F32     exact float32 products on v_mfma_f32_16x16x4_f32;
 *   DPPM_PREC_BF16X3  each operand split into two bf16 pieces (hi + lo), a product = hi*hi + hi*lo + lo*hi on
 *                     v_mfma_f32_16x16x32_bf16 with float32 accumulation: ~6e-6 of the result's magnitude per Linear,
 *                     inside the 1e-3 float32 parity bar, at 3/16 of the f32 matrix time;
 *   DPPM_PREC_BF16    operands rounded to bf16, float32 accumulation (~3e-3 per Linear).  An experiment switch for timing the
 *                     matrix side of the split kernels: NO tolerance is claimed for it and no parity test rests on it (through
 *                     1-3 layers of a random-init network it is 1.5e-1 of max |logit| off). */
#define DPPM_PREC_F32 0
#define DPPM_PREC_BF16X3 1
#define DPPM_PREC_BF16 2
/*   DPPM_PREC_F32X6   each operand split into three bf16 pieces (hi + mid + lo = all 24 significand bits), a product = the six
 *                     terms down to 2^-16 on v_mfma_f32_16x16x32_bf16, float32 accumulation: as close to float64 as exact
 *                     float32 is, at 6/16 of the f32 matrix time.  dppm_plane_matmul only (129 <= n <= 208, all three
 *                     forms); what training uses for the contraction and its two gradients.  Elsewhere = DPPM_PREC_F32. */
#define DPPM_PREC_F32X6 3

const char* dppm_last_error(void);
/* ABI version, bumped on any signature change. */
int dppm_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Diffusion ops (bit-exact float32 restatements; compiled with FP contraction off)
 * ---------------------------------------------------------------------------------------------- */

/* mask[b][i][j] = (flags[b][i] >= 0.9) * (flags[b][j] >= 0.9).
 * Replaces utils/graph_utils.py:312-319 generate_mask (a Python triple loop on the CPU). */
int dppm_generate_mask(const float* flags, int B, int N, float* mask, void* stream);

/* q(A_t | A_0):  p = A0 > 1/2 ? 1 - sigma : sigma ; up = bernoulli(p) for i<j ; A_t = sym(up) * f_i * f_j ;
 * target = |A_t_unmasked - A0| * f_i * f_j.   sigma_dev: [B] doubles on the device (a sigma per graph, as
 * gen_list_of_data_single does) or NULL to use sigma_scalar for every graph (as the samplers do).
 * (1 - sigma) is formed in double and rounded once to float, like torch.full_like(x, 1 - sigma).
 * target may be NULL.  A_t may alias A0.
 * Replaces utils/graph_utils.py:155-165 discretenoise, :236-246 discretenoise_single, :96-138
 * gen_list_of_data_single (the loop over graphs) and :7-19 mask_adjs. */
int dppm_edgeflip_fwd(const float* A0, const float* flags, const double* sigma_dev, double sigma_scalar,
                      const float* U, uint64_t seed, uint32_t graph_id0, uint32_t offset,
                      int B, int N, float* A_t, float* target, void* stream);

/* x_T = sym(triu(bernoulli(1/2 where both nodes active), 1)).
 * Replaces sample_ppgn_simple.py:31-41 / sample_ppgn_vlb.py:51-62 (gen_init_data's Python loop + bernoulli). */
int dppm_init_xT(const float* flags, const float* U, uint64_t seed, uint32_t graph_id0, uint32_t offset,
                 int B, int N, float* A_T, void* stream);

/* One reverse step of the "simple" sampler after the denoiser:
 *   r = sigmoid(logits) (or r = logits when is_probs != 0); r = (r + r^T)/2 * mask; F = sym(triu(bernoulli(r),1));
 *   out = A_t XOR F.            out may alias A_t.
 * Replaces sample_ppgn_simple.py:58-68 (take_step after the model call). */
int dppm_reverse_step_simple(const float* A_t, const float* logits, int is_probs, const float* flags,
                             const float* U, uint64_t seed, uint32_t graph_id0, uint32_t offset,
                             int B, int N, float* out, void* stream);

/* One reverse step of the VLB sampler after the denoiser (posterior mix):
 *   r = sym(sigmoid(logits)); s = A_t ? 1 - r : r;
 *   p = (1-s) * m1 * st1 / d0 + s * m1 * (1-st1) / d1,  m1 = A_t ? 1-beta_t : beta_t,
 *   d0 = A_t ? st : 1-st, d1 = A_t ? 1-st : st;  out = sym(triu(bernoulli(sym(p) * mask), 1)).
 * sigmatilde_t, beta_t, sigmatilde_t1 are the python doubles of the schedule; the six float constants are
 * formed in double and rounded once, as torch.where(cond, python_float, python_float) does.
 * Replaces sample_ppgn_vlb.py:82-107 (take_step after the model call). */
int dppm_reverse_step_vlb(const float* A_t, const float* logits, int is_probs, const float* flags,
                          double sigmatilde_t, double beta_t, double sigmatilde_t1,
                          const float* U, uint64_t seed, uint32_t graph_id0, uint32_t offset,
                          int B, int N, float* out, void* stream);

/* Workspace (bytes) for the two loss entry points below. */
size_t dppm_loss_workspace_bytes(int B, int N);

/* Simple loss, forward + gradient:  loss = sum_{b,i,j} sym(BCEwithLogits(score, target) * w_b) * mask / (B_global*N*N),
 * w_b = 1 - 2 sigma_b + 1/B_global.   loss_out: one double on the device.  dscore (nullable): d loss / d score.
 * B_global = number of graphs in the whole (possibly data-parallel) batch; this call handles B of them.
 * Replaces ppgn_simple.py:25-34 loss_func_bce (+ its autograd backward). */
int dppm_loss_bce_fwd_bwd(const float* score, const float* target, const double* sigma_dev, const float* mask,
                          int B, int N, int B_global, double* loss_out, float* dscore,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The same loss for callers that cannot allocate or pass a workspace (code under HIP graph capture, tight serving / training loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  The partial sums live in a file-scope device array of the
 * library (256 KiB per GPU, held until the process ends) instead of a caller workspace; the call makes no hipMalloc / hipFree /
 * allocating library call.  At most DPPM_NOALLOC_MAX_GRAPHS graphs per call (more: DPPM_E_INVALID, use the entry point above).
 * All calls on one GPU share that array: calls that overlap on one GPU (different streams, or threads) race; calls on one stream
 * are ordered and safe.  Results are bit-identical to dppm_loss_bce_fwd_bwd.
 * The FIRST call on each device looks up the array's device address (hipGetSymbolAddress, once per device, under a mutex): that lookup
 * is not a capturable operation, so make one call on the device before a stream capture begins; every later call only launches kernels. */
#define DPPM_NOALLOC_MAX_GRAPHS 4096
int dppm_loss_bce_fwd_bwd_noalloc(const float* score, const float* target, const double* sigma_dev, const float* mask,
                                  int B, int N, int B_global, double* loss_out, float* dscore, void* stream);

/* VLB loss, forward + gradient: sum over graphs and i != j of sym(KL(q || p)) * mask, with q the true posterior
 * q(A_{t-1}=1 | A_t, A_0) and p the model posterior mix (see dppm_reverse_step_vlb).
 * sched_dev: [B][3] doubles (sigmatilde_t, beta_t, sigmatilde_{t-1} = st - st/t) per graph.
 * loss_out: one float on the device.   Replaces ppgn_vlb.py:79-125 loss_func_kld (+ autograd backward). */
int dppm_loss_vlb_fwd_bwd(const float* score, const float* A_t, const float* A_0, const double* sched_dev,
                          const float* mask, int B, int N, float* loss_out, float* dscore,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PPGN ("Powerful") denoiser, configuration built by utils/loading_utils.py:98-119
 * ---------------------------------------------------------------------------------------------- */

/* Pointers to the used parameters, in the reference's own shapes (torch nn.Linear: weight [out][in] row-major).
 * State-dict names in comments.  The same struct (with writable buffers) receives the gradients. */
typedef struct dppm_ppgn_layer {
  float* m1_w0; float* m1_b0;   /* convs.{l}.m1.0.{weight,bias}  [H][H], [H] */
  float* m1_w1; float* m1_b1;   /* convs.{l}.m1.2.*              [H][H], [H] */
  float* m2_w0; float* m2_b0;   /* convs.{l}.m2.0.* */
  float* m2_w1; float* m2_b1;   /* convs.{l}.m2.2.* */
  float* m4_w;  float* m4_b;    /* convs.{l}.m4.0.*              [H][2H], [H] */
} dppm_ppgn_layer;

typedef struct dppm_ppgn_params {
  int32_t num_layers;           /* L */
  int32_t hidden;               /* H = hidden = hidden_final */
  float* in_w;  float* in_b;    /* in_lin.0.*         [H][2], [H] */
  float* cat_w; float* cat_b;   /* layer_cat_lin.0.*  [H][L*H+2], [H] */
  float* after_w; float* after_b; /* after_conv.0.*   [H][H], [H] */
  float* final_w; float* final_b; /* final_lin.0.*    [1][H], [1] */
  const dppm_ppgn_layer* layers_host; /* HOST array of L entries (entries hold device pointers) */
  /* Per-layer noise conditioning of the Powerful_Ind / Powerful_Ind_Mult variants (model/ppgn_indnoise.py:306-315,
   * model/ppgn_indnoise_multdim.py: the learned scalar of layer l on the diagonal of an extra input channel).  Nullable.
   * [L][B] floats on the device: before layer l runs, layer_cond[l][b] is written on the whole diagonal of channel H-1 of its
   * INPUT planes.  The caller passes the variant's (H-1)-wide matrices zero-padded to H (pad row / column H-1, zero bias), so
   * that channel is identically zero everywhere else.  In a grads struct handed to dppm_ppgn_backward the same field is the
   * OUTPUT buffer [L][B] that receives d/d layer_cond (required when the params struct carries layer_cond; the gradients of
   * the pad rows / columns of the padded matrices are meaningless and are to be dropped by the caller). */
  const float* layer_cond;
} dppm_ppgn_params;

/* Bytes of workspace dppm_ppgn_forward needs; with training != 0 it also keeps what dppm_ppgn_backward reads. */
size_t dppm_ppgn_workspace_bytes(int B, int N, int H, int L, int training);

/* logits[B][N][N] = Powerful.forward(node_features (ignored), A, mask, noiselevel) in eval() mode.
 * sigma: [B] floats on the device (the reference passes one python float per call; a vector lets a whole
 * training batch run in one call).  mask: [B][N][N], binary, node 0 of every graph active (the reference divides
 * by sqrt(sum_j mask[b][0][j]), model/ppgn.py:51).
 * precision: DPPM_PREC_* (see above); ignored (exact float32) when training != 0.
 * Inference with hidden = 32, N <= 20 and no layer_cond (the reference's ego_small / community_small networks) runs as ONE launch,
 * one workgroup per graph, exact float32, and leaves the workspace untouched (csrc/dppm_smallgraph.hip).
 * Replaces model/ppgn.py:257-337 Powerful.forward/forward_cat, :47-71 PowerfulLayer.forward and
 * util/model_helper.py:176-191 masked_instance_norm2D. */
int dppm_ppgn_forward(const dppm_ppgn_params* params_host, const float* A, const float* mask, const float* sigma,
                      int B, int N, float* logits, void* workspace, size_t workspace_bytes, int training,
                      int precision, void* stream);

/* Gradients of sum(logits * dlogits) w.r.t. every parameter in *grads_host (same layout as params; every
 * buffer is overwritten).  `workspace` must be the one a dppm_ppgn_forward(training=1) call on the same inputs
 * filled.  dsigma (nullable): [B], receives the gradient w.r.t. sigma (needed only when a learned noise_mlp produces sigma,
 * model/ppgn.py:273-274).
 * Replaces torch autograd over model/ppgn.py:257-337. */
int dppm_ppgn_backward(const dppm_ppgn_params* params_host, const dppm_ppgn_params* grads_host,
                       const float* A, const float* mask, const float* sigma, int B, int N,
                       const float* dlogits, float* dsigma, void* workspace, size_t workspace_bytes, void* stream);

/* A whole sampling chain of small graphs in ONE launch -- one workgroup per graph; adjacency, activations and logits stay in
 * registers and LDS, only x_out [B][N][N] is written.  Same Philox counters and float expressions as the per-level entry points:
 * x_out is bit-identical to calling them level by level at the same precision.
 *   dppm_ppgn_chain_simple: x_T (dppm_init_xT, draw offset 0), then for level c = 0 .. T-1 the re-noising of the running estimate
 *     (dppm_edgeflip_fwd, offset 1 + 2 c), the denoiser (dppm_ppgn_forward, inference), the reverse step (dppm_reverse_step_simple,
 *     offset 2 + 2 c).  level_table: T x 3 floats ON THE DEVICE, row c = { (float)sigma_c : the network's noise input,
 *     (float)(1.0 - sigma_c), (float)sigma_c : keep / flip probabilities, rounded from double as dppm_edgeflip_fwd does }.
 *     Replaces the loop of sample_ppgn_simple.py:70-88 (48 launches for 16 levels).
 *   dppm_ppgn_chain_vlb: x_T, then per level the denoiser on A_t and dppm_reverse_step_vlb (offset 1 + c).  level_table: T x 7
 *     floats on the device, row c = { (float)sigmatilde_t, then (float) of 1 - beta_t, beta_t, sigmatilde_{t-1}, 1 - sigmatilde_{t-1},
 *     sigmatilde_t, 1 - sigmatilde_t : the constants dppm_reverse_step_vlb rounds from its three doubles }.
 *     Replaces the loop of sample_ppgn_vlb.py:111-131.
 * Only where dppm_ppgn_chain_supported() returns 1 (hidden 32, N <= 20, <= 16 layers, no layer_cond: the reference's
 * community_small / ego_small networks); otherwise DPPM_E_INVALID, and the chain is run through the per-level entry points. */
int dppm_ppgn_chain_supported(const dppm_ppgn_params* params_host, int N);
int dppm_ppgn_chain_simple(const dppm_ppgn_params* params_host, const float* flags, const float* mask, const float* level_table,
                           int T, uint64_t seed, uint32_t graph_id0, int B, int N, float* x_out, int precision, void* stream);
int dppm_ppgn_chain_vlb(const dppm_ppgn_params* params_host, const float* flags, const float* mask, const float* level_table,
                        int T, uint64_t seed, uint32_t graph_id0, int B, int N, float* x_out, int precision, void* stream);

/* Building blocks, exported for tests and for callers that schedule layers themselves. */

/* C[g] = epilogue(A[g] (MxK) * B[g] (KxN)) for g in [0,batch): the per-channel adjacency contraction
 * (model/ppgn.py:62 `out1 @ out2`), planes [N][ld].  trans_a / trans_b select A^T / B^T (backward).
 * scale (nullable): [batch / scale_div] multiplies the product; mask (nullable): [batch / scale_div][n][ld]
 * multiplies it elementwise.  DPPM_PREC_BF16X3 / DPPM_PREC_BF16 apply to the forward form (no transposes) with 129 <= n <= 1024,
 * DPPM_PREC_F32X6 to every form with 129 <= n <= 208; other cases run exact float32. */
int dppm_plane_matmul(const float* A, const float* Bm, float* C, int batch, int n, int ld, int trans_a, int trans_b,
                      const float* scale, const float* mask, int scale_div, int precision, void* stream);

/* Masked instance norm over each [N][ld] plane (util/model_helper.py:176-191 + the `u * mask` of
 * model/ppgn.py:315).  y, u: [B][C][N*ld] with batch strides y_bs / u_bs (floats); maskp: [B][N*ld];
 * mean_out / rstd_out (nullable): [B][C]. */
int dppm_inorm_fwd(const float* y, long y_bs, const float* maskp, float* u, long u_bs, int B, int C, int P,
                   float* mean_out, float* rstd_out, void* stream);

/* Y[b] (M x P) = epilogue(W (M x K) * X[b] (K x P)): one nn.Linear of the reference applied at every (i,j), in
 * channel-first planes (model/ppgn.py:58-59,69,301,327,331).  W is torch's [out][in] weight with row pitch ldw;
 * trans_w != 0 applies W^T instead (W is then [K][M]: the data-gradient form).  bias (nullable) [M]; act != 0 applies
 * LeakyReLU(0.01); maskp (nullable) [B][P] multiplies the result; accumulate != 0 adds into Y.
 * x_bs / y_bs: floats between consecutive graphs. */
int dppm_linear_fwd(const float* W, int ldw, int trans_w, const float* X, long x_bs, int K, const float* bias, int act,
                    const float* maskp, float* Y, long y_bs, int M, int B, int P, int accumulate, void* stream);

/* The two per-position MLPs of one PPGN layer in one pass over X (model/ppgn.py:58-59):
 *   o_k[b] = (W_k1 * LeakyReLU(W_k0 * X[b] + b_k0) + b_k1) * maskp[b],  k = 1, 2      (all matrices H x H, planes H x P).
 * hid1 / hid2 (nullable, both or none): the hidden activations, written only when a backward pass will need them.
 * H = 128 runs the weight-stationary fused kernel (hidden activations stay in LDS); other H use four GEMM launches and
 * then need hid1 / hid2 as scratch (not nullable; H = 32 with 16-byte aligned planes runs its own wide kernel).  precision != DPPM_PREC_F32
 * applies to H = 128 and H = 32 without saved hidden activations. */
int dppm_mlp_pair_fwd(const float* X, long x_bs, const dppm_ppgn_layer* layer_host, int H, const float* maskp,
                      float* o1, float* o2, float* hid1, float* hid2, int B, int P, int precision, void* stream);

/* dW (M x Nw, row pitch ldo) = sum_b dY[b] (M x P) * X[b]^T (P x Nw), db (nullable) [M] = sum_{b,p} dY[b][m][p]:
 * the weight/bias gradient of the Linear above (split-K partial slabs + ordered reduction, bitwise reproducible).
 * workspace: dppm_linear_wgrad_workspace_bytes(M, Nw, B, P). */
size_t dppm_linear_wgrad_workspace_bytes(int M, int Nw, int B, int P);
int dppm_linear_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, int M, int Nw, int B, int P, float* dW,
                      int ldo, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser
 * ---------------------------------------------------------------------------------------------- */

/* One fused Adam step over a flat parameter buffer (torch.optim.Adam semantics: betas, eps, L2 weight decay
 * added to the gradient, bias correction with `step` starting at 1).  Replaces optimizer.step() at
 * ppgn_simple.py:93 / ppgn_vlb.py:198 (optim.Adam built at ppgn_simple.py:259-262). */
int dppm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * Graph statistics for model selection (the step after sampling; the reference does it with networkx / pyemd on the CPU)
 * ---------------------------------------------------------------------------------------------- */

/* Degree and clustering-coefficient histograms of B graphs given as dense adjacencies adj[B][N][N] (entries > 0.5 are edges,
 * the diagonal is ignored, isolated nodes do not belong to the graph and an edgeless graph counts as one isolated node,
 * as evaluation/stats.py:276-286 adjs_to_graphs builds them).  deg_hist[B][N]: number of nodes of degree d
 * (nx.degree_histogram, zero-padded to N entries); clus_hist[B][bins]: numpy.histogram(nx.clustering(G).values(),
 * bins, range=(0, 1)) with edges_dev = the bins+1 float64 edges numpy.linspace(0, 1, bins+1) gives (device array).
 * N <= 1024.  Replaces the histogram loops of evaluation/stats.py:41-69 (degree_stats) and :140-186 (clustering_stats). */
int dppm_graph_stats(const float* adj, int B, int N, int bins, const double* edges_dev, float* deg_hist, float* clus_hist,
                     void* stream);

/* MMD^2 between two sets of histograms (rows of h1[n1][bins], h2[n2][bins]; raw counts, normalised here) with the
 * Gaussian-EMD kernel k = exp(-emd^2 / (2 sigma^2)), emd = first Wasserstein distance with ground distance
 * |i - j| / distance_scaling.  out4 (device doubles): {mmd, mean k(1,1), mean k(2,2), mean k(1,2)}.
 * Replaces evaluation/mmd.py:90-106 compute_mmd with :32-41 gaussian_emd (pyemd on the toeplitz distance matrix). */
size_t dppm_mmd_workspace_bytes(int n1, int n2);
int dppm_mmd_gaussian_emd(const float* h1, int n1, const float* h2, int n2, int bins, double sigma, double distance_scaling,
                          double* out4, void* workspace, size_t workspace_bytes, void* stream);

/* Measurement and test hooks (per-launch timing, LDS poison, experiment switches) are NOT part of this interface: include/dppm_hip_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* DPPM_HIP_H */
