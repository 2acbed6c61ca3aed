/*
 * emlight_hip.h -- C ABI of libemlight_hip.so, the MI355X (gfx950) hot path of EMLight.
 *
 * Drop-in boundary.  Every entry point replaces a chain of ATen
 * ops in the reference's Python (the reference has no native code of its own); the
 * reference file:line each one stands in for is cited on the declaration.  A reference
 * maintainer binds these with ctypes -- plain pointers and sizes,
 * no torch types.
 *
 * Conventions
 *   - all tensors are contiguous f32 device buffers owned by the caller (PyTorch);
 *     the library allocates nothing and keeps no global mutable state;
 *   - every launcher only ENQUEUES work on `stream` (a hipStream_t passed as void*);
 *     no host synchronisation, no host reads of device data;
 *   - return value: 0 on success, otherwise a negative EML_E* code or a positive
 *     hipError_t; eml_last_error() returns a thread-local message for the last failure;
 *   - launchers never throw and never exit().
 */
#ifndef EMLIGHT_HIP_H
#define EMLIGHT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* eml_stream_t; /* hipStream_t */

#define EML_OK 0
#define EML_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define EML_ELAUNCH (-2)  /* kernel launch failed, see eml_last_error()    */

#define EML_MAX_EPS 64    /* capacity of an epsilon schedule buffer (floats) */

/* Library ABI version (bumped on any signature change; the ctypes binding refuses a library built from another
 * version of this header) and last-error text. */
#define EML_ABI_VERSION 29
int eml_abi_version(void);
const char* eml_last_error(void);

/* ---------------------------------------------------------------- SG rasteriser
 * lights[b,c,h,w] = sum_i colors[b,3i+c] * exp((dirs[b,3i:3i+3] . xyz[:,h,w] - 1) / sizes[b,i])
 * with xyz the W = 2H equirect view-vector grid.
 * Replaces convert_to_panorama: RegressionNetwork/util.py:222-245 (copies:
 * representation/util.py:205-228, GenProjector/util.py:346-369; variable latitude:
 * RegressionNetwork/panorama.py:68-82,142-152).
 * dirs (B,3N), sizes (B,N), colors (B,3N) -> out (B,3,H,W); W must equal 2H. */
int eml_sg_rasterise_f32(const float* dirs, const float* sizes, const float* colors,
                         float* out, int B, int N, int H, int W, eml_stream_t stream);

/* The same launch with options (tests and the bench): flags EML_SG_EXHAUSTIVE = evaluate every light for every pixel
 * (no cull at all: the reference's loop, util.py:239-244, in its order -- the culled kernel must match it bit for bit);
 * executed_exp (device, may be NULL) += the number of exponentials the launch evaluated (the reference: B*N*H*W). */
#define EML_SG_EXHAUSTIVE 1
int eml_sg_rasterise_ex_f32(const float* dirs, const float* sizes, const float* colors, float* out,
                            int B, int N, int H, int W, int flags, unsigned long long* executed_exp,
                            eml_stream_t stream);

/* d loss / d colors of the rasteriser (autograd of util.py:239-244 wrt `colors`):
 * gcolors[b,3i+c] = sum_hw gout[b,c,h,w] * exp((dirs_i . xyz_hw - 1) / sizes_i). */
int eml_sg_rasterise_bwd_colors_f32(const float* dirs, const float* sizes, const float* gout,
                                    float* gcolors, int B, int N, int H, int W,
                                    eml_stream_t stream);
/* The same gradient with the forward's hierarchical cull (same tiles, bounding caps and survivor lists): exponentials are
 * evaluated only for the lights that reach a wave's 16 x 8 patch; per-tile partial sums in `work`
 * (eml_sg_rasterise_bwd_work_floats(B, N, H, W) floats) are added in tile order -- no atomics, run-to-run exact.  flags:
 * EML_SG_EXHAUSTIVE = every light for every tile (a culled light contributes exactly 0, so both give the same bits; tests). */
size_t eml_sg_rasterise_bwd_work_floats(int B, int N, int H, int W);
int eml_sg_rasterise_bwd_colors_ex_f32(const float* dirs, const float* sizes, const float* gout, float* gcolors, float* work,
                                       int B, int N, int H, int W, int flags, eml_stream_t stream);

/* ---------------------------------------------------------------- Sinkhorn (spherical mover's loss)
 * Chord-length ground cost M_ij = ||a_i - a_j||_2 over N anchors (N x 3, f32).
 * Replaces the N^2 Python torch.norm loop of distance.__init__,
 * RegressionNetwork/geomloss/utils.py:65-76. */
int eml_emd_anchor_cost_f32(const float* anchors, float* M, int N, eml_stream_t stream);

/* Epsilon schedule on the device (no .item() host sync):
 *   d     = diameter > 0 ? diameter : max(x U y) - min(x U y)        (n values each)
 *           range_lo_hi (device, 2 floats, may be NULL): (min, max) of x U y over the OTHER data-parallel ranks'
 *           shards, all-reduced by the caller -- folded into the scan, so that every rank derives the schedule of the
 *           GLOBAL batch like a single-process run (max_diameter, sinkhorn_divergence.py:9-18, takes the range over the
 *           whole batch); no host sync either
 *   eps_s = [d^p] + [exp(e) for e in arange(p ln d, p ln blur, p ln scaling)] + [blur^p]
 * evaluated in f64 exactly as numpy does, stored as f32.
 * Replaces max_diameter / scaling_parameters / epsilon_schedule,
 * RegressionNetwork/geomloss/sinkhorn_divergence.py:9-36.
 * Outputs: eps_out[EML_MAX_EPS], *n_eps_out (device int), *diameter_out (device float). */
int eml_sinkhorn_schedule_f32(const float* x, const float* y, long n, double blur,
                              double scaling, int p, double diameter, const float* range_lo_hi,
                              float* eps_out, int* n_eps_out, float* diameter_out, eml_stream_t stream);

/* Whole debiased Sinkhorn divergence in one call (loop kernel + finishing kernel): the diameter and the
 * epsilon schedule (as eml_sinkhorn_schedule_f32, computed inside the loop kernel), cost build
 * C = .5*(.1*(x_i-y_j)^2 + M_ij) (never materialised in HBM), init sweep, n_eps symmetrised eps-scaling
 * sweeps, last extrapolation, loss_b = <alpha,b_x-a_x> + <beta,a_y-b_y>, and the analytic gradients
 * d loss_b/d x, d loss_b/d y of the last extrapolation.
 * Replaces SamplesLoss.sinkhorn_tensorized: geomloss/samples_loss.py:79-92 with
 * sinkhorn_divergence.py:9-36 (schedule), utils.py:85-99 (cost), samples_loss.py:75-77 (softmin),
 * sinkhorn_divergence.py:72-109 (loop), :65-69 (cost), and the autograd backward of :102-107.
 *   x, y        (B,N)  1-D "points" (the mass at each anchor)
 *   M, Mt       (N,N)  ground cost and its transpose (may alias when M is symmetric)
 *   alpha, beta (B,N)  weights, or NULL for uniform 1/N
 *   blur, scaling, p, diameter   as SamplesLoss(...); diameter <= 0: range of x U y over the batch
 *   range_lo_hi (2), device, or NULL   see eml_sinkhorn_schedule_f32 (diameter over the global batch under DDP)
 *   eps_out[EML_MAX_EPS], n_eps_out, diameter_out   device outputs of the schedule, or NULL
 *   loss        (B)
 *   gx, gy      (B,N)  d loss_b / d x_i, d loss_b / d y_j, or NULL
 *   work        caller-owned scratch of eml_sinkhorn_work_floats(B,N) floats (16-byte aligned); on return its first
 *                      (4,B,N) floats hold the final duals a_x, b_y, a_y, b_x (the next (4,B,N) the expectation rows of
 *                      the gradient; the rest is the inter-workgroup exchange buffer of the small-batch kernel, zeroed
 *                      by the call itself, and the status word described below) */
size_t eml_sinkhorn_work_floats(int B, int N);
int eml_sinkhorn_fwd_f32(const float* x, const float* y, const float* M, const float* Mt,
                         const float* alpha, const float* beta, double blur, double scaling, int p,
                         double diameter, const float* range_lo_hi, float* eps_out, int* n_eps_out,
                         float* diameter_out, float* loss, float* gx, float* gy, float* work, int B, int N,
                         eml_stream_t stream);

/* The same call with options.  Small batches at N >= 192 (N % 64 == 0, N <= 512) run the SPLIT kernel: a sample's rows
 * over S = 8 (or 4) workgroups that exchange the dual vectors through global memory after every sweep, which needs the S
 * slices of a (sample, role) resident together.  The launcher takes that path only when every workgroup has a CU of its own
 * among the CUs the STREAM may use (its CU mask or ROC_GLOBAL_CU_MASK, hipExtStreamGetCUMask); since residency still cannot be
 * guaranteed from the host (another stream, another process on the device), a slice that does not see its partners within
 * 50 ms raises a status word, and the tiled kernel -- enqueued behind the split kernel by the same call, gated on that
 * word -- recomputes the batch: the outputs are always those of a completed Sinkhorn loop, never NaN.
 *   flags   EML_SINKHORN_NO_SPLIT    never take the split path (a caller that saw the status word raised)
 *           EML_SINKHORN_FORCE_SPLIT take it whenever the shape allows, without the sizing (tests)
 *           EML_SINKHORN_TEST_STALL  (tests) one slice of every group withholds its exchange granules, as if it were not
 *                                    resident: its partners must give up and the rescue must recompute the batch
 *   status  = ((int*)work)[24*B*N]: set to 1 by a call whose split kernel gave up (the rescue ran); zeroed at the start
 *           of every call that takes the split path, untouched by calls that do not. */
#define EML_SINKHORN_NO_SPLIT 1
#define EML_SINKHORN_FORCE_SPLIT 2
#define EML_SINKHORN_TEST_STALL 4
int eml_sinkhorn_fwd_ex_f32(const float* x, const float* y, const float* M, const float* Mt,
                            const float* alpha, const float* beta, double blur, double scaling, int p,
                            double diameter, const float* range_lo_hi, float* eps_out, int* n_eps_out,
                            float* diameter_out, float* loss, float* gx, float* gy, float* work, int B, int N,
                            int flags, eml_stream_t stream);

/* Backward of the loss vector: gout[b,i] = gloss[b] * gunit[b,i]  (gunit = gx or gy above). */
int eml_sinkhorn_bwd_f32(const float* gloss, const float* gunit, float* gout, int B, int N,
                         eml_stream_t stream);


/* ---------------------------------------------------------------- DenseNet-BC encoder, forward
 * All activations are pixel-major (NHWC) f32: a dense block is ONE buffer X[P][ld] whose channel
 * axis is the concatenation (torch.cat of DenseNet.py:55 becomes a pointer offset).  Train-mode
 * BatchNorm statistics are emitted by each producer as per-workgroup f64 partial (sum, sumsq)
 * pairs, `grid` rows of them, and folded by eml_dense_bn_prepare_f32 -- no atomics.
 * `grid` = number of persistent workgroups to launch (the caller sizes `partials` by it). */

/* conv0: 3 -> C0 (=24) 3x3 pad 1 on the NCHW input (DenseNet.py:88-90), raw output into
 * X[:, 0:C0] (ld = ldx); partials [grid][C0][2]. */
int eml_dense_conv0_fwd_f32(const float* x, const float* w0, float* X, int ldx, int B, int H, int W,
                            int C0, double* partials, int grid, eml_stream_t stream);
/* the same layer on the matrix unit (ABI 29; the engine's default, EML_CONV0_MFMA=0: the entry above): the 3x3x3
 * neighbourhood of 16 pixels as the B operand of 14 v_mfma_f32_16x16x4_f32 (K = 27 of 28), one 4-byte load per lane and step;
 * X 16-byte aligned.  Outputs AND partials are bit for bit those of eml_dense_conv0_fwd_f32: the MFMA adds its k terms in
 * order (that kernel's fma chain) and the sums are formed in the association of its wave reduction. */
int eml_dense_conv0_fwd_mfma_f32(const float* x, const float* w0, float* X, int ldx, int B, int H, int W,
                                 int C0, double* partials, int grid, eml_stream_t stream);

/* dst[p][c] = act(scale[c]*src[p][c] + shift[c]), c < C (norm0+relu0, DenseNet.py:91-92; last_norm,
 * DenseNet.py:122); partials [grid][C][2] = stats of dst. */
int eml_dense_bn_apply_f32(const float* src, int ld_src, float* dst, int ld_dst, int C, long P,
                           const float* scale, const float* shift, int relu, double* partials,
                           int grid, eml_stream_t stream);

/* Fold partial stats of n_new fresh channels [c_new0, c_new0+n_new) into mean/var/istd (biased var,
 * istd = 1/sqrt(var+eps)), then (if scale != NULL) write the BatchNorm2d affine for channels
 * [0, C): scale = gamma*istd, shift = beta - mean*scale (zeros up to Cpad), and update the running
 * statistics with `momentum` (unbiased var) when training; eval uses the running buffers
 * (nn.BatchNorm2d semantics; DenseNet.py:17,29,40,91,122). */
int eml_dense_bn_prepare_f32(const double* partials, int G, int pstride, int n_new, int c_new0,
                             double count, float* mean, float* var, float* istd, const float* gamma,
                             const float* beta, float* rmean, float* rvar, int C, int Cpad, float eps,
                             float momentum, int training, float* scale, float* shift,
                             eml_stream_t stream);

/* Weight re-layouts into MFMA fragment order (done once per step):
 * W [Cout][Cin] (1x1) -> Wp [ceil(Cout/48)][Kp/16][4][48][4];  W2 [Cout<=16][48][3][3] -> W2p [9][3][4][16][4]. */
int eml_dense_permute_w1_f32(const float* W, int Cout, int Cin, int Kp, float* Wp, eml_stream_t stream);
int eml_dense_permute_w2_f32(const float* W2, int Cout, float* W2p, eml_stream_t stream);

/* All re-layouts of a pass in one launch: descs is a DEVICE array of n descriptors (the caller uploads it once; it stays
 * valid while the weight and workspace pointers do).  kind 0 = eml_dense_permute_w1_f32 (Cout, Cin, Kp), 1 =
 * eml_dense_permute_w2_f32 (Cout), 2 = eml_dense_permute_w1_bwd_f32 (Cout, Cin, Kp, Ko), 3 = eml_dense_permute_w2_tp_f32;
 * same outputs, bit for bit. */
typedef struct eml_permute_desc {
  const float* src;
  float* dst;
  int kind, Cout, Cin, Kp, Ko, reserved;
} eml_permute_desc;
int eml_dense_permute_batch_f32(const eml_permute_desc* descs, int n, eml_stream_t stream);

/* out[p][o] = sum_k relu(scale_k*X[p][k] + shift_k) * W[o][k]:  BN1 -> ReLU -> conv1 of a dense layer
 * (DenseNet.py:30-37; Cout = 48) and, with pool != 0, a transition BN -> ReLU -> conv -> avgpool2
 * (DenseNet.py:14-21; the pool is applied to the operand: it commutes with the 1x1 conv).
 * P output pixels; partials [ceil(Cout/48)][grid][48][2].
 * relu_mask (may be NULL; dense layers only): the ReLU mask of the BN1 output as bits, for the backward --
 * 64-bit words [ceil(P/256)*16 pixel groups][Kp/16][4]: word (pg, j, t), bit r + 16*q <-> pixel 16*pg + r,
 * channel 16*j + 4*q + t (q < 4, r < 16); P * Kp / 8 bytes rounded up to whole 256-pixel tiles. */
int eml_dense_conv1x1_fwd_f32(const float* X, int ldx, long P, int Hin, int Win, int pool, int Kp,
                              const float* scale, const float* shift, const float* Wp, int Cout,
                              float* out, int ldo, double* partials, int grid,
                              unsigned long long* relu_mask, eml_stream_t stream);

/* X[p][c_out0 + o] = conv3x3(scale2*Z + shift2)[p][o], o < 12: BN2 -> conv2 of a dense layer
 * (DenseNet.py:38-43, no ReLU between them); Z is (B,H,W,48); partials [grid][16][2]. */
int eml_dense_conv3x3_fwd_f32(const float* Z, const float* scale2, const float* shift2,
                              const float* W2p, float* X, int ldx, int c_out0, int B, int H, int W,
                              double* partials, int grid, eml_stream_t stream);

/* The same layer (DenseNet.py:38-43) TAP-PACKED: the 9 taps x 12 output channels are the 108 rows of the MFMA product of one
 * INPUT pixel (7 row tiles, 84 MFMAs per 16 pixels instead of 108; no padded rows, no halo tile) and the 3x3 sum is a
 * shift-and-add of accumulator registers (csrc/dense_fwd_tp.hip).  Same operands and results (f32 round-off of the summation
 * order) as eml_dense_conv3x3_fwd_f32 except the weight layout W2t (7*3*64*4 floats, eml_dense_permute_w2_tp_f32 or kind 3
 * of eml_dense_permute_batch_f32).  A workgroup owns `band_rows` output rows of one image at full width; `grid` workgroups
 * walk the B * ceil(H / band_rows) bands; partials [grid][16][2] (a workgroup without a band writes zeros).
 * eml_dense_conv3x3_fwd_tp_supported: 0, or the number of wavefronts side by side (1, 2, 4) when W = 16 * {4, 5} * that. */
int eml_dense_permute_w2_tp_f32(const float* W2, float* W2t, eml_stream_t stream);
int eml_dense_conv3x3_fwd_tp_supported(int B, int H, int W);
int eml_dense_conv3x3_fwd_tp_f32(const float* Z, const float* scale2, const float* shift2,
                                 const float* W2t, float* X, int ldx, int c_out0, int B, int H, int W,
                                 int band_rows, double* partials, int grid, eml_stream_t stream);

/* Transition operand: A[p'][c] = 2x2 mean of relu(scale[c]*X + shift[c]), p' over (B, Hin/2, Win/2), c < Kp
 * (Kp % 4 == 0; padded channels have scale = shift = 0).  The transition's 1x1 conv and its weight gradient
 * run on A with pool = 0 and a unit BN (scale 1, shift 0) -- replaces torch's avg_pool2d after the conv in
 * DenseNet.py:14-21 (the pool commutes with the 1x1 conv).
 * relu_mask16 (may be NULL): [B*Hin/2*Win/2][Kp/4] 16-bit words for the backward, bit 4*sub + g <-> pixel sub of the 2x2
 * window (row-major), channel 4*q + g: relu(scale*X + shift) > 0. */
int eml_dense_pool_act_f32(const float* X, int ldx, int B, int Hin, int Win, int Kp, const float* scale,
                           const float* shift, float* A, int lda, unsigned short* relu_mask16,
                           eml_stream_t stream);

/* out (B, C, H/k, W/k) = avg_pool_k(relu(F)) for NHWC F: the head of DenseNet.forward
 * (DenseNet.py:136-137), flattened in the reference's (C,h,w) order for `fc`. */
int eml_dense_head_pool_fwd_f32(const float* F, int ldf, int C, int B, int H, int W, int k,
                                float* out, eml_stream_t stream);

/* ---------------------------------------------------------------- DenseNet-BC encoder, backward
 * (autograd of DenseNet.py:14-65,88-122 -- the reference has no explicit backward code).
 * G mirrors the block buffer X and accumulates d loss / d X.  BatchNorm backward is affine per
 * channel, dx = cA*dy + cB*x + cC, so it is folded into the operand loads of the consumers. */

/* dzn = conv2^T(g) -> DZ (B,H,W,48); partials [grid][48][2] = (sum dzn, sum dzn*zhat).
 * X == NULL: g = G[:, c0:c0+12].  X != NULL: the deferred BN1 affine is applied on the fly,
 * g = G[:, c0+i] + sB[cx+i]*X[:, cx+i] + sC[cx+i] (what eml_dense_grad_materialize_f32 would write), and g is also
 * stored compactly to GF (B*H*W, 12) for eml_dense_conv3x3_bwd_weight_f32(GF, 12, 0, ...).  cx = the layer's channel
 * offset in X / sB / sC; (G, ldg, c0) is either the block gradient (ld, cx) or the compact tensor (12, 0) written by
 * eml_dense_conv1x1_bwd_narrow_f32. */
int eml_dense_conv3x3_bwd_data_f32(const float* G, int ldg, int c0, const float* W2, const float* Z,
                                   const float* zmean, const float* zistd, float* DZ, int B, int H,
                                   int W, double* partials, int grid, const float* X, int ldx, int cx,
                                   const float* sB, const float* sC, float* GF, eml_stream_t stream);

/* The two launches around this comment as ONE pass over the tiles (round 4): the data gradient with the fused BN1 affine
 * (X, sB, sC, GF as in eml_dense_conv3x3_bwd_data_f32 with X != NULL) and the weight gradient of the same layer,
 * dW2 = sum_p g[p] (x) (scale2*Z + shift2)[p+tap], on the g tile the data gradient has just staged (autograd of
 * DenseNet.py:38-43: conv2's backward w.r.t. its input and its weight).  Round 6: the weight gradient is summed over the
 * pixel the BN2(z) operand is taken at, with the 9 taps x 12 output channels as the MFMA rows (21 accumulator tiles instead
 * of 27; z at the tile's own pixels only, csrc/dense_bwd.hip: conv3x3_bwd_fused_tp_kernel).
 * Needs even ldg, c0, ldx, cx and 16-byte aligned buffers (eml_dense_conv3x3_bwd_fused_supported; otherwise EML_EINVAL:
 * issue the two launches); offsets that are multiples of 4 get 16-byte staging loads, the others pairs of 8-byte ones.
 * partials / grid as the data gradient's; partialW: grid*21*256 floats of scratch (sized as the weight gradient's
 * 2*grid*27*256 it always fits); dzn, GF and the statistics are what the separate launches return, dW2 agrees with the
 * separate launch to f32 round-off of the summation order. */
int eml_dense_conv3x3_bwd_fused_supported(int ldg, int c0, int ldx, int cx);
int eml_dense_conv3x3_bwd_fused_f32(const float* G, int ldg, int c0, const float* W2, const float* Z,
                                    const float* zmean, const float* zistd, float* DZ, int B, int H, int W,
                                    double* partials, int grid, const float* X, int ldx, int cx,
                                    const float* sB, const float* sC, float* GF, const float* scale2,
                                    const float* shift2, float* partialW, float* dW2, eml_stream_t stream);

/* dW2 (12,48,3,3) = sum_p G[p, c0:c0+12] (x) (scale2*Z + shift2)[p+tap]; partial: 2*grid*27*256 floats (two pixel halves per block). */
int eml_dense_conv3x3_bwd_weight_f32(const float* G, int ldg, int c0, const float* Z,
                                     const float* scale2, const float* shift2, int B, int H, int W,
                                     float* partial, float* dW2, int grid, eml_stream_t stream);

/* Finish a BatchNorm backward from R rows of (S1 = sum dy, S2 = sum dy*xhat) partials: dgamma, dbeta
 * and the affine dx = cA*dy + cB*x + cC (cA = gamma*istd, cB = -gamma*istd^2*S2/n,
 * cC = -gamma*istd*S1/n + gamma*istd^2*S2/n*mean), zero-padded to Cpad.  cA/cB/cC may be NULL.
 * sB/sC (may be NULL): running per-channel sums of (cB, cC) -- the deferred x-affine of a dense
 * block's gradient (s_accumulate = 0 overwrites, 1 adds).  Only channels [c_lo, c_hi) are processed.
 * W != NULL (BN1 of a dense layer, followed by ReLU and the 1x1 conv W [w_rows][C]): S2 is not read from the
 * partials but derived from the conv's finished weight gradient dW -- with a = relu(bn(x)) and dy = mask * (W^T dz),
 * sum_p dy*bn(x) = sum_o W[o][c]*dW[o][c], and bn(x) = gamma*xhat + beta, so
 * S2 = (sum_o W[o][c]*dW[o][c] - beta[c]*S1) / gamma[c]  (|gamma[c]| < 1e-12: S2 = 0).  The data-gradient pass then
 * needs no x at all (eml_dense_conv1x1_bwd_data_multi_f32 with relu_masks).
 * cond (C ints, may be NULL; W != NULL only): cond[c] = 1 where |gamma[c]| < 1e-3 * |beta[c]| or gamma[c] == 0 -- the
 * quotient then amplifies the f32 summation noise of the shared beta * sum dy term to more than ~0.1 % of dgamma -- else
 * 0; *any_cond (may be NULL) is set to 1 if any channel was flagged.  eml_dense_bn_dgamma_direct_f32 recomputes the
 * flagged channels. */
int eml_dense_bn_bwd_finalize_f32(const double* partials, int R, int pstride, double count,
                                  const float* gamma, const float* mean, const float* istd, int C,
                                  int Cpad, int training, float* dgamma, float* dbeta, float* cA,
                                  float* cB, float* cC, float* sB, float* sC, int s_accumulate,
                                  int c_lo, int c_hi, const float* beta, const float* W, const float* dW,
                                  int w_rows, int* cond, int* any_cond, eml_stream_t stream);

/* dgamma of the channels flagged in cond, from the definition: dgamma[c] = sum_p dy[p][c] * xhat[p][c] accumulated per
 * element in f64, dy = relu-mask(scale1*X + shift1) * sum_o W[o][c] dz[p][o], dz = cA*DY + cB*Zr + cC (Zr == NULL: DY
 * already holds dz), pool != 0: dz lives on the 2x2-pooled grid and is spread over the window (/4) -- the BatchNorm
 * backward of nn.BatchNorm2d before a ReLU + 1x1 conv (DenseNet.py:29-37, :17-20) without the weight-gradient identity.
 * Unflagged channels are not touched; with no flag at all both kernels return immediately.  P = input pixels,
 * W (Cout, Cin), Cin <= 384, Cout <= 192, scratch = grid*Cin doubles. */
int eml_dense_bn_dgamma_direct_f32(const float* X, int ldx, long P, int Hin, int Win, int pool,
                                   const float* DY, int ld_dy, const float* Zr, int ld_z, const float* cA,
                                   const float* cB, const float* cC, int Cout, const float* W, int Cin,
                                   const float* scale1, const float* shift1, const float* mean,
                                   const float* istd, const int* cond, double* scratch, float* dgamma,
                                   int grid, eml_stream_t stream);

/* dW (Cout,Cin) = sum_p dz[p] (x) relu(scale1*X[p] + shift1) with dz = cA*DY + cB*Zr + cC rebuilt
 * in the operand load (pool != 0: transition, 2x2 mean of the activation).
 * partial: grid*Kp*48 floats of scratch.
 * dz_out (may be NULL; dense layers only, Cout == 48, pool == 0): the rebuilt dz is also written to this (P,48)
 * buffer for the data-gradient passes; it may alias DY (in place).
 * N12 != NULL (dense layers): the narrow data pass of eml_dense_conv1x1_bwd_narrow_f32 rides on the dz tile this
 * kernel holds in LDS -- N12 (P,12) = G[:, k_lo:k_lo+12] + scale1*relu-mask*(dz W1[:, k_lo:k_lo+12]) with W1 (48,Cin)
 * the conv's weight (PyTorch layout), G (P, ldg) read only, k_lo even, and partials_n [grid][Kp][2] receiving S1 of
 * those 12 channels (S2 slot 0: see eml_dense_bn_bwd_finalize_f32).  ldg == 12: G is the COMPACT (P,12) tensor of
 * exactly those channels (eml_dense_conv1x1_bwd_data_multi_top_f32's top[0]), read from column 0. */
int eml_dense_conv1x1_bwd_weight_f32(const float* X, int ldx, long P, int Hin, int Win, int pool,
                                     int Kp, int Cin, const float* scale1, const float* shift1,
                                     const float* DY, int ld_dy, const float* Zr, int ld_z,
                                     const float* cA, const float* cB, const float* cC, int Cout,
                                     float* partial, float* dW, int grid, float* dz_out,
                                     const float* W1, int k_lo, const float* G, int ldg, float* N12,
                                     double* partials_n, eml_stream_t stream);

/* W (Cout,Cin) -> Wd [Kp/16][Ko/16][4][16][4], the B-fragment order of the data-gradient kernel. */
int eml_dense_permute_w1_bwd_f32(const float* W, int Cout, int Cin, int Kp, int Ko, float* Wd,
                                 eml_stream_t stream);

/* dam[p][k] = relu-mask(scale1*X+shift1) * sum_o dz[p][o] W[o][k]  (pool != 0: spread over the 2x2
 * input pixels, /4) and G[p][k] (+)= scale1[k]*dam[p][k] (accumulate = 0 overwrites);
 * partials [grid][Kp][2] = (sum dam, sum dam*xhat) for the BN1 backward.
 * relu_mask16 (transition form only; may be NULL): the words eml_dense_pool_act_f32 stored -- then X, mean and istd are
 * not read (may be NULL) and the partials carry S1 only (S2: eml_dense_bn_bwd_finalize_f32 with the conv's W / dW). */
int eml_dense_conv1x1_bwd_data_f32(const float* DY, int ld_dy, const float* Zr, int ld_z,
                                   const float* cA, const float* cB, const float* cC, int Ko,
                                   const float* Wd, const float* X, int ldx, const float* scale1,
                                   const float* shift1, const float* mean, const float* istd, long P,
                                   int Hin, int Win, int pool, int Kp, float* G, int ldg,
                                   int accumulate, double* partials, int grid,
                                   const unsigned short* relu_mask16, eml_stream_t stream);

/* The same data gradient for 1 or 2 CONSECUTIVE dense layers in one pass over the channel range
 * [k_lo, k_hi): G[p][k] += sum_j scale1_j[k]*dam_j[p][k] -- X is read once and G read-modify-written once
 * for both layers (the backward of a dense block is HBM-bound on exactly that traffic).  Per-layer
 * arrays of length n_layers (1 or 2); DZ/Zr are (P,48); partials_j is [grid][Kp_j][2].
 * Zr == NULL (then cA/cB/cC are ignored): DZ_j already holds dz_j (dz_out of eml_dense_conv1x1_bwd_weight_f32).
 * relu_masks != NULL (per layer: the relu_mask the forward wrote): the ReLU mask comes from those bits and X / mean /
 * istd are NOT read (they may be NULL) -- a third of the pass's bytes; partials then carry S1 only (S2 slot = 0; BN1's
 * S2 comes from the weight gradient, see eml_dense_bn_bwd_finalize_f32). */
int eml_dense_conv1x1_bwd_data_multi_f32(int n_layers, const float* const* DZ, const float* const* Zr,
                                         const float* const* cA, const float* const* cB,
                                         const float* const* cC, const float* const* Wd,
                                         const float* const* scale1, const float* const* shift1,
                                         double* const* partials, const int* Kp, const float* X, int ldx,
                                         const float* mean, const float* istd, long P, int k_lo,
                                         int k_hi, float* G, int ldg, int grid,
                                         const unsigned long long* const* relu_masks, eml_stream_t stream);

/* The two-layer masked pass above (n_layers = 2, relu_masks, materialised dz, channel range [0, k_hi)) with its top 24
 * channels leaving as two COMPACT (P,12) tensors instead of going back into the wide rows of G (round 6; same reference
 * lines: DenseNet.py:50-55 under autograd): top[0] = the updated gradient of channels [k_hi - 24, k_hi - 12), top[1] =
 * [k_hi - 12, k_hi) -- the output channels of the NEXT pair of layers down the block, which are read by that pair's
 * backward (eml_dense_conv3x3_bwd_*_f32 with (G, ldg, c0) = (top[1], 12, 0); eml_dense_conv1x1_bwd_weight_f32's narrow
 * operand with (G, ldg) = (top[0], 12)) and by nothing else.  A 48-byte slice of an 896-byte row costs its reader one or
 * two whole 128-byte lines per pixel; the compact tensors are read line for line.  G's columns [k_hi - 24, k_hi) keep
 * their OLD values.  k_hi >= 24 and a multiple of 4; top: (2, P, 12) floats. */
int eml_dense_conv1x1_bwd_data_multi_top_f32(const float* const* DZ, const float* const* Wd,
                                             const float* const* scale1, const float* const* shift1,
                                             double* const* partials, const int* Kp, long P, int k_hi, float* G,
                                             int ldg, int grid, const unsigned long long* const* relu_masks,
                                             float* top, eml_stream_t stream);

/* Narrow data pass (autograd of DenseNet.py:50-55 restricted to 12 channels): the data gradient of a dense layer
 * over the 12 output channels [k_lo, k_lo+12) of the layer below it, added to what G holds there and written as the
 * compact tensor N12 (P,12) = G[p][k] + scale1[k] * relu-mask * sum_o DZ[p][o] * W1[o][k]  (DZ = materialised dz
 * (P,48), W1 = conv1.weight (48,Cin)), which eml_dense_conv3x3_bwd_data_f32 then takes as its (G, 12, 0);
 * partials[grid][Kp][2] receives (sum dam, sum dam*xhat) at channels k_lo..k_lo+11 for the BN1 backward.
 * (The stand-alone form of the pass; the training engine uses the one riding on eml_dense_conv1x1_bwd_weight_f32.) */
int eml_dense_conv1x1_bwd_narrow_f32(const float* DZ, const float* W1, int Cin, int k_lo, const float* X,
                                     int ldx, const float* scale1, const float* shift1,
                                     const float* mean, const float* istd, long P, const float* G, int ldg,
                                     float* N12, double* partials, int Kp, int grid, eml_stream_t stream);

/* G[p][c] += sB[c]*X[p][c] + sC[c] for c in [c0, c0+n): applies the deferred BN1-backward affine once
 * the gradient of those channels is complete. */
int eml_dense_grad_materialize_f32(float* G, int ldg, const float* X, int ldx, const float* sB,
                                   const float* sC, int c0, int n, long P, eml_stream_t stream);

/* Stats of an elementwise BN(+ReLU) backward (last_norm; norm0+relu0): partials [grid][C][2]. */
int eml_dense_bn_bwd_stats_f32(const float* DY, int ld_dy, const float* raw, int ld_raw,
                               const float* out, int ld_out, int relu, int C, long P,
                               const float* mean, const float* istd, double* partials, int grid,
                               eml_stream_t stream);

/* dW0 (C0,3,3,3) with norm0+relu0's backward folded into the operand; partial: grid*4*1024 floats. */
int eml_dense_conv0_bwd_weight_f32(const float* x, const float* G, int ldg, const float* X1, int ldx,
                                   const float* Y0, int C0, const float* cA, const float* cB,
                                   const float* cC, int B, int H, int W, float* partial, float* dW0,
                                   int grid, eml_stream_t stream);

/* Block 1's input channels without the eml_dense_grad_materialize_f32 pass and without the block buffer (round 6; same
 * reference lines: DenseNet.py:88-92 under autograd): G is taken as the data-gradient passes left it -- the deferred BN1 affine
 * (sB, sC) still outstanding -- and Y0 is the raw conv0 output.  x = relu(scale0*Y0 + shift0) is what eml_dense_bn_apply_f32
 * wrote into the block buffer (the same expression, so the same bits), g = G + (sB*x + sC) what grad_materialize would have
 * written: the statistics of norm0 + relu0's backward (partials [grid][C][2], as eml_dense_bn_bwd_stats_f32 with relu = 1) and
 * conv0's weight gradient (as eml_dense_conv0_bwd_weight_f32) from those.  C a multiple of 4, <= 32; 16-byte aligned vectors. */
int eml_dense_norm0_bwd_stats_f32(const float* G, int ldg, const float* Y0, int ld_y0, const float* scale0,
                                  const float* shift0, const float* sB, const float* sC, int C, long P,
                                  const float* mean, const float* istd, double* partials, int grid, eml_stream_t stream);
int eml_dense_conv0_bwd_weight_fused_f32(const float* x, const float* G, int ldg, const float* Y0, int C0,
                                         const float* scale0, const float* shift0, const float* sB, const float* sC,
                                         const float* cA, const float* cB, const float* cC, int B, int H, int W,
                                         float* partial, float* dW0, int grid, eml_stream_t stream);

/* dF = relu-mask(F) * unpool_k(gpooled) / k^2 : backward of DenseNet.py:136-137. */
int eml_dense_head_pool_bwd_f32(const float* gpooled, const float* F, int ldf, int C, int B, int H,
                                int W, int k, float* dF, int ldd, eml_stream_t stream);

/* ---------------------------------------------------------------- GenProjector: SphereConv2D
 * models/networks/spherenet/sphere_cnn.py:111-124: y = conv2d(grid_sample(x, grid), weight, bias, stride=3) with a
 * fixed (1, 3Ho, 3Wo, 2) tangent-plane grid (:31-84) -- restated as A9 = im2col_sphere(x), Y = A9 * W2^T (library
 * GEMM on the caller's side), dx = col2im_sphere(dY * W2).  Pixel-major (channels-last) f32 throughout:
 * X (B, H*W, C), A9 (B*Ho*Wo*9, C) with row (b*Po + p)*9 + tap, tap = 3a + b as in the grid. */

/* The 4 bilinear corners of every (output pixel, tap): idx (Po*9, 4) = input pixel y*W + x or -1 (zero
 * padding), wgt (Po*9, 4), computed with torch.nn.functional.grid_sample's arithmetic (bilinear,
 * align_corners = False -- what sphere_cnn.py:122 runs with on torch >= 1.3).  grid: (3Ho, 3Wo, 2) as (x, y) in [-1, 1]. */
int eml_sphere_tap_table_f32(const float* grid, int H, int W, int Ho, int Wo, int* idx, float* wgt,
                             eml_stream_t stream);

/* A9[(b*Po + p)*9 + tap][c] = sum_k wgt[p,tap,k] * X[b][idx[p,tap,k]][c]   (replaces F.grid_sample, :122). */
int eml_sphere_im2col_f32(const float* X, const int* idx, const float* wgt, float* A9, int B, int HW,
                          int Po, int C, eml_stream_t stream);

/* Transpose of the above as a deterministic gather (replaces grid_sampler_2d_backward's atomicAdd scatter):
 * dX[b][q][c] = sum_{k in [ptr[q], ptr[q+1])} w[k] * dA9[(b*Po*9 + src[k])][c], the CSR transpose of the tap
 * table (src = p*9 + tap), built once per geometry by the caller. */
int eml_sphere_col2im_f32(const float* dA9, const int* ptr, const int* src, const float* w, float* dX,
                          int B, int HW, int Po, int C, eml_stream_t stream);

/* Fused SphereConv2D (sphere_cnn.py:111-124): the taps are gathered straight into the LDS operand tile of an f32-MFMA
 * implicit GEMM, so the (B*Po, 9C) operand of eml_sphere_im2col_f32 is never written.  X (B, HW, C) pixel-major,
 * idx / wgt = eml_sphere_tap_table_f32's table, W2 (O, 9C) with columns ordered (tap, c) (= weight.permute(0,2,3,1)),
 * bias (O) or NULL, Y (B*Po, O).  Forward: C % 32 == 0, O % 64 == 0.
 * Weight gradient dW2 (O, 9C) = sum_m dY[m] (x) Ag[m]: C % 64 == 0, O >= 64, O % 16 == 0; split_k workgroups share the
 * pixel axis, partial = eml_sphere_conv_wgrad_partial_floats(C, O, split_k) floats of scratch (deterministic sum).
 * PRECONDITION of the fused / narrow / small gather kernels (not of eml_sphere_im2col_f32, which selects the value): X is
 * finite.  A corner that falls off the map (idx = -1) is loaded from a clamped address and multiplied by weight 0, so an
 * Inf / NaN at pixel 0 of a sample would surface where grid_sample's zero padding gives a finite value. */
/* ke = table entries per (pixel, tap): 4 = the bilinear corners of the tap table; 1 = a single (index, weight) pair
 * (idx / wgt then hold Po*9 entries) -- an ordinary zero-padded 3x3 convolution written as a gather, used by the VGG19
 * feature stack of the perceptual loss (architecture.py:92-125): a quarter of the operand loads, no bilinear combine. */
int eml_sphere_conv_fwd_fused_f32(const float* X, const int* idx, const float* wgt, const float* W2,
                                  const float* bias, float* Y, int B, int HW, int Po, int C, int O, int ke,
                                  eml_stream_t stream);
/* The same product with its consumer's epilogue folded in: Y = leaky_relu(conv + bias + residual, act_slope).  residual
 * (B*Po, O) or NULL -- the `x_s + dx` of SPADEResnetBlock (architecture.py:60); Y may alias it.  act_slope in [0, 1]: 1 = none,
 * 0 = ReLU (VGG19's nn.ReLU after every convolution, architecture.py:92-125), 0.2 = the LeakyReLU before the generator's last
 * convolution (generator.py:84).  eml_sphere_conv_fwd_fused_f32 is this entry with (NULL, 1). */
/* table_flags: EML_TAP_ROWSHARE -- the caller vouches that the table has the row structure of a stride-1 sphere grid
 * (sphere_cnn.py:31-58: a tap samples source column c + const(row, tap) of two adjacent source rows): for every destination
 * pixel p with p % 4 != 3 and every tap, entries 1 of p and 0 of p + 1 (north-east of p / north-west of its right neighbour)
 * are the SAME source pixel unless one of them is -1 (the zero-padded wrap-around column; its weight is 0), likewise entries
 * 3 and 2; every -1 entry has weight 0; Po % 4 == 0.  The kernel then fetches 2 x 5 lines per 4 pixels instead of 4 x 4 (same
 * values, same summation order: identical results).  Honoured for ke == 4 only. */
#define EML_TAP_ROWSHARE 1
int eml_sphere_conv_fwd_fused_ex_f32(const float* X, const int* idx, const float* wgt, const float* W2,
                                     const float* bias, float* Y, int B, int HW, int Po, int C, int O, int ke,
                                     const float* residual, float act_slope, int table_flags, eml_stream_t stream);
/* The 3-channel input layers -- SPADE's mlp_shared 3 -> 128 + ReLU (normalization.py:92-96) and VGG19's conv1_1
 * 3 -> 64 + ReLU -- are bound by the write of their output: one pass each way.  (C, O) in {(3, 64), (3, 128)}
 * (eml_sphere_conv_small_supported); idx / wgt = the 4-entry tap table; X (B, HW, C), W2 (O, 9C),
 * Y (B*Po, O) = leaky_relu(conv + bias, act_slope).
 * Weight gradient: dW2 (O, 9C) = sum_m g'[m] (x) A[m] and db (O, or NULL) = sum_m g'[m], g' = dY * (Yact > 0 ? 1 : act_slope)
 * (Yact = the forward's output; NULL with act_slope = 1); partial = eml_sphere_conv_small_wgrad_partial_floats floats of
 * scratch (per-workgroup partial sums, reduced in a fixed order: deterministic). */
int eml_sphere_conv_small_supported(int C, int O);
int eml_sphere_conv_small_fwd_f32(const float* X, const int* idx, const float* wgt, const float* W2, const float* bias,
                                  float* Y, int B, int HW, int Po, int C, int O, float act_slope, eml_stream_t stream);
size_t eml_sphere_conv_small_wgrad_partial_floats(int B, int Po, int C, int O);
int eml_sphere_conv_small_wgrad_f32(const float* X, const int* idx, const float* wgt, const float* dY, const float* Yact,
                                    float act_slope, float* partial, float* dW2, float* db, int B, int HW, int Po, int C,
                                    int O, eml_stream_t stream);
/* Input gradient of those layers, first half (where the 3-channel input carries a gradient: the guide map of the joint step,
 * VGG19's conv1_1 on the generated panorama): dA9 (M, 9C) = g' W2, g' as above, M = B*Po pixel rows -- one read of (dY, Yact);
 * eml_sphere_col2im_f32 then gathers dA9 into dX.  Replaces an activation-backward pass + an N = 27 library GEMM. */
int eml_sphere_conv_small_da9_f32(const float* dY, const float* Yact, float act_slope, const float* W2, float* dA9, long M,
                                  int C, int O, eml_stream_t stream);
/* The few-channel OUTPUT layers (generator.py:60, 84-86: conv_img 64 -> 3 at full resolution; discriminator.py:70-74: the
 * final 512 -> 1 convolutions): sphere_cnn.py:111-124 with O <= 4 as three one-pass kernels -- no 9x im2col operand, no N = 3
 * library GEMM, no dA9 + col2im in the backward.  C % 64 == 0, C <= 512, 1 <= O <= 4 (eml_sphere_conv_narrow_supported).
 * X (B, HW, C) pixel-major, idx / wgt (Po*9*4) the bilinear tap table, W2 (O, 9C) columns (tap, c), Y / dY (B*Po, O).
 *   fwd   Y = A9(X) W2^T + bias (bias may be NULL)
 *   dgrad dX (B*HW, C) through the TRANSPOSED tap table tidx / twgt (HW*9*ke, ke in 1..8; -1 / 0 = empty slot): a gather
 *   wgrad dW2 (O, 9C); partial = eml_sphere_conv_narrow_wgrad_partial_floats(B, Po, C, O) floats of scratch; fixed-order sums */
int eml_sphere_conv_narrow_supported(int C, int O);
int eml_sphere_conv_narrow_fwd_f32(const float* X, const int* idx, const float* wgt, const float* W2, const float* bias, float* Y,
                                   int B, int HW, int Po, int C, int O, eml_stream_t stream);
int eml_sphere_conv_narrow_dgrad_f32(const float* dY, const int* tidx, const float* twgt, int ke, const float* W2, float* dX, int B,
                                     int HW, int Po, int C, int O, eml_stream_t stream);
size_t eml_sphere_conv_narrow_wgrad_partial_floats(int B, int Po, int C, int O);
int eml_sphere_conv_narrow_wgrad_f32(const float* X, const int* idx, const float* wgt, const float* dY, float* partial, float* dW2,
                                     int B, int HW, int Po, int C, int O, eml_stream_t stream);

/* The same layers, project-then-gather (round 6, csrc/sphere_conv_narrow2.hip; same reference lines: sphere_cnn.py:111-124).
 * With O <= 4 the 36 gathers per pixel at full channel width (9.2 KB of L1 traffic per pixel of conv_img) are what bounds the
 * kernels above; the convolution is linear and the bilinear weights do not depend on the channel, so
 *   forward:  P[q][tap][o] = sum_c W2[o][tap*C + c] X[q][c] for every SOURCE pixel (f32-MFMA GEMM), then
 *             Y[m][o] = bias[o] + sum_{tap,e} wgt[p,tap,e] P[idx[p,tap,e]][tap][o]   (36 16-byte gathers per pixel);
 *   wgrad / dgrad:  V[q][tap][o] = sum_s twgt[q,tap,s] dY[tidx[q,tap,s]][o] over the TRANSPOSED tap table (tidx / twgt / ke as for
 *             eml_sphere_conv_narrow_dgrad_f32; rowmax (HW bytes, may be NULL): the largest entry count of a tap of source
 *             pixel q -- pixels with <= 4 skip slots 4..7 of a ke = 8 table), then dW2[o][tap*C + c] = sum_q V[q][tap][o] X[q][c] (split-K f32-MFMA GEMM,
 *             fixed-order second pass).
 * scratch: eml_sphere_conv_narrow_scratch_floats(B, HW) floats (P or V: (B*HW, 36)); partial:
 * eml_sphere_conv_narrow_wgrad2_partial_floats(B, HW, C) floats.  Same supported widths (eml_sphere_conv_narrow_supported).
 * Results agree with the first generation to f32 round-off (the order of the sums differs). */
size_t eml_sphere_conv_narrow_scratch_floats(int B, int HW);
int eml_sphere_conv_narrow_fwd2_f32(const float* X, const int* idx, const float* wgt, const float* W2, const float* bias,
                                    float* Y, float* scratch, int B, int HW, int Po, int C, int O, eml_stream_t stream);
size_t eml_sphere_conv_narrow_wgrad2_partial_floats(int B, int HW, int C);
int eml_sphere_conv_narrow_wgrad2_f32(const float* X, const int* tidx, const float* twgt, int ke,
                                      const unsigned char* rowmax, const float* dY, float* scratch, float* partial, float* dW2, int B, int HW, int Po, int C, int O,
                                      eml_stream_t stream);
/* dX (B*HW, C) = V W2 in the (tap, o) x c arrangement (f32 MFMA, K = 36): the same V.  scratch_has_v != 0: `scratch` still
 * holds the V that eml_sphere_conv_narrow_wgrad2_f32 built from this dY; else it is built first. */
int eml_sphere_conv_narrow_dgrad2_f32(const float* dY, const int* tidx, const float* twgt, int ke,
                                      const unsigned char* rowmax, const float* W2, float* dX, float* scratch,
                                      int scratch_has_v, int B, int HW, int Po, int C, int O, eml_stream_t stream);
/* Input gradient with the same kernel: dX (B*HW, C) = sum_{tap,o} Dg[q][tap][o] * W2t[c][tap*O + o], Dg = dY gathered through
 * the TRANSPOSED tap table tidx / twgt (HW*9*ke ints / floats: for input pixel q and tap t, the output pixels whose tap t
 * samples q, -1 / weight 0 = empty slot; ke = 4 or 8 slots).  rowmax (HW bytes, required for ke = 8) = per input pixel the
 * largest slot count over its taps, so that only tiles with pole rows fetch slots 4..7; ke = 1: one entry per (pixel, tap), as
 * in the forward.  W2t (C, 9*O) = weight.permute(1,2,3,0).
 * O % 32 == 0, C % 64 == 0.  Deterministic (a gather, no atomics); neither dA9 nor its col2im pass exist. */
int eml_sphere_conv_dgrad_fused_f32(const float* dY, const int* tidx, const float* twgt, const unsigned char* rowmax,
                                    int ke, const float* W2t, float* dX, int B, int HW, int Po, int C, int O,
                                    int table_flags, eml_stream_t stream);
/* ROUND 6: the same product for the LOW-RESOLUTION, WIDE layers (1024 -> 1024 @8x16, 128 -> 2048 / 1024 -> 512 @16x32,
 * 128 -> 1024 / 512 -> 256 @32x64 of generator.py:33-39; sphere_cnn.py:111-124) -- the layers that ran as eml_sphere_im2col_f32 +
 * a library GEMM (+ eml_sphere_col2im_f32 for the input gradient) until round 5.  csrc/gather_gemm3.h: the source rows the 9
 * taps of a 128-pixel tile touch (its FOOTPRINT, a contiguous pixel range of the sample) are staged in LDS once per 32-channel
 * chunk and serve all 9 taps; a lane builds its MFMA fragments straight from them (4 corner reads + the bilinear combine in
 * grid_sample's order: bit-identical to the kernels above); no gathered global load, no operand tile, no 9x tensor.
 *   X (B*HW, C) pixel-major; wgt (Po*9*ke): the weights of a forward tap table (ke = 4) or a transposed one (ke = 4 / 8 with
 *       rowmax): the input gradient is this entry on (dY, tidx, twgt, W2t) with the roles of HW / Po and C / O swapped, as above.
 *   fp (2 ints per 128-pixel tile of a sample: first source pixel, count) when Po % 128 == 0, NULL when Po divides 128 (a
 *       tile then spans 128 / Po whole samples); fp_max = the largest count (resp. (128 / Po) * HW): selects the variant.
 *   lidx (Po*9*ke): the table's indices made FOOTPRINT-LOCAL by eml_sphere_conv_lowres_table_i32 (once per geometry): row offset
 *       into the tile's staged footprint, 0 for an empty slot (weight 0) -- the tap loop then adds, it does not select or multiply.
 *   split >= 1 divides the C / 32 channel chunks over that many workgroups per tile (layers with too few tiles for 256 CUs):
 *       raw sums go to partial (eml_sphere_conv_lowres_partial_floats) and a second launch adds them in a fixed order and
 *       applies the epilogue  Y = leaky_relu(conv + bias + residual, act_slope).  Deterministic, no atomics.
 * eml_sphere_conv_lowres_variant: 0 = shape not supported (C % 32, O % 128, Po % 128 == 0 or Po | 128, footprint <= 256
 * pixels -- or <= 448 with O % 256 == 0), else the tile variant that will run. */
int eml_sphere_conv_lowres_variant(int C, int O, int Po, int fp_max);
int eml_sphere_conv_lowres_table_i32(const int* idx, const int* fp, int* lidx, int Po, int ke, eml_stream_t stream);
size_t eml_sphere_conv_lowres_partial_floats(long M, int O, int split);
int eml_sphere_conv_lowres_f32(const float* X, const int* lidx, const float* wgt, const unsigned char* rowmax, int ke,
                               const int* fp, int fp_max, const float* W2, const float* bias, float* Y, float* partial,
                               int split, int B, int HW, int Po, int C, int O, const float* residual, float act_slope,
                               eml_stream_t stream);
size_t eml_sphere_conv_wgrad_partial_floats(int C, int O, int split_k);
int eml_sphere_conv_wgrad_fused_f32(const float* X, const int* idx, const float* wgt, const float* dY,
                                    float* partial, float* dW2, int B, int HW, int Po, int C, int O,
                                    int split_k, eml_stream_t stream);

/* SPADE modulation (normalization.py:113-115 + the LeakyReLU of architecture.py:56-57) on pixel-major rows:
 * y[r][c] = leaky_relu(xn[r][c] * (1 + gb[r][c]) + gb[r][C + c], slope), slope = 1 for the plain modulation.
 * gb (rows, >= 2C) holds gamma | beta side by side, as the fused gamma/beta SphereConv writes them.  C % 4 == 0,
 * every ld a multiple of 4.  Backward: d = gy * leaky_relu'(t); dxn = d * (1 + gamma); dgb = (d * xn | d). */
int eml_spade_modulate_fwd_f32(const float* xn, int ld_x, const float* gb, int ld_gb, float* y, int ld_y,
                               long rows, int C, float slope, eml_stream_t stream);
int eml_spade_modulate_bwd_f32(const float* gy, int ld_gy, const float* xn, int ld_x, const float* gb,
                               int ld_gb, float* dxn, int ld_dx, float* dgb, int ld_dgb, long rows, int C,
                               float slope, eml_stream_t stream);

/* SPADE's parameter-free BatchNorm (normalization.py:86-104; across replicas: sync_batchnorm/batchnorm.py:105-126)
 * folded into the modulation.  Rows are pixel-major (channels-last), C % 4 == 0.
 *   eml_bn_stats_f32     partials[grid][C][2] (f64) = per-block (sum x, sum x^2), accumulated per element in f64
 *   eml_bn_fold_f64      sums[k] = sum_r partials[r][k], k < n (n = 2C); the caller appends the row count at sums[2C]
 *                        and may all-reduce sums[0..2C] across ranks before finalising (= SynchronizedBatchNorm)
 *   eml_bn_finalize_f32  mean, istd = rsqrt(var + eps) from sums (count at sums[2C]); running_mean / running_var
 *                        (NULL to skip) updated like nn.BatchNorm2d (momentum, unbiased variance)
 *   eml_spade_norm_modulate_fwd_f32   y = leaky_relu(((x-mean)*istd) * (1 + gamma) + beta, slope)
 *   eml_spade_norm_modulate_bwd_cols_f32 (below)   dxn, dgb as eml_spade_modulate_bwd_f32 with xn = (x-mean)*istd rebuilt
 *                        inline, + per-block partials of (sum dxn, sum dxn*xhat) and of dgb's column sums
 *   eml_bn_bwd_apply_f32 dx = istd * (dxn - S1/n - xhat*S2/n) with (S1,S2) = sums[c][0..1], n = sums[2C];
 *                        sums == NULL: eval mode, dx = istd * dxn.  dx may alias dxn. */
int eml_bn_stats_f32(const float* x, int ld, long rows, int C, double* partials, int grid, eml_stream_t stream);
int eml_bn_fold_f64(const double* partials, int R, int n, double* sums, eml_stream_t stream);
int eml_bn_finalize_f32(const double* sums, int C, float eps, float momentum, float* mean, float* istd,
                        float* running_mean, float* running_var, eml_stream_t stream);
int eml_spade_norm_modulate_fwd_f32(const float* x, int ld_x, const float* gb, int ld_gb, float* y, int ld_y,
                                    long rows, int C, float slope, const float* mean, const float* istd,
                                    eml_stream_t stream);
int eml_bn_bwd_apply_f32(const float* dxn, int ld_d, const float* x, int ld_x, long rows, int C,
                         const float* mean, const float* istd, const double* sums, float* dx, int ld_o,
                         eml_stream_t stream);

/* The same three SPADE passes with the generator's nearest-neighbour x2 upsample of the block input (generator.py:70-82,
 * `x = self.up(x)` before up_0..up_3) folded in: x_lo (B, H/2, W/2, C) is the map BEFORE the upsample, everything else lives
 * on the (B, H, W) grid; pixel (b, h, w) reads x_lo at (b, h/2, w/2).  All tensors dense pixel-major (row stride C, 2C for
 * gb / dgb).  mean / istd are the statistics of x_lo (those of the upsampled map: every value appears four times); the
 * backward (eml_spade_norm_modulate_bwd_cols_f32 with up2 = 1) reduces over the B*H*W upsampled pixels, and
 * eml_bn_bwd_apply_up2_f32 returns the gradient of x_lo:
 *   dx_lo[q] = istd * (sum over q's 4 children of dxn - 4 S1/n - 4 xhat[q] S2/n). */
int eml_spade_norm_modulate_up2_fwd_f32(const float* x_lo, const float* gb, float* y, int B, int H, int W, int C, float slope,
                                        const float* mean, const float* istd, eml_stream_t stream);
int eml_bn_bwd_apply_up2_f32(const float* dxn, const float* x_lo, int B, int H, int W, int C, const float* mean,
                             const float* istd, const double* sums, float* dx_lo, eml_stream_t stream);

/* The modulation's backward (plain or with the folded upsample), which also leaves the COLUMN SUMS of dgb -- the bias gradient of the gamma | beta
 * SphereConv2D that produced gb (normalization.py:97-98) -- in its partials, so that the largest gradient tensor of the step is
 * not re-read just to be added up.  Partial row = (C, 2) (sum dxn, sum dxn*xhat) | 1 unused slot | (C, 2) (sum dgamma, sum
 * dbeta) = 4C + 1 doubles per block: eml_bn_fold_f64(partials, grid, 4C + 1, out) gives `sums` (2C + 1, the caller sets the
 * count at [2C]) followed by the bias-gradient table.  up2 != 0: x = the map before the x2 upsample, (H, W) the upsampled size.
 * Dense tensors: row stride C (gy, x, dxn), 2C (gb, dgb). */
int eml_spade_norm_modulate_bwd_cols_f32(const float* gy, const float* x, const float* gb, float* dxn, float* dgb, int B, int H,
                                         int W, int C, int up2, float slope, const float* mean, const float* istd,
                                         double* partials, int grid, eml_stream_t stream);

/* SPADE in ONE launch (normalization.py:101-115: `normalized * (1 + gamma) + beta` with gamma = mlp_gamma(actv), beta =
 * mlp_beta(actv), both SphereConv2D 128 -> norm_nc; architecture.py:56-57 for the LeakyReLU that follows): the gather-GEMM of
 * the concatenated heads over `actv` (B, H, W, Cin) with the modulation of x as its epilogue,
 *     Y[m][c] = leaky_relu(((x[m'][c] - mean[c]) * istd[c]) * (1 + gamma[m][c]) + beta[m][c], act_slope)       (B*H*W, Cn)
 * so the (B*H*W, 2 Cn) gamma | beta tensor is never written or re-read.  idx / wgt: the stride-1 bilinear tap table of the
 * (H, W) grid (ke = 4).  W2r (2 Cn, 9 Cin) / bias_r (2 Cn, or NULL): the rows of cat(gamma head, beta head), columns (tap, c),
 * REORDERED so that a wave of the kernel holds gamma and beta of the same channels: position p holds source row
 * c + half * Cn with c = 64 (p / 128) + 32 ((p % 128) / 64) + p % 32 and half = (p % 64) / 32.  up2 != 0: x (B, H/2, W/2, Cn)
 * is the map before the generator's nearest x2 upsample (m' = the parent pixel), else x (B*H*W, Cn).  gamma_out (B*H*W, Cn) or
 * NULL: gamma, which the backward needs (eml_spade_norm_modulate_bwd_y_f32 takes it together with Y; beta is not needed: the
 * activation's mask is Y's sign).  eml_sphere_conv_spade_supported: Cin % 32 == 0, Cin >= 64, Cn % 64 == 0 and the 32-bit
 * offsets of the kernel reach (H*W*Cin < 2^29, 2 Cn * 9 Cin < 2^30). */
int eml_sphere_conv_spade_supported(int Cin, int Cn, long HW);
int eml_sphere_conv_spade_fwd_f32(const float* actv, const int* idx, const float* wgt, const float* W2r, const float* bias_r,
                                  const float* x, const float* mean, const float* istd, float* Y, float* gamma_out, int B, int H,
                                  int W, int Cin, int Cn, int up2, float act_slope, int table_flags, eml_stream_t stream);
/* eml_spade_norm_modulate_bwd_cols_f32 for that forward: gamma (B*H*W, C) and the forward's output y (B*H*W, C) in place of
 * the (gamma | beta) tensor; dgb (B*H*W, 2C) = (dgamma | dbeta), partials as above.  slope in [0, 1]. */
int eml_spade_norm_modulate_bwd_y_f32(const float* gy, const float* x, const float* gamma, const float* y, float* dxn, float* dgb,
                                      int B, int H, int W, int C, int up2, float slope, const float* mean, const float* istd,
                                      double* partials, int grid, eml_stream_t stream);

/* torch.nn.utils.spectral_norm of a 3x3 convolution weight (normalization.py:24-33, architecture.py:41-45), fused with
 * the (O, tap, c) re-layout of the gather-GEMM kernels.  W (O, C, 3, 3) = weight_orig; u (O), v (9C) = the module's
 * weight_u / weight_v buffers.  iterate != 0 (training): one power iteration in place, v = normalize(W^T u),
 * u = normalize(W v) with normalize(x) = x / max(|x|_2, eps); then sigma = u . (W v) and W2[o][tap*C + c] = W[o][c][tap] / sigma.
 * uv_used (O + 9C) receives the (u | v) the result was formed with -- the constants of the backward:
 *   dW[o][c][tap] = (dW2[o][tap*C + c] - <dW2, W2> u[o] v[c*9 + tap]) / sigma.
 * scratch = eml_spectral_norm_scratch_floats(O, C) floats; partial = 256 doubles.  4 launches forward, 2 backward, in place
 * of the dozen-plus of the stock hook and the separate re-layout copy. */
size_t eml_spectral_norm_scratch_floats(int O, int C);
int eml_spectral_norm_w2_f32(const float* W, float* u, float* v, int iterate, float eps, float* W2, float* sigma,
                             float* uv_used, float* scratch, int O, int C, eml_stream_t stream);
int eml_spectral_norm_w2_bwd_f32(const float* dW2, const float* W2, const float* u_used, const float* v_used,
                                 const float* sigma, double* partial, float* dW, int O, int C, eml_stream_t stream);
/* eml_spectral_norm_w2_f32 for n weights at once (host arrays of n device pointers / shapes; one `iterate` and `eps` for all):
 * 5 launches for every spectrally normalised convolution of a network (the generator: 23, the discriminators: 6) at the top
 * of its forward, where torch's per-module hook (architecture.py:41-45 wraps each convolution) puts them in front of each
 * layer.  Item i is computed from and into its own buffers exactly as by the single call: bit-identical results. */
int eml_spectral_norm_w2_batch_f32(int n, const float* const* W, float* const* u, float* const* v, int iterate, float eps,
                                   float* const* W2, float* const* sigma, float* const* uv_used, float* const* scratch,
                                   const int* O, const int* C, eml_stream_t stream);

/* SPADE's two heads (normalization.py:96-98: mlp_gamma, mlp_beta -- SphereConv2D(nhidden, Cn) each) as the ONE (2 Cn, 9 C)
 * operand of the gamma | beta product, columns (tap, c): W2[p][tap*C + c] = W_head(p)[c(p)][c][tap], b2[p] = b_head(p)[c(p)]
 * (b2 / bg / bb may be NULL).  reorder = 0: cat order (p < Cn: gamma row p, else beta row p - Cn) -- what
 * eml_sphere_conv_fwd_fused_f32 / the im2col GEMM take; reorder != 0: the row order of eml_sphere_conv_spade_fwd_f32
 * (Cn % 64 == 0).  One launch instead of cat(weights), cat(biases) and a re-layout copy per SPADE and pass. */
int eml_spade_heads_w2_f32(const float* Wg, const float* Wb, const float* bg, const float* bb, float* W2, float* b2, int Cn,
                           int C, int reorder, eml_stream_t stream);

/* nn.InstanceNorm2d(affine=False) + the LeakyReLU that follows it in the PatchGAN discriminator and the crop encoder
 * (normalization.py:44-45 'instance'; discriminator.py:84-98; generator.py:113-122): y = leaky_relu((x - mean) * istd, slope),
 * mean / biased variance per (sample, channel) over the HW pixels (f64 accumulation), istd = 1/sqrt(var + eps).
 * x, y, gy, dx: dense (B, HW, C) pixel-major tensors when channels_last != 0 (C % 4 == 0), dense (B, C, HW) otherwise.
 * stats (B, C, 2) f32 = (mean, istd), written by the forward and read by the backward:
 *   g' = gy * (xhat > 0 ? 1 : slope),  dx = istd * (g' - mean_HW(g') - xhat * mean_HW(g' * xhat)).
 * slope in [0, 1] (1 = no activation).  One launch each way; replaces ATen's copy + batch_norm_* + leaky_relu chains. */
int eml_instance_norm_act_fwd_f32(const float* x, float* y, float* stats, int B, int HW, int C, int channels_last,
                                  float eps, float slope, eml_stream_t stream);
int eml_instance_norm_act_bwd_f32(const float* gy, const float* x, const float* stats, float* dx, int B, int HW, int C,
                                  int channels_last, float slope, eml_stream_t stream);

/* ---------------------------------------------------------------- the generator's L1-type loss terms, one launch each way
 * models/pix2pix_model.py:99-117 (mask-weighted feature matching over the discriminators' feature maps, x50 off the lights)
 * and models/networks/loss.py:102-114 (VGGLoss: sum_i w_i L1(vgg_i(fake), vgg_i(real))).  A term is a pair (f, r) of
 * channels-last maps -- rows[i] pixel rows of C[i] floats each -- with an optional per-row weight w[i] (rows[i] floats, or NULL;
 * |w| is used) and a scale[i] (coefficient / numel):
 *   fwd  out[0] = sum_i scale[i] * sum_e |f_i[e] - r_i[e]| * |w_i[row(e)]|     partial: eml_l1_pairs_partial_doubles(npairs) f64
 *   bwd  g_i[e] = gout[0] * scale[i] * sign(f_i[e] - r_i[e]) * |w_i[row(e)]|;  gzero[i] (or NULL): a region of nzero[i] floats
 *        zero-filled by the same launch (the real half of a feature map whose gradient tensor spans fake | real).
 * The host arrays (npairs <= 16 entries each) are read during the call and travel in the kernel arguments: no device table, no
 * host sync.  Deterministic (per-workgroup f64 partials, fixed-order fold). */
size_t eml_l1_pairs_partial_doubles(int npairs);
int eml_l1_pairs_fwd_f32(int npairs, const float* const* f, const float* const* r, const float* const* w, const long* rows,
                         const int* C, const float* scale, double* partial, float* out, eml_stream_t stream);
int eml_l1_pairs_bwd_f32(int npairs, const float* const* f, const float* const* r, const float* const* w, const long* rows,
                         const int* C, const float* scale, const float* gout, float* const* g, float* const* gzero,
                         const long* nzero, eml_stream_t stream);

/* out[c] = sum_r x[r][c] for a tall row-major (rows, cols) matrix with cols <= 64: the bias gradient of the few-output-channel
 * layers (autograd of sphere_cnn.py:124's bias for conv_img / the discriminators' heads).  partial:
 * eml_colsum_partial_doubles(cols) f64 of scratch; deterministic. */
size_t eml_colsum_partial_doubles(int cols);
int eml_colsum_f32(const float* x, long rows, int cols, double* partial, float* out, eml_stream_t stream);

/* ---------------------------------------------------------------- ground-truth parametrisation (data preparation)
 * representation/distribution_representation.py:65-120 (`extract_mesh`), the inverse of the rasteriser.
 * idx (H*W) = nearest anchor of every pixel of the endpoint-inclusive (theta, phi) grid (:74-87); anchors (N,3) f64. */
int eml_gt_anchor_index_i32(const double* anchors, int N, int H, int W, int* idx, eml_stream_t stream);

/* compute() for a batch (:90-107).  hdr (B,H,W,3) f32 -> maxv (B) = max steradian-weighted luminance,
 * sums (B, N+1, 3) f64 = RGB energy of the lit (> 5 % of max) pixels of each anchor's cell, row N = ambient
 * (the unlit remainder), map (B,H,W) u8 lit mask or NULL.  csr_ptr (N+1) / csr_pix (H*W): pixels grouped by idx. */
int eml_gt_parametrise_f64(const float* hdr, const int* csr_ptr, const int* csr_pix, int B, int H, int W,
                           int N, double* maxv, double* sums, unsigned char* map, eml_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EMLIGHT_HIP_H */
