/*
 * corenet_hip.h -- C ABI of libcorenet_hip.so, the MI355X (gfx950) native
 * implementation of the CoReNet forward/backward hot path.
 *
 * Plain C: raw device pointers, explicit sizes/strides, a hipStream_t passed as
 * void*.  No torch types.  Every entry point returns 0 on success, a negative
 * CRN_E* code on bad arguments, or a positive hipError_t.
 *
 * Each entry point names the reference (google-research/corenet, paths relative
 * to src/corenet/) interface it replaces.  The Python host in corenet_amd/
 * mirrors the reference's module API on top of these.
 */
#ifndef CORENET_HIP_H_
#define CORENET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRN_OK 0
#define CRN_EINVAL (-1)   /* bad argument (shape / alignment / unsupported mode) */
#define CRN_ENOMEM (-2)   /* workspace too small */
#define CRN_ENOCONV (-3)  /* iterative kernel did not converge within max rounds */

typedef void* crnStream;  /* hipStream_t */

/* Logical NCDHW view of an fp32 tensor.  element(b,c,d,h,w) lives at
 *   base + b*sB + (chan_off ? chan_off[c] : c*sC) + d*sD + h*sH + w*sW
 * chan_off (device int32 table) expresses space-to-depth / depth-to-space
 * (pixel-shuffle) channel layouts without materialising them.              */
typedef struct {
  float* base;
  int32_t B, C, D, H, W;
  int64_t sB, sC;
  const int32_t* chan_off;
  int32_t sD, sH, sW;
} crnView;

/* Per-channel affine(+ReLU) applied to the INPUT of a conv while it is staged
 * into LDS: v = x; if(pre_relu) v=max(v,0); v = v*scale[c]+shift[c];
 * if(post_relu) v=max(v,0).  Zero padding is applied AFTER the transform
 * (reference: BatchRenorm/ReLU modules precede the padded conv,
 * reconstruction_decoder.py:56-60, resnet50.py:62-69).  scale==NULL => identity. */
typedef struct {
  const float* scale;
  const float* shift;
  int32_t pre_relu, post_relu;
} crnInTransform;

/* Structural zeros of the packed weights (transposed convolutions written as window correlations:
 * each of the 8 output/input parities only uses a sub-box of the window, 343 of 8*64 taps for k=7).
 * The logical OUTPUT channels [0, y->C) form n_groups equal contiguous groups, the logical INPUT
 * channels [0, x->C) c_groups groups; for group g every weight with a tap outside the half-open box
 * (d0,d1,h0,h1,w0,w1) is zero, and the kernels skip those taps.  0 groups = no information.  Host memory. */
typedef struct {
  int32_t n_groups, c_groups;          /* 0..8 */
  int8_t n_box[8][6];
  int8_t c_box[8][6];
} crnTapBoxes;

/* ---------------- convolutions (MFMA fp32 implicit GEMM) -------------------
 * One stride-1 window correlation covers Conv2d / Conv3d / ConvTranspose3d
 * forward and data-gradient of the reference (resnet50.py:62-69,95-107,124;
 * reconstruction_decoder.py:49-95; ray_traced_skip_connection.py:38) through
 * views + packed weights (corenet_amd/model/conv_geometry.py):
 *   y[b,n,o] = bias[b*bias_sB+n] + sum_{c,t} T(x)[b,c,o - pad_lo + t] * w[(c*T+t)*Npad + n]
 * w: packed [Cin][kd*kh*kw][Npad] fp32, Npad multiple of 16.
 * splits>1 splits the Cin reduction over blocks (atomic accumulate; y is
 * zeroed first by the call).                                                 */
int crn_conv_fwd(const crnView* x, const crnInTransform* tr, const float* w, int Npad,
                 const float* bias, int bias_sB, const crnView* y,
                 int kd, int kh, int kw, int pd, int ph, int pw,
                 int splits, int accumulate /* y += result instead of y = result */,
                 const crnTapBoxes* boxes /* may be NULL */, crnStream stream);

/* The same operation on the split-bf16 ("bf16x3") MFMA engine (csrc/conv_bf3.hip): every fp32 operand is
 * split into two bf16 terms and a product is three v_mfma_f32_16x16x32_bf16 with fp32 accumulation
 * (hi*hi + hi*lo + lo*hi; ~2^-16 relative per product instead of fp32's 2^-24).  A throughput mode for
 * the big decoder layers (reconstruction_decoder.py:72-95), selected per layer by the host; the fp32
 * engine above stays the parity default.  Returns CRN_EINVAL for shapes it does not cover
 * (output W not a multiple of 16, views other than unit-stride / stride-2 space-to-depth, Cin > 256).   */
int crn_conv_fwd_bf3(const crnView* x, const crnInTransform* tr, const float* w, int Npad,
                     const float* bias, int bias_sB, const crnView* y,
                     int kd, int kh, int kw, int pd, int ph, int pw,
                     int accumulate, const crnTapBoxes* boxes /* may be NULL */, crnStream stream);

/* crn_conv_fwd_bf3 with the weights pre-split and pre-arranged by crn_bf3_operands (slab order): the kernel copies
 * its weight slabs instead of gathering and splitting them at every staging step.  Same results bit for bit.     */
int crn_conv_fwd_bf3_slabs(const crnView* x, const crnInTransform* tr, const void* wslab, int Npad,
                           const float* bias, int bias_sB, const crnView* y,
                           int kd, int kh, int kw, int pd, int ph, int pw,
                           int accumulate, const crnTapBoxes* boxes /* may be NULL */, crnStream stream);

/* crn_conv_fwd_bf3_slabs as the DATA GRADIENT in front of a BatchRenorm's backward (reconstruction_decoder.py:56-60: a decoder
 * block is ReLU -> BatchRenorm -> conv, so the data gradient of the conv is the output gradient dy of the norm, and
 * batch_renorm.py:41-47's autograd needs sum(dy) and sum(dy * xn) over the tensor before it can form dx).  The launch that
 * writes dy = y also accumulates those two sums per channel from its accumulators, reading the norm's input x once, and
 * stores them per workgroup in `ws` ([C][nparts][2] doubles, the layout of crn_batch_renorm_bwd's own first pass), so that
 * crn_batch_renorm_bwd_apply can run without that pass: one full read of dy and of x less per norm.  The order of the sums is
 * fixed (lane, wave, workgroup slot): results do not depend on scheduling.
 * fuse->nparts is set to 0 when the launch could not produce the sums (split reduction, strided / non-dense y or x, more
 * workgroups than `ws` holds): the caller then runs crn_batch_renorm_bwd; y is written either way.  dsum (may be NULL) is
 * zeroed by the launch as crn_batch_renorm_bwd's first pass would.                                                       */
typedef struct crnBnBwdFuse {
  const float* x;      /* input of the norm: dense [B][C][D*H*W] inside each sample, D, H, W, C of y */
  int64_t sB_x;
  const float* saved;  /* [4][C] of crn_batch_renorm_stats: mu, rstd, r, d */
  int pre_relu;        /* the norm sees max(x, 0) */
  double* ws; size_t ws_bytes;
  float* dsum; int ndsum;
  int nparts;          /* out */
} crnBnBwdFuse;
int crn_conv_fwd_bf3_slabs_bnbwd(const crnView* x, const crnInTransform* tr, const void* wslab, int Npad,
                                 const float* bias, int bias_sB, const crnView* y,
                                 int kd, int kh, int kw, int pd, int ph, int pw,
                                 int accumulate, const crnTapBoxes* boxes /* may be NULL */, crnBnBwdFuse* fuse,
                                 crnStream stream);

/* ... or the partial sums of the BatchRenorm BEHIND the convolution (reconstruction_decoder.py:56-60): sum(y'), sum(y'^2) per
 * channel and workgroup, y' = max(y, 0) with pre_relu, in ws[(n * *nparts + i) * 2 + {0,1}] for crn_batch_renorm_finalize.
 * *nparts = 0 when the launch could not produce them (as above): the caller runs crn_batch_renorm_stats.          */
int crn_conv_fwd_bf3_slabs_stats(const crnView* x, const crnInTransform* tr, const void* wslab, int Npad,
                                 const float* bias, int bias_sB, const crnView* y,
                                 int kd, int kh, int kw, int pd, int ph, int pw,
                                 int accumulate, const crnTapBoxes* boxes /* may be NULL */, int pre_relu,
                                 double* ws, size_t ws_bytes, int* nparts, crnStream stream);

/* A convolution that splits its reduction writes partial sums to the library's scratch and adds them up in a
 * second launch.  crn_splitk_defer(1) arms, for the NEXT convolution call of this host thread (crn_conv2d_bf3, or
 * the 1x1 path of crn_conv_fwd), the following shortcut: if that call splits and does not accumulate, the sum is
 * left pending (recorded per host thread, together with the stream and device the convolution ran on) and a
 * crn_batch_renorm_stats (x = the conv's output) or crn_batch_renorm_bwd (dy = the conv's output) that follows ON THE
 * SAME STREAM adds the partial sums up while it loads them (stats also stores the sum to x; bwd does not write dy: it
 * is its only reader).  EVERY other entry point of this library that takes a stream -- and a BatchRenorm call on
 * another tensor or another stream -- first completes a pending sum with an ordinary reduction launch on the
 * convolution's own stream, so arming is always safe for callers that reach the tensor through this library on that
 * host thread; it only pays when the BatchRenorm call follows directly (resnet50.py:62-69: every conv of the encoder is
 * followed by its norm; in backward every data gradient feeds the norm's backward).  A caller that hands y to code
 * OUTSIDE the library (a torch op, a memcpy) or to another host thread while a sum may be pending must not arm.   */
int crn_splitk_defer(int on);
/* Split-K scratch is kept per (device, stream) and grown on demand -- which a stream under HIP-graph capture cannot do.
 * Call this BEFORE the capture: reserves `floats` (or, with 0, as much as any stream of the device has needed so far).  */
int crn_splitk_reserve(int64_t floats, crnStream stream);
/* Releases the scratch of `stream` (call before destroying a stream that was given one; waits for the device).  The table of
 * per-stream scratch buffers holds 64 entries per process and is guarded by a mutex.                                      */
int crn_splitk_release(crnStream stream);

/* Weight gradient in the same packed layout:
 *   dw[(c*T+t)*Npad+n] = sum_{b,o} T(x)[b,c,o-pad_lo+t] * dy[b,n,o]
 * dw must be zeroed by the caller or zero_first!=0.  Replaces autograd of the
 * modules above.                                                             */
int crn_conv_wgrad(const crnView* x, const crnInTransform* tr, const crnView* dy,
                   float* dw, int Npad, int kd, int kh, int kw, int pd, int ph, int pw,
                   int zero_first, const crnTapBoxes* boxes /* may be NULL; entries of dw at structural zeros may be left untouched */,
                   crnStream stream);

/* The weight gradient on the split-bf16 MFMA engine (csrc/conv_bf3.hip; see crn_conv_fwd_bf3): same packed
 * dw, same zero_first contract; CRN_EINVAL for shapes it does not cover (windows other than 5^3 / 4^3, dy W neither
 * a multiple of 16 nor 8, views other than unit-stride x and unit-stride / stride-2 space-to-depth dy).       */
int crn_conv_wgrad_bf3(const crnView* x, const crnInTransform* tr, const crnView* dy,
                       float* dw, int Npad, int kd, int kh, int kw, int pd, int ph, int pw,
                       int zero_first, crnStream stream);
/* ... with the tap boxes of the output columns (transposed convolutions): window rows that are structural zeros for all
 * columns of a workgroup are not multiplied (4^3 windows; their dw entries are not touched).                        */
int crn_conv_wgrad_bf3_boxes(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad,
                             int kd, int kh, int kw, int pd, int ph, int pw, int zero_first,
                             const crnTapBoxes* boxes /* may be NULL */, crnStream stream);

/* Encoder engine (csrc/conv_e2d.hip): the ResNet-50 encoder's stride-1 Conv2d 1x1 / 3x3 layers (resnet50.py:49-115;
 * the stride of a down-sampling block is applied by crn_stride2_gather before its first 1x1) at batch sizes where
 * a layer is 1 GFLOP: split-bf16 MFMA like crn_conv_fwd_bf3, with the weights pre-arranged in MFMA operand order.
 *
 * crn_bf3_operands: packed fp32 weights [Cin][T][Npad] (the layout crn_conv_fwd reads) -> operand blocks, for
 * `nlayers` layers in one launch.  desc (DEVICE, int64 [nlayers][7]) = (first float of the layer in `packed`,
 * first 32-byte entry of the layer in `out`, Cin, T taps, Npad (multiple of 16), first workgroup of the layer,
 * KHW); ceil(entries/256) workgroups per layer, total_blocks = their sum.  An entry is 8 bf16 hi terms followed by
 * 8 bf16 lo terms of 8 consecutive input channels (w = hi + lo, hi = bf16(w)).
 * KHW == 0, encoder engine (Cin % 32 == 0): (Cin/32)*T*(Npad/16)*64 entries, entry ((cb*T + t)*(Npad/16) + ntile)*64
 *   + kk*16 + i = output column 16*ntile + i, tap t, channels 32*cb + 8*kk .. +7.
 * KHW == kh*kw, decoder engine (crn_conv_fwd_bf3_slabs): ceil(Cin/8)*kd*TP*Npad entries, TP = KHW rounded up to 4,
 *   entry ((chunk*kd + zd)*TP + tp)*Npad + n = output column n, tap zd*KHW + tp (zeros for tp >= KHW), channels
 *   8*chunk .. +7 (zeros past Cin).                                                                              */
int crn_bf3_operands(const float* packed, const int64_t* desc, int nlayers, int64_t total_blocks, void* out,
                     crnStream stream);
/* y = bias + window correlation of T(x) with the operand blocks `wop` of one layer (same operation, transform and
 * bias contract as crn_conv_fwd; forward pass, or data gradient with the packed data-gradient weights).
 * Covers: window 1x1 (pads 0) or 3x3, dense NC(D)HW views (unit W stride, rows and planes back to back), x and y of
 * the same extent, Cin % 32 == 0, Cout % 64 == 0 (= Npad), H*W a multiple of 64 (3x3: W % 8 == 0 and tiles of
 * 4x16 / 8x8 positions divide the image); CRN_EINVAL otherwise (callers keep such layers on crn_conv_fwd).
 * Split-K partial sums use the per-stream scratch shared with crn_conv_fwd (crn_splitk_reserve).                */
int crn_conv2d_bf3(const crnView* x, const crnInTransform* tr, const void* wop, int Npad, const float* bias,
                   int bias_sB, const crnView* y, int kh, int kw, int ph, int pw, int accumulate, crnStream stream);

/* Weight gradient of a 1x1 layer on the split-bf16 MFMA (csrc/conv_e2d.hip: both operands straight from HBM, K =
 * positions): dw[c*Npad + n] += sum_{b,p} T(x)[b,c,p] * dy[b,n,p], the contract of crn_conv_wgrad for a 1x1x1 window
 * (Conv2d 1x1 of resnet50.py:62-69; dw zeroed by the caller or zero_first).  Dense NC(D)HW views with the same
 * extent, positions per sample a multiple of 32; CRN_EINVAL otherwise.                                          */
int crn_conv_wgrad_1x1_bf3(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad,
                           int zero_first, crnStream stream);
/* ... and of a 1x1 (pads 0) or 3x3 (pads 1, W a multiple of 8, 2-D images) layer: for 3x3 the three zw taps of an
 * input row come from one aligned 8-position load plus its two edge elements (dw row = c*9 + zh*3 + zw).        */
int crn_conv_wgrad_2d_bf3(const crnView* x, const crnInTransform* tr, const crnView* dy, float* dw, int Npad,
                          int kh, int kw, int ph, int pw, int zero_first, crnStream stream);

/* dst[i] = idx[i] >= 0 ? src[idx[i]] : 0        (weight packing)            */
/* Tiled index copy between a reference-layout buffer and a packed buffer (weight pack / gradient un-pack).
 * Tile t covers packed positions desc[t][0] + r*desc[t][1] + c (r, c in 0..7); bit (r*8+c) of mask[t] says
 * whether the element exists; its reference-layout index is desc[t][2] + r*desc[t][3] + c*desc[t][4], or
 * explicit_idx[desc[t][5] + r*8 + c] when desc[t][5] >= 0.
 * reverse == 0: dst[packed position] = src[index];  reverse != 0: dst[index] = src[packed position].       */
int crn_copy_tiles_f32(const float* src, float* dst, const int32_t* desc /* [ntiles][6] */,
                       const uint64_t* mask /* [ntiles] */, const int32_t* explicit_idx, int64_t ntiles,
                       int reverse, crnStream s);
/* The same copy for index maps that are (batches of) 2-D transposes -- every plain convolution's forward and
 * data-gradient layout -- staged through LDS so that both sides move in runs of 64+ floats.  Block t holds
 * G x A x B elements (g, a, b) with  reference index = fbase + g*fg + a*fa + b,  packed position = pbase + g*pg + a*pa
 * + b*pb  (pa == 1 or pb == 1);  desc[t] = {A, B, G, fbase, fa, fg, pbase, pa, pb, pg, ceil(2^32/(A*B)), ceil(2^32/B),
 * ceil(2^32/A), 0, 0, 0} (the reciprocals as uint32; unused where the divisor is 1),  G*A*(B|1) <= 8448.
 * conv_geometry.mat_index derives the blocks from the same index arrays as the tiles and proves them equal.      */
int crn_copy_mats_f32(const float* src, float* dst, const int32_t* desc /* [nblocks][16] */, int64_t nblocks,
                      int reverse, crnStream s);
int crn_gather_f32(const float* src, const int32_t* idx, float* dst, int64_t n, crnStream s);
/* dst[idx[i]] (+)= src[i] for idx[i] >= 0      (gradient un-packing)        */
int crn_scatter_f32(const float* src, const int32_t* idx, float* dst, int64_t n,
                    int accumulate, crnStream s);
/* bias gradient: db[c] = sum_{b,s} dy[b,c,s]  (dy: [B][C][S], batch stride sB) */
int crn_bias_grad(const float* dy, int B, int C, int64_t S, int64_t sB, float* db,
                  int accumulate, double* workspace /* crn_batch_renorm_workspace_bytes(C) */,
                  size_t workspace_bytes, crnStream s);

/* ---------------- BatchRenorm (batch_renorm.py:33-62) ----------------------
 * x: [B][C][S] with batch stride sB (spatial contiguous).  pre_relu: statistics
 * of max(x,0) (decoder blocks are ReLU->BN->conv).
 * train: batch mean / biased var -> r,d clamps from num_batches_tracked,
 * running-stat update (Q3,Q4); writes per-channel
 *   scale = gamma*r/sigma_b, shift = beta + gamma*(d - mu*r/sigma_b)
 * and saves mu, 1/sigma_b, r, d for the backward.  eval: running stats.          */
int crn_batch_renorm_stats(const float* x, int B, int C, int64_t S, int64_t sB, int pre_relu,
                           const float* gamma, const float* beta,
                           float* running_mean, float* running_var,
                           const int64_t* num_batches_tracked /* read only; see crn_add_i64 */,
                           float eps, float momentum, int training,
                           float* scale, float* shift, float* saved /* [4][C] */,
                           double* workspace, size_t workspace_bytes, crnStream s);
size_t crn_batch_renorm_workspace_bytes(int C);
/* crn_batch_renorm_stats (pre_relu = 0) followed by crn_affine_add_relu(x, scale, shift, r, rscale, rshift, ...) -- the tail
 * of a ResNet bottleneck (resnet50.py:71-78: bn of the last conv + shortcut + ReLU) -- as ONE call: where a workgroup owns
 * a channel with all of it in registers (training, B*S <= 16384: every bottleneck of the encoder) the tail is written by
 * the statistics launch itself, otherwise by a second launch.  Same results as the two calls, bit for bit.
 * y2 (may be NULL): additionally the stride-2 compaction of y, y2[b,c,i,j] = y[b,c,2i,2j] (what crn_stride2_gather(y, ...) gives:
 * the input of the down-sampling block that follows, resnet50.py:94-97), W = row width of a channel plane (S = H * W; y dense
 * [B][C][H][W]); written from the same registers when W is a multiple of 4, by a gather launch otherwise.               */
int crn_batch_renorm_stats_tail(const float* x, int B, int C, int64_t S, int64_t sB,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                const int64_t* num_batches_tracked, float eps, float momentum, int training,
                                float* scale, float* shift, float* saved, double* workspace, size_t workspace_bytes,
                                const float* r, const float* rscale, const float* rshift, int64_t sB_r,
                                float* y_pre, int64_t sB_pre, float* y, int64_t sB_y, int relu,
                                float* y2, int W, crnStream s);

/* Eval mode (batch_renorm.py:59: (x - running_mean) / sqrt(running_var + eps) * weight + bias) for all
 * BatchRenorm instances of a model at once: n channels, table [n][5] = offsets of (weight, bias) in `params`,
 * of (running_mean, running_var) in `buffers`, and of the channel in the `scale` / `shift` output slabs.  */
int crn_batch_renorm_eval_affine(const float* params, const float* buffers, const int32_t* table,
                                 int n, float eps, float* scale, float* shift, crnStream s);

/* Backward.  With x' = pre_relu ? max(x,0) : x, xn = (x'-mu)*rstd,
 * out = x'*scale+shift, g = dy * (post_relu ? out>0 : 1):
 *   dbeta = sum g ; dgamma = sum g*(r*xn+d)
 *   dx = gamma*r*rstd*(g - mean(g) - xn*mean(g*xn)) * (pre_relu ? x>0 : 1)
 * saved = [4][C]: mu, rstd, r, d written by crn_batch_renorm_stats(training).
 * dgamma/dbeta are accumulated when accumulate != 0.
 * dsum (optional, may be NULL): dsum[c] = sum_{b,s} dx[b,c,s] for c < ndsum -- the bias
 * gradient of the convolution whose output x is (autograd's sum over the conv output
 * gradient, conv bias + norm in resnet50.py:40-61 / reconstruction_decoder.py:33-71),
 * fused here so that dx is not read again by crn_bias_grad.                        */
int crn_batch_renorm_bwd(const float* x, int64_t sB_x, const float* dy, int64_t sB_dy,
                         int B, int C, int64_t S, int pre_relu, int post_relu,
                         const float* gamma, const float* scale, const float* shift,
                         const float* saved, float* dx, int64_t sB_dx,
                         float* dgamma, float* dbeta, int accumulate,
                         float* dsum, int ndsum,
                         double* workspace, size_t workspace_bytes, crnStream s);

/* The second pass of crn_batch_renorm_bwd alone: `workspace` already holds the per-channel partial sums
 * [C][nparts][2] (sum g, sum g*xn) written by crn_conv_fwd_bf3_slabs_bnbwd (fuse->nparts > 0) on the same stream.
 * post_relu must be 0 (the fused sums do not mask).  Same dx / dgamma / dbeta / dsum as crn_batch_renorm_bwd up to the
 * summation order of the two sums.                                                                                */
int crn_batch_renorm_bwd_apply(const float* x, int64_t sB_x, const float* dy, int64_t sB_dy,
                               int B, int C, int64_t S, int pre_relu,
                               const float* gamma, const float* scale, const float* shift,
                               const float* saved, float* dx, int64_t sB_dx,
                               float* dgamma, float* dbeta, int accumulate,
                               float* dsum, int ndsum,
                               double* workspace, size_t workspace_bytes, int nparts, crnStream s);

/* crn_relu_bwd_add(g, act, g2 -> dy) followed by crn_batch_renorm_bwd(x, dy, pre_relu = post_relu = 0, ...) -- the backward
 * of a bottleneck's tail and of its last norm -- as ONE call: dy = (act > 0 ? g : 0) + g2 (g2 may be NULL) is formed and
 * stored by the norm's backward launch where a workgroup owns a channel in registers (B*S <= 16384), by a launch of its
 * own otherwise.  Same results as the two calls, bit for bit.
 * g_compact (may be NULL): g is the data gradient of the stride-2 1x1 convolutions of the down-sampling block behind this one
 * (resnet50.py:94-97), given in its compact form [B][C][ceil(H/2)][ceil(W/2)] (g = crn_stride2_scatter(g_compact), rows of width W,
 * S = H * W): the register form reads it from there and never touches `g`; otherwise the call expands it into `g` (which must
 * then be a dense [B][C][S] buffer) first.                                                                            */
int crn_batch_renorm_bwd_head(const float* x, int64_t sB_x, float* dy, int64_t sB_dy,
                              const float* g, int64_t sB_g, const float* act, int64_t sB_act,
                              const float* g2 /* may be NULL */, int64_t sB_g2,
                              int B, int C, int64_t S, const float* gamma, const float* scale, const float* shift,
                              const float* saved, float* dx, int64_t sB_dx, float* dgamma, float* dbeta,
                              int accumulate, float* dsum, int ndsum,
                              double* workspace, size_t workspace_bytes, const float* g_compact, int W, crnStream s);

/* y = act( x*scale[c]+shift[c] [+ r*rscale[c]+rshift[c]] ) ; optional second
 * output y_pre (before the final ReLU).  Encoder block tails
 * (resnet50.py:72-82,109-115).                                               */
int crn_affine_add_relu(const float* x, const float* scale, const float* shift,
                        const float* r, const float* rscale, const float* rshift,
                        int B, int C, int64_t S, int64_t sB_x, int64_t sB_r,
                        float* y_pre, int64_t sB_pre, float* y, int64_t sB_y,
                        int relu, crnStream s);

/* dx = dy * (y_pre > 0) [+ dy2]  : backward of the block-tail ReLU, merging the
 * gradient arriving through the skip connection (dy2 may be NULL).           */
int crn_relu_bwd_add(const float* dy, const float* y_pre, const float* dy2,
                     int B, int C, int64_t S, int64_t sB_dy, int64_t sB_pre, int64_t sB_dy2,
                     float* dx, int64_t sB_dx, crnStream s);

/* ---------------- the encoder's stem on its own kernels ----------------------
 * ZeroPad2d(3) + Conv2d(3 -> 64, 7x7, stride 2) of resnet50.py:122-124 on the preprocessed image img [B,3,H,W]
 * (H even, W % 8 == 0; CRN_EINVAL otherwise: the caller keeps crn_conv_fwd / crn_conv_wgrad on the 2x2 space-to-depth
 * view), y / dy [B,64,H/2,W/2] dense.  w_packed / dw_packed: the layer's packed weights / weight gradient as
 * crn_conv_fwd / crn_conv_wgrad see them ([12][16][64], conv_geometry.stem_fwd), bias [64] or NULL.
 * stats_ws != NULL: the forward also leaves sum(y), sum(y^2) per channel and workgroup in
 * stats_ws[(n * parts + i) * 2 + {0,1}], parts = crn_stem_conv_parts(B, H, W) (0: shape not covered), for
 * crn_batch_renorm_finalize -- the BatchRenorm of resnet50.py:125 without a statistics pass over y.
 * crn_stem_conv_wgrad ADDS to dw_packed (atomics); CRN_EINVAL in deterministic mode.                              */
size_t crn_stem_conv_parts(int B, int H, int W);
int crn_stem_conv_fwd(const float* img, int B, int H, int W, const float* w_packed, const float* bias,
                      float* y, double* stats_ws /* may be NULL */, size_t ws_bytes, crnStream s);
int crn_stem_conv_wgrad(const float* img, int B, int H, int W, const float* dy, float* dw_packed, crnStream s);
/* scale / shift / saved / running statistics of a BatchRenorm (batch_renorm.py:41-57, training mode) from partial sums
 * ws[(c * nparts + i) * 2 + {0,1}] = sum(x), sum(x^2) of part i; count = elements per channel.                   */
int crn_batch_renorm_finalize(const double* ws, int nparts, int C, double count,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              const int64_t* nbt, float eps, float momentum,
                              float* scale, float* shift, float* saved, crnStream s);

/* ---------------- encoder odds and ends -------------------------------------
 * preprocess_image_caffe (resnet50.py:189-204): u8 RGB -> f32 BGR + means.    */
int crn_preprocess_caffe(const uint8_t* img, int B, int H, int W, float* out, crnStream s);
/* ZeroPad2d(1)+MaxPool2d(3,2) on relu(x*scale+shift)  (resnet50.py:126-131).
 * argmax (int32 flat input index, -1 when the max is the zero pad) is saved. */
int crn_bn_relu_maxpool_fwd(const float* x, const float* scale, const float* shift,
                            int B, int C, int H, int W, float* y, int32_t* argmax, crnStream s);
int crn_bn_relu_maxpool_bwd(const float* dy, const int32_t* argmax, int B, int C, int H, int W,
                            float* dx_bn /* grad wrt bn output, relu applied */, crnStream s);
/* avg[b,c] = mean_s relu(x_pre[b,c,s]) (resnet50.py:183) and its backward,
 * added into dx.                                                              */
int crn_relu_mean_fwd(const float* x_pre, int B, int C, int64_t S, int64_t sB, float* avg, crnStream s);
/* y[b,n] = bias[n] + sum_k x[b,k] w[n,k]  (nn.Linear, reconstruction_decoder.py:49) */
int crn_linear_fwd(const float* x, const float* w, const float* bias, int B, int K, int N,
                   float* y, int ldy, crnStream s);
int crn_linear_bwd(const float* x, const float* w, const float* dy, int lddy, int B, int K, int N,
                   float* dx, float* dw, float* db, crnStream s);
/* dx[b,c,s] (+)= (x_pre>0) * davg[b,c] / S                                    */
int crn_relu_mean_bwd(const float* x_pre, const float* davg, int B, int C, int64_t S, int64_t sB,
                      float* dx, int64_t sB_dx, int accumulate, crnStream s);
/* Stride-2 sub-sampling of a contiguous [B][C][hin][win] tensor, y[b,c,i,j] = x[b,c,2i,2j] with h = ceil(hin/2),
 * w = ceil(win/2), and its adjoint (dx[b,c,2i,2j] = dy[b,c,i,j], zeros elsewhere; every element of dx is written): the
 * stride-2 1x1 convolutions of the ResNet downscale blocks (resnet50.py:94-97) run on the compacted tensor.           */
int crn_stride2_gather(const float* x, float* y, int B, int C, int h, int w, int hin, int win, crnStream s);
int crn_stride2_scatter(const float* dy, float* dx, int B, int C, int h, int w, int hin, int win, crnStream s);
/* fill channels [c0,c0+3) of a [B][Ctot][S] tensor with offset[b][j]
 * (reconstruction_decoder.py:108-110)                              */
int crn_fill_offset_channels(float* x, int B, int64_t sB, int64_t S, int c0,
                             const float* offset, crnStream s);

/* The decoder's per-call inputs in one launch: layer_mats[s][b] = v2s[b] . scale(f_s, f_s, f_s) for the nscales <= 4 skip grids
 * (reconstruction_decoder.py:111-116: layer_matrix = v2s . scale(resolution / grid); columns 0-2 of the row-major 4x4 times f_s,
 * exact) and offset_out = offset.  v2s [B][16], offset [B][3] device; scales: HOST array of nscales floats.                   */
int crn_decoder_inputs(const float* v2s, const float* offset, int B, int nscales, const float* scales,
                       float* layer_mats, float* offset_out, crnStream s);

/* ---------------- ray-traced skip connection --------------------------------
 * Gather part of SampleGrid2d.forward (ray_traced_skip_connection.py:91-144):
 * per voxel centre (x,y,z)+off: p = M*(.,1); u=(px/pw)/2+.5; ix=(int)(u*W);
 * iy likewise (C truncation); +1 and clamp into the 1-px zero pad;
 * zero when p.z < 0 (Q7).  map: element (b, c, iy, ix) at b*map_sB + c*map_sC +
 * (iy*w+ix)*map_sP: (h*w, 1) is [B][C][h][w]; (1, C) is the channel-last [B][h][w][C]
 * that the compress conv can write and that is gathered four channels per load.
 * out: channels [0,C) of a view with batch stride out_sB, spatial D*H*W.
 * matrix: [B][16] row-major layer matrix, offset: [B][3].                     */
int crn_ray_sample_fwd(const float* map, int64_t map_sB, int64_t map_sC, int64_t map_sP,
                       int B, int C, int h, int w, const float* matrix, const float* offset,
                       float* out, int64_t out_sB, int D, int H, int W, crnStream s);
/* Backward (reference: autograd index_put_(accumulate=True)): dmap must be
 * zeroed by the caller unless zero_first.  dmap is channel-major [B][C][h][w] with batch stride dmap_sB.
 * Without a saved index tensor the library projects into its own scratch first (crn_ray_project).          */
int crn_ray_sample_bwd(const float* dout, int64_t dout_sB, int B, int C, int D, int H, int W,
                       const float* matrix, const float* offset,
                       float* dmap, int64_t dmap_sB, int h, int w, int zero_first, crnStream s);
/* The saved index tensor.  The reference's forward builds the index tensors of its advanced indexing
 * (ray_traced_skip_connection.py:118-135) and autograd keeps them for the index_put_ of the backward pass; here the
 * forward leaves ONE uint16 per voxel, idx[b][z][y][x] = iy*w+ix of the pixel the voxel reads, 0xFFFF = the outside
 * value (behind the camera or off the image), and the backward scatters from it without projecting again (no
 * floating-point arithmetic but the sums themselves).  Needs h*w < 65535 (CRN_EINVAL otherwise: use the plain entry
 * points above).  crn_ray_project writes the index tensor alone.                                            */
int crn_ray_sample_fwd_idx(const float* map, int64_t map_sB, int64_t map_sC, int64_t map_sP,
                           int B, int C, int h, int w, const float* matrix, const float* offset,
                           float* out, int64_t out_sB, int D, int H, int W, uint16_t* idx, crnStream s);
int crn_ray_project(const float* matrix, const float* offset, int B, int D, int H, int W, int h, int w,
                    uint16_t* idx, crnStream s);
int crn_ray_sample_bwd_idx(const float* dout, int64_t dout_sB, int B, int C, int D, int H, int W,
                           const uint16_t* idx, float* dmap, int64_t dmap_sB, int h, int w, int zero_first,
                           crnStream s);

/* ---------------- losses (losses.py) ---------------------------------------
 * kind: 0 iou_fgbg (:64-114), 1 xent_times_iou_agnostic (:144-160),
 *       2 iou_agnostic (:19-61), 3 xent (:117-141), 4 xent_times_iou_fgbg.
 * logits [B][C][S] fp32, gt [B][S] int32 labels, weights [B][S] fp32 per-voxel
 * loss weights or NULL (losses.py:47-49,99-102,134-136).  Writes loss[0] and,
 * when dlogits != NULL, d loss / d logits * grad_scale.
 * workspace: crn_loss_workspace_bytes(B,C).  A label outside [0,C) (the
 * reference raises inside F.one_hot / cross_entropy) is computed as class 0 and
 * sets the int at crn_loss_status_ptr(workspace, B) (device memory) to 1.      */
int crn_loss_fwd_bwd(int kind, const float* logits, const int32_t* gt, const float* weights,
                     int B, int C, int64_t S, float* loss, float* dlogits, float grad_scale,
                     void* workspace, size_t workspace_bytes, crnStream s);
size_t crn_loss_workspace_bytes(int B, int C);
const int* crn_loss_status_ptr(void* workspace, int B);

/* ---------------- eval epilogue ---------------------------------------------
 * argmax over classes + confusion matrix (evaluation_results.py:40-51,
 * voxel_metrics.py:33-58): cm[gt*K+pred] += 1 (int64).  labels may be NULL.   */
int crn_argmax_confusion(const float* logits, const int32_t* gt, int B, int C, int64_t S,
                         int32_t* labels, int64_t* cm, crnStream s);

/* Multi-offset inference epilogue (super_resolution.py:92-125): logits [m^3][B][C][D][H][W], one
 * native-resolution forward per sampling offset n = (iz*m+iy)*m+ix (offset = (ix,iy,iz)/m);
 * out [B][C][mD][mH][mW] = softmax over C, interleaved: out[b][c][z*m+iz][y*m+iy][x*m+ix].
 * m == 1: plain channel softmax (pipeline inference_fn, super_resolution.py:121-126).          */
int crn_softmax_superres(const float* logits, int m, int B, int C, int D, int H, int W, float* out,
                         crnStream s);

/* ---------------- optimizer --------------------------------------------------
 * torch.optim.Adam step (state.py:65; train hot loop pipeline.py:230) on a flat
 * fp32 parameter slab.  grad_scale multiplies the gradient first (1/world).   */
int crn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  int64_t n, float lr, float beta1, float beta2, float eps,
                  float grad_scale, int step, crnStream s);
/* The same step with its scalars read from device memory, for HIP-graph capture of the training step
 * (the replayed launch carries no per-step host value): hyper = 7 floats written by crn_adam_set_hyper
 * (lr, beta1, beta2, eps, grad_scale, 1 - beta1^step, sqrt(1 - beta2^step)) with an ordinary launch
 * ordered before the graph replay.                                                                     */
int crn_adam_set_hyper(float* hyper, float lr, float beta1, float beta2, float eps, float grad_scale,
                       int step, crnStream s);
int crn_adam_step_hyper(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                        int64_t n, const float* hyper, crnStream s);

/* Decoder stage_6.t1 -- ConvTranspose3d(16 -> Cout <= 16, k 7, stride 2, padding 3, output_padding 1), the layer that writes the
 * logits (model/reconstruction_decoder.py:89-95) -- on its own parity-walk kernels of the split-bf16 MFMA engine
 * (csrc/convt_par.hip): the input patch of a 4 x 8 x 16 tile stays in LDS while the workgroup walks the 8 output parities over
 * exactly the window rows that hold weights.  Same arithmetic as crn_conv_fwd_bf3 on the layer's window-correlation form.
 * wimg: the weights as a step-ordered image of bf16 hi / lo entries, built by crn_bf3_gather_image from any fp32 array that
 * holds the reference weight [16][Cout][7][7][7] (e.g. the flat parameter slab) through a device table of int32 [n][10] rows
 * (8 source indices, -1 = zero; destination entry of the hi part; of the lo part: corenet_amd/model/conv_geometry.py
 * convt_par_fwd_table / convt_par_dgrad_table).
 * fwd:   x [B][16][D][H][W] (dense inside a sample, batch stride x_sB) with the input transform tr -> y [B][..][2D][2H][2W]
 *        channels [0, Cout) (+ bias[n]), strides y_sB / y_sC.  D % 4 == H % 8 == W % 16 == 0.
 * dgrad: dy [B][..][2D][2H][2W] channels [0, Cout) -> dx [B][16][D][H][W] (accumulate: dx +=).                              */
int crn_bf3_gather_image(const float* src, const int32_t* table, int n_entries, void* dst, crnStream s);
int crn_convt_s2k7_fwd_bf3(const float* x, int64_t x_sB, int B, int D, int H, int W, const crnInTransform* tr,
                           const void* wimg, const float* bias, float* y, int64_t y_sB, int64_t y_sC, int Cout, crnStream s);
int crn_convt_s2k7_dgrad_bf3(const float* dy, int64_t dy_sB, int64_t dy_sC, int Cout, int B, int D, int H, int W,
                             const void* wimg, float* dx, int64_t dx_sB, int accumulate, crnStream s);
/* The same two operations for exactly 2 classes (h7), where 8 parities x 2 classes are one 16-column block and the layer's
 * weights (64 KB as bf16 hi / lo) stay resident in LDS: one persistent workgroup per CU walks its tiles (convt_res_kernel).
 * Images: conv_geometry.convt_res_fwd_table / convt_res_dgrad_table.                                                     */
int crn_convt_s2k7_c2_fwd_bf3(const float* x, int64_t x_sB, int B, int D, int H, int W, const crnInTransform* tr,
                              const void* wimg, const float* bias, float* y, int64_t y_sB, int64_t y_sC, int Cout, crnStream s);
int crn_convt_s2k7_c2_dgrad_bf3(const float* dy, int64_t dy_sB, int64_t dy_sC, int Cout, int B, int D, int H, int W,
                                const void* wimg, float* dx, int64_t dx_sB, int accumulate, crnStream s);
/* wgrad of the same layer (autograd of reconstruction_decoder.py:89-95): dw += sum over (b, q) of T(x)[b, c, q + z - 1] *
 * dy[b, n, 2 q + r] into the layer's PACKED gradient [16][64 window taps][Npad] (parity-major columns: the layout crn_conv_wgrad
 * writes for the window-correlation form, conv_geometry.convt_fwd), partial sums over position slices added with atomics;
 * zero_first clears dw.  D % 2 == H % 8 == W % 16 == 0.  Returns CRN_EINVAL in deterministic mode (crn_set_deterministic):
 * the caller then takes crn_conv_wgrad_bf3_boxes, whose unsplit form has a fixed order.
 * ximg (may be NULL): the operand image of T(x) left by crn_convt_s2k7_ximage for the same x and tr -- the workgroups then copy
 * their patches from it instead of transforming and splitting x themselves (same operands bit for bit, same result).     */
int crn_convt_s2k7_wgrad_bf3(const float* x, int64_t x_sB, int B, int D, int H, int W, const crnInTransform* tr,
                             const float* dy, int64_t dy_sB, int64_t dy_sC, int Cout, float* dw, int Npad, int zero_first,
                             const void* ximg, crnStream s);
/* Operand image of that layer's input for the weight gradient: img[b][chunk * 2 + (hi, lo)][D * H * W] entries of 8 bf16 = the
 * split-bf16 halves of T(x)[b, chunk * 8 .. + 7, position] (T = the BatchRenorm + ReLU in front of the layer,
 * reconstruction_decoder.py:56-60, 89-95).  One pass: x read once, crn_convt_s2k7_ximage_bytes(B, D, H, W) written.     */
size_t crn_convt_s2k7_ximage_bytes(int B, int D, int H, int W);
int crn_convt_s2k7_ximage(const float* x, int64_t x_sB, int B, int D, int H, int W, const crnInTransform* tr,
                          void* img, size_t img_bytes, crnStream s);

/* ---------------- ground-truth side -------------------------------------------
 * fill_inside_voxels_gpu (cc/fill_voxels_gpu.cu:136-171, module.cc:18-29):
 * grid [N][D][H][W] of dtype (0 f32, 1 u8, 2 i32, 3 f64, 4 i64, 5 i16, 6 i8),
 * contiguous.  out may alias grid (inplace=True).  Output strictly {0,1}
 * (SURVEY Q10).  Bit-exact with the reference semantics for every input.     */
int crn_fill_voxels(const void* grid, void* out, int dtype, int N, int D, int H, int W,
                    void* workspace, size_t workspace_bytes, crnStream s);
size_t crn_fill_voxels_workspace_bytes(int N, int D, int H, int W);
/* The same operator on strided views: the reference reads `grid` and writes `result` through packed accessors
 * (cc/fill_voxels_gpu.cu:146-163), so non-contiguous tensors -- and in-place calls on them -- need no copy.
 * grid_strides / out_strides: 4 host int64 element strides (N, D, H, W), all >= 0; same workspace, same single
 * asynchronous launch, voxels move one per lane.                                                              */
int crn_fill_voxels_strided(const void* grid, const int64_t* grid_strides, void* out, const int64_t* out_strides,
                            int dtype, int N, int D, int H, int W, void* workspace, size_t workspace_bytes, crnStream s);
/* crn_fill_voxels never waits for the GPU (the reference op is asynchronous too,
 * fill_voxels_gpu.cu:158-165): ONE kernel launch per call (per 256-CU load of
 * grids); a workgroup that gives up raises a device-side flag and the last
 * workgroup to leave the launch redoes its grids.  The launch's control words
 * live in a buffer owned by the library, keyed by `workspace`: the first call
 * with a new workspace allocates it (not capturable), later calls are.        */

/* fill_inside_voxels_cpu (cc/module.cc:24-29, cc/fill_voxels_cpu.cc:158-183):
 * HOST memory in and out (out may alias grid), same dtype codes.  Reference CPU
 * semantics (fill_voxels_cpu.cc:150-154): voxels outside the empty
 * region connected to the x==0 / y==0 / z==0 faces become 1, every other voxel
 * keeps its input value (the GPU op writes strict {0,1}).  Grids are processed
 * by num_threads host threads (<= 0: all hardware threads).                    */
int crn_fill_voxels_cpu(const void* grid, void* out, int dtype, int N, int D, int H, int W,
                        int num_threads);

/* voxelize_mesh (geometry/voxelization.py:32-164 + shaders/voxelize.{geom,frag}):
 * triangles [T][3][3] (view space), tri_mesh [T] mesh index per triangle
 * (misc_util.dynamic_tile), view2voxel [M][16]; grid [M][D][H][W] (or the
 * (2D+1)(2H+1)(2W+1) sub-grid when sub_grid_side>0) is zeroed then marked.    */
int crn_voxelize_mesh(const float* triangles, const int32_t* tri_mesh, int T,
                      const float* view2voxel, int M, int D, int H, int W,
                      int sub_grid_side, float image_resolution_multiplier,
                      int conservative, int depth_multiplier, float* grid, crnStream s);
/* batch() geometry (data/batched_example.py:73-81 -> geometry/transformations.py:139-169 transform_mesh):
 * out[t][k] = (mesh_matrix[tri_mesh[t]] . (triangles[t][k], 1)).xyz / .w for the object-space triangles
 * [T][3][3] of all meshes of a batch, mesh_matrix [M][16] = view_transform . object_to_world per mesh.
 * out may alias triangles.                                                                       */
int crn_transform_meshes(const float* triangles, const int32_t* tri_mesh, int T,
                         const float* mesh_matrix, int M, float* out, crnStream s);
/* per-scene label merge (batched_example.py:186-196): out[b] = max_m label_m*grid_m
 * over the meshes [scene_mesh_start[b], scene_mesh_start[b+1]) of scene b, int32.
 * sub_grid!=0: meshes_grid is the (2D+1)(2H+1)(2W+1) grid and the sub-grid
 * centres are taken (voxelization.get_sub_grid_centers :167-182).            */
int crn_merge_labels(const float* meshes_grid, const int32_t* scene_mesh_start,
                     const int32_t* mesh_label, int B, int D, int H, int W, int sub_grid,
                     int32_t* out, crnStream s);

/* ---------------- data parallelism: RCCL from the library ------------------------
 * The reference exchanges gradients through torch's DistributedDataParallel over NCCL (pipeline.py:199-200,229) and
 * broadcasts the BatchRenorm buffers from rank 0 (DDP broadcast_buffers).  Here the exchange can be enqueued by the
 * library itself on the caller's stream (the engine's side stream, behind the un-pack of a gradient bucket):
 *   rank 0: crn_comm_unique_id(id)  ->  the host hands the 128 bytes to every rank (torch.distributed store, MPI, ...)
 *   every rank (its GPU current): crn_comm_init(id, rank, nranks, &comm)
 *   crn_allreduce_f32(comm, buf, n, stream): in-place sum over the ranks, asynchronous, stream-ordered.
 * librccl.so is opened on first use; without it these calls return CRN_EINVAL and the rest of the library works.
 * RCCL picks ring / tree / direct by message size and topology; NCCL_ALGO / NCCL_PROTO (read by RCCL at communicator
 * creation) override it -- tools/scale_probe.sh times the engine's bucket sizes under each.                        */
int crn_comm_unique_id(void* id128 /* 128 bytes, host */);
int crn_comm_init(const void* id128, int rank, int nranks, void** comm);
int crn_comm_destroy(void* comm);
int crn_comm_info(void* comm /* may be NULL */, int* rccl_version, int* rank, int* nranks);
int crn_allreduce_f32(void* comm, float* buf, int64_t n, crnStream s);
int crn_broadcast_f32(void* comm, float* buf, int64_t n, int root, crnStream s);

/* rocprofv3 markers: a roctx range around the calls that follow (one per layer and direction when the Python engine
 * runs with CRN_ROCTX=1); `rocprofv3 --kernel-trace --marker-trace` then attributes kernels to layers.  No-ops
 * (CRN_EINVAL) when no roctx library can be opened.                                                                 */
int crn_roctx_push(const char* label);
int crn_roctx_pop(void);

/* Deterministic mode (also env CRN_DETERMINISTIC=1): every floating-point sum of the library is taken in an order that
 * does not depend on how workgroups are scheduled, so two runs from the same state are bit-identical -- a debugging
 * aid (the reference gets the same from torch.use_deterministic_algorithms; its index_put_(accumulate=True) and cuDNN
 * weight gradients are atomic-order dependent too).  Default mode: weight gradients, the bias gradient fused into the
 * BatchRenorm backward and the ray-sample scatter add partial sums with fire-and-forget atomics.  Deterministic mode:
 * one workgroup per weight-gradient element (slow: 10-50x for those launches), an ordered two-level reduction for the
 * bias gradient, 64-bit fixed-point accumulation (scale from max |dout|) for the scatter.  Process-wide.           */
int crn_set_deterministic(int on);

/* misc */
int crn_zero_f32(float* p, int64_t n, crnStream s);
/* p[i] += v, i < n : steps every BatchRenorm's num_batches_tracked (batch_renorm.py:57) */
int crn_add_i64(int64_t* p, int n, int64_t v, crnStream s);
const char* crn_version(void);

#ifdef CRN_TOOLS
/* Tuning aids: NOT in libcorenet_hip.so.  corenet_amd.build.build_tools() compiles the library a second time with
 * -DCRN_TOOLS into tools/_build/libcorenet_hip_tools.so for the scripts under tools/ (shader-clock stamps of workgroup 0 of
 * the last launch of the split-bf16 decoder kernels, CRN_BF3_STAMPS=1, the encoder engine, CRN_E2D_DBG=16, and the pointwise
 * kernel, CRN_PW_STAMPS=1; copied to the host after a device synchronize; CRN_EINVAL when the switch was not set).  The MFMA
 * hardware probe of DESIGN section 3e is tools/mfma_probe.hip -> tools/_build/libcrn_probe.so.                          */
int crn_bf3_debug_stamps(long long* out192);
int crn_e2d_debug_stamps(long long* out32);
int crn_pw_debug_stamps(long long* out32);
#endif

#ifdef __cplusplus
}
#endif
#endif  /* CORENET_HIP_H_ */
