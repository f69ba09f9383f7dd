/* libmsml_hip.so -- C ABI of the MI355X (gfx950) MSML hot path.
 *
 * This is the drop-in boundary: every entry point replaces a
 * torch.nn / torch.nn.functional / torch.distributed call site of the reference
 * (ygtxr1997/MSML), cited per function as <file>:<line> relative to the reference root.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers owned
 *     by the caller (PyTorch caching allocator); the library never allocates, frees,
 *     synchronises or changes the current device.
 *   - every function enqueues on `stream` (a hipStream_t passed as void*) and returns
 *     immediately: 0 on success, a negative MSML_ERR_* otherwise; the message is available
 *     from msml_last_error() (thread-local).  No C++ exception crosses the ABI.
 *   - activations are NHWC ("pixel-major"): [N][H][W][Cp], Cp = channel count rounded up to a
 *     multiple of 8 (the Python host uses 8 or a multiple of 32, which selects the LDS-DMA
 *     fast conv path); pad channels hold exact zeros.  `dtype` selects the STORAGE type of
 *     activations / packed weights: MSML_F32 (exact-f32 MFMA path, parity mode) or MSML_BF16
 *     (bf16 operands, f32 accumulate).  Parameters, statistics and gradients of parameters
 *     are always f32 in the reference's own layouts (OIHW etc.).
 *   - reentrant: no global mutable state besides the thread-local error string.
 */
#ifndef MSML_HIP_H
#define MSML_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSML_ABI_VERSION 1

enum { MSML_F32 = 0, MSML_BF16 = 1, MSML_BF16X3 = 2 };   /* BF16X3: split-bf16 planes, see msml_conv2d_x3 */

enum {
  MSML_OK = 0,
  MSML_ERR_SHAPE = -1,       /* bad / inconsistent dimensions */
  MSML_ERR_DTYPE = -2,       /* unsupported dtype enum */
  MSML_ERR_LAUNCH = -3,      /* hipGetLastError() after launch */
  MSML_ERR_UNSUPPORTED = -4, /* valid request this build has no kernel for */
  MSML_ERR_WORKSPACE = -5    /* workspace too small */
};

/* FM operator enums: backbones/fm/fmoperator.py:110-126 */
enum { MSML_ACT_TANH = 0, MSML_ACT_SIGMOID = 1 };
enum { MSML_ARITH_ADD = 0, MSML_ARITH_SUB = 1, MSML_ARITH_MUL = 2, MSML_ARITH_DIV = 3 };

int msml_version(void);
const char* msml_last_error(void);
/* 1 when the library was built with -DMSML_EXPERIMENTS (tools/build_variant.py --all MSML_EXPERIMENTS): the measured-slower
 * kernel variants of DESIGN.md section 8 are instantiated and their opt-in switches live (MSML_HALO_R15, MSML_BNBWD_IN,
 * MSML_BNIN_ACC_WS, MSML_HALO_WGRAD_S2).  The shipped library returns 0: those entry points answer
 * MSML_ERR_UNSUPPORTED / their *_applies queries 0. */
int msml_has_experiments(void);
/* Id of the graph capture `stream` is currently part of (hipStreamGetCaptureInfo; every stream forked into
 * one capture reports the same id), 0 when the stream is not capturing, negative on a HIP error.  The host
 * side keys per-capture scratch state on it (zeroed accumulator chunks: a second capture must not inherit the
 * first one's slices, whose zero fill only the first graph replays). */
long msml_stream_capture_id(void* stream);
/* `waiter` waits for everything queued on `src` so far: torch.cuda.Stream.wait_stream (the fork of the weight-gradient
 * side stream, backbones/msml.py has no counterpart: the reference runs one stream) through one re-recorded hipEvent per
 * calling thread.  Valid inside a graph capture (the waiter joins the capture). */
int msml_stream_wait_stream(void* waiter, void* src);

/* ---------------------------------------------------------------- layout (boundary) ------
 * The reference keeps NCHW f32 tensors end to end (backbones/msml.py:150); the HIP path
 * converts once at the MSML.forward boundary. */
int msml_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int Cp,
                      int dtype, void* stream);
int msml_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int Cp,
                      int dtype, void* stream);

/* Pack a 4-D f32 parameter w[A][B][R][S] (Conv2d: A=Cout,B=Cin; ConvTranspose2d: A=Cin,
 * B=Cout) into the GEMM operand the conv kernel streams: dst[KOp][Ktot] with
 *   transpose == 0: ko = a, ci = b      (conv forward, deconv backward-data)
 *   transpose == 1: ko = b, ci = a      (conv backward-data, deconv forward)
 * ci is split into up to two input segments (channel concat, e.g. cat(yf, yo)
 * fmoperator.py:285): segment 0 = ci in [0, C1), segment 1 = ci in [C1, C1+C2); each segment
 * is laid out [r][s][c] with c padded to C1p / C2p and the segment's K padded to a multiple
 * of 32 with zeros.  ko is padded to KOp rows of zeros.  Ktot = Kseg0 + Kseg1. */
int msml_pack_weight(const float* w, void* dst, int A, int B, int R, int S, int transpose,
                     int C1, int C1p, int C2, int C2p, int KOp, int dtype, void* stream);

/* ---------------------------------------------------------------- FM fusion ---------------
 * backbones/fm/fmoperator.py:288,304-310: M = act(x); z = arith(yf, M) + yf.
 * x, yf, z: [n] elements, storage dtype.  Algorithmic bytes: 3 * n * sizeof(dtype). */
int msml_fm_fuse_fwd(const void* x, const void* yf, void* z, long n, int act, int arith,
                     int dtype, void* stream);
/* backward: reads dz, x, yf; writes dx (grad of the pre-activation) and dyf.  5 streams. */
int msml_fm_fuse_bwd(const void* dz, const void* x, const void* yf, void* dx, void* dyf,
                     long n, int act, int arith, int dtype, void* stream);

/* Peer-guided branch of the FM operators (backbones/fm/fmoperator.py:293-308) and dropout
 * (backbones/frb/iresnet.py:231).  n elements (a multiple of 8), storage dtype.
 *   msml_fm_act_fwd / _bwd : M = act(x) as a tensor (the input of conv_m, :294-296) / dx = dM * act'(x)
 *   msml_mul_fwd / _bwd    : m_bar * identity, m_bar * yt (:297,300); da / db may be NULL
 *   msml_axpb              : y = a*x + b ('invert' mask transform 1 - M, :163-164)
 *   msml_mse_fwd / _bwd    : torch.nn.MSELoss()(f_occ, f_out) (:302): loss[0] = sum (a-b)^2 / count (f64
 *                            partial sums, fixed order; workspace >= 2048 doubles); da = 2 g (a-b) / count,
 *                            db = -da with g a device scalar
 *   msml_dropout           : y = x * keep / (1 - p), keep from a counter-based hash of (seed, index); the
 *                            backward is the same call on dy */
int msml_fm_act_fwd(const void* x, void* m, long n, int act, int dtype, void* stream);
int msml_fm_act_bwd(const void* dm, const void* x, void* dx, long n, int act, int dtype, void* stream);
int msml_mul_fwd(const void* a, const void* b, void* out, long n, int dtype, void* stream);
int msml_mul_bwd(const void* g, const void* a, const void* b, void* da, void* db, long n, int dtype,
                 void* stream);
int msml_axpb(const void* x, void* y, long n, float a, float b, int dtype, void* stream);
int msml_mse_fwd(const void* a, const void* b, long n, double count, float* loss, double* workspace,
                 long ws_doubles, int dtype, void* stream);
int msml_mse_bwd(const void* a, const void* b, const float* g, double count, void* da, void* db, long n,
                 int dtype, void* stream);
int msml_dropout(const void* x, void* y, long n, float p, long seed, int dtype, void* stream);
/* the same with the seed read from device memory (seed[0]): a dropout captured into a hipGraph draws a new mask on
 * every replay once the host side advances the seed tensor inside the graph */
int msml_dropout_dev(const void* x, void* y, long n, float p, const long* seed, int dtype, void* stream);

/* ---------------------------------------------------------------- convolution -------------
 * Implicit-GEMM convolution on MFMA.  Replaces nn.Conv2d / nn.ConvTranspose2d / nn.Linear
 * forward and backward-data at: backbones/frb/iresnet.py:56-67,209,232 (IBasicBlock, stem,
 * fc), backbones/fm/fmoperator.py:40-48,285-286 (bottleneck, same_conv on cat(yf, yo)),
 * backbones/osb/unet.py:32-38,193-221 (encoder, GCM 7x1/1x7, deconvs on cat(seg, gcm)),
 * headers/partial_fc.py:98 (logits GEMM) and the autograd of all of them.
 *
 *   out[n,oy,ox,ko] = bias[ko] + sum_seg sum_{r,s,c} in_seg[n,iy,ix,c] * wp[ko][seg,r,s,c]
 *   transposed == 0:  iy = oy*stride - pad_h + r                    (conv fwd, deconv bwd-data)
 *   transposed == 1:  iy = (oy + pad_h - r)/stride when divisible   (conv bwd-data, deconv fwd)
 *
 * in0/in1: NHWC [N][H][W][c0p]/[c1p] (in1 may be NULL with c1p = 0); wp from
 * msml_pack_weight with kop rows (>= coutp rounded up to msml_conv_tile_n(coutp));
 * out: NHWC [N][P][Q][coutp]; bias: [coutp] f32 or NULL.
 * stats (or NULL): [ceil(N*P*Q / msml_conv_tile_m(coutp))][2][coutp] f32 partial per-channel
 * (sum, sum of squares) of the f32 results, one row pair per pixel tile, for training-mode
 * BatchNorm (finalised by msml_bn_finalize).  in_dtype/out_dtype: (F32,F32), (BF16,BF16),
 * (BF16,F32). */
int msml_conv_tile_m(int coutp);
int msml_conv_tile_n(int coutp);
int msml_conv2d(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                const float* bias, void* out, int coutp, float* stats, int N, int H, int W,
                int P, int Q, int R, int S, int stride, int pad_h, int pad_w, int transposed,
                int in_dtype, int out_dtype, void* stream);

/* msml_conv2d whose BatchNorm statistics go to an ACCUMULATOR instead of partial rows: `acc` is a zero-initialised
 * double[8][2][coutp]; every workgroup adds its per-channel (sum, sumsq) of the stored output with f64 atomics into
 * row (workgroup index % 8).  f64 sums of f32 partials are exact unless two partials differ by more than 2^29, so the
 * totals do not depend on the order of the adds.  Consumed by msml_bn_fin_act_fwd (no finalize launch in between:
 * the nn.BatchNorm2d that follows a conv in IBasicBlock, backbones/frb/iresnet.py:56-67). */
int msml_conv2d_acc(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                    const float* bias, void* out, int coutp, double* acc, int N, int H, int W,
                    int P, int Q, int R, int S, int stride, int pad_h, int pad_w, int transposed,
                    int in_dtype, int out_dtype, void* stream);

/* Weight gradient of the same family of layers (autograd of the call sites above, and
 * headers/partial_fc.py:169 sub_weight.grad):
 *   dw[a][boff+b][r][s] (+)= sum_{n,py,px} u[n,py,px,a] * v[n, py*stride-pad_h+r, px*stride-pad_w+s, b]
 * u: NHWC [N][P][Q][up] on the natural grid; v: NHWC [N][H][W][vp] shifted operand.
 * Conv2d: u = dY, v = X -> dw = [Cout][Cin][R][S]; ConvTranspose2d: u = X, v = dY ->
 * dw = [Cin][Cout][R][S].  dw is f32 in the parameter's own layout with Btot columns; only
 * a < A and b < Breal are written (channel-concat inputs call once per segment with boff).
 * Deterministic split-K: partial slabs in `workspace` (msml_conv_wgrad_workspace bytes),
 * summed in a fixed order.  N*P*Q must be < 2^24. */
long msml_conv_wgrad_workspace(int up, int vp, int N, int P, int Q, int R, int S);
int msml_conv_wgrad(const void* u, int up, const void* v, int vp, float* dw, int A, int Breal,
                    int Btot, int boff, int N, int H, int W, int P, int Q, int R, int S,
                    int stride, int pad_h, int pad_w, int accumulate, void* workspace,
                    long ws_bytes, int dtype, void* stream);

/* Weight gradients of `group` same-shape 3x3 / stride-1 / pad-1 conv layers in ONE launch pair (backbones/frb/
 * iresnet.py:40-67: consecutive IBasicBlocks of a stage share every dimension).  u / v / dw are HOST arrays of `group`
 * device pointers (dY, X, dW of each layer); everything else as msml_conv_wgrad.  The split-K slab traffic per layer
 * falls by the factor `group`.  MSML_ERR_UNSUPPORTED when the strip / halo kernel does not cover the shape.
 * msml_conv_wgrad_group_max: the largest useful group for a shape (1 = call msml_conv_wgrad instead). */
int msml_conv_wgrad_group_max(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q, int R, int S,
                              int stride, int pad_h, int pad_w);
int msml_conv_wgrad_group(const void* const* u, const void* const* v, float* const* dw, int group, int up, int vp,
                          int A, int Breal, int Btot, int boff, int N, int H, int W, int P, int Q, int R, int S,
                          int stride, int pad_h, int pad_w, int accumulate, void* workspace, long ws_bytes, int dtype,
                          void* stream);

/* 1 when msml_conv_wgrad (bf16) runs this shape on the narrow-operand kernel (wgrad_n32.hip: both operands
 * 32 stored channels, 4x4 / stride-2 transposed convs of the OSB decoder or 3x3 / stride-1). */
int msml_conv_wgrad_kernel_is_n32(int up, int vp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                                  int pad_h, int pad_w);

/* ---------------------------------------------------------------- BatchNorm / PReLU --------
 * nn.BatchNorm2d(eps=1e-5, momentum=0.1) + nn.PReLU + residual of IBasicBlock
 * (backbones/frb/iresnet.py:56-67, backbones/osb/unet.py:80-91), resblock_bottle
 * (backbones/fm/fmoperator.py:53-68), the stems (iresnet.py:209-211, unet.py:193-195), bn2 and
 * BatchNorm1d `features` (iresnet.py:225,233).  Tensors are [M pixels][C], C % 8 == 0.
 *
 * msml_bn_stats: per-channel partial (sum, sumsq) rows -> partial[msml_bn_stats_rows(M,C)][2][C]
 * (the conv kernel's `stats` output has the same row format).
 * msml_bn_finalize: rows > 0 (training): mean / biased var from the partial rows (f64, fixed
 *   order), running stats updated in place like torch (unbiased var, momentum), mean/invstd saved;
 *   rows == 0 (eval): coefficients from the running stats.  Outputs scale = gamma*invstd,
 *   shift = beta - mean*scale.  gamma/beta NULL mean 1/0.
 * msml_bn_act_fwd: res_first == 0: y = prelu(x*scale[c] + shift[c], alpha[c]) + residual
 *   (IBasicBlock); res_first == 1: y = prelu(x*scale[c] + shift[c] + residual, alpha[c])
 *   (resblock_bottle, fmoperator.py:65-67).  alpha / residual optional.
 * msml_bn_act_bwd (training statistics): dx, dgamma, dbeta, dalpha from dy and the saved x;
 *   residual_first (the saved residual, only for res_first == 1) and dres (gradient flowing
 *   to that residual) optional; accumulate != 0 adds the parameter gradients into dgamma / dbeta /
 *   dalpha (the flat gradient arena) instead of overwriting; workspace >= rows*3*C + 2*C floats. */
int msml_bn_stats_rows(long M, int C);
int msml_bn_stats(const void* x, long M, int C, float* partial, int dtype, void* stream);
int msml_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                     void* stream);
int msml_bn_act_fwd(const void* x, const float* scale, const float* shift, const float* alpha,
                    const void* residual, int res_first, void* y, long M, int C, int dtype,
                    void* stream);
/* msml_bn_act_fwd that also emits (sum, sumsq) partial rows of its OUTPUT (as stored), in the row
 * format of msml_bn_stats: the statistics of the next IBasicBlock's leading BatchNorm
 * (iresnet.py:57 bn1 on the previous block's output) without re-reading the tensor. */
int msml_bn_act_fwd_stats_rows(long M, int C);
int msml_bn_act_fwd_stats(const void* x, const float* scale, const float* shift, const float* alpha,
                          const void* residual, int res_first, void* y, long M, int C,
                          float* stats, int dtype, void* stream);
/* msml_bn_finalize (training mode) + msml_bn_act_fwd[_stats] in ONE launch from an accumulator (msml_conv2d_acc):
 * every workgroup folds acc[8][2][C] and derives the coefficients itself, workgroup 0 writes scale / shift /
 * save_mean / save_invstd and updates the running statistics; acc_out (optional, zero-initialised
 * double[8][2][C]) receives the (sum, sumsq) of the stored output.  C / 8 must divide 256. */
int msml_bn_fin_act_fwd(const double* acc, double count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* scale,
                        float* shift, float* save_mean, float* save_invstd, const void* x,
                        const float* alpha, const void* residual, int res_first, void* y, long M, int C,
                        double* acc_out, int dtype, void* stream);
int msml_bn_stats_acc(const void* x, long M, int C, double* acc, int dtype, void* stream);
/* Backward in accumulator mode (the three sums sum g, sum g*xhat, sum dy*min(z,0) as double[8][3][C], f64 atomics of
 * the producer): msml_conv2d_bnbwd_acc is msml_conv2d_bnbwd with that output; msml_bn_fin_bwd_apply is
 * k_bn_bwd_finalize + msml_bn_act_bwd_apply[_next][_s2] in one launch (add_h > 0: compact stride-2 `add`; next_acc:
 * zero-initialised accumulator of the activation-free BatchNorm whose output gradient dx is); msml_bn_act_bwd_acc is
 * msml_bn_act_bwd with its reduce pass adding into `acc`.  C / 8 must divide 256. */
int msml_conv2d_bnbwd_acc(const void* in0, int c0p, const void* wp, int kop, void* out, int coutp, int N,
                          int H, int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                          int transposed, const void* bn_x, const float* bn_scale, const float* bn_shift,
                          const float* bn_alpha, const float* bn_mean, const float* bn_invstd, double* acc,
                          void* stream);
int msml_bn_fin_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift,
                          const float* alpha, const float* save_mean, const float* save_invstd,
                          const double* acc, const void* residual_first, const void* add, int add_h,
                          int add_w, void* dx, void* dres, float* dgamma, float* dbeta, float* dalpha,
                          int accumulate, long M, int C, const void* next_x, const float* next_mean,
                          const float* next_invstd, double* next_acc, int dtype, void* stream);
/* msml_bn_fin_bwd_apply whose NEXT BatchNorm is followed by a PReLU (round 6; the stems, iresnet.py:209-211 / unet.py:193-195:
 * conv -> bn -> prelu -> first IBasicBlock's bn1): dx is the gradient of PReLU(next_x * next_scale + next_shift) and next_acc
 * receives that BatchNorm's three sums (sum g', sum g' * xhat, sum dx * min(z, 0); g' = dx through the PReLU mask), so the
 * stem's own backward is an apply pass only (msml_bn_fin_bwd_apply with next_acc as its `acc`). */
int msml_bn_fin_bwd_apply_next_act(const void* dy, const void* x, const float* scale, const float* shift,
                                   const float* alpha, const float* save_mean, const float* save_invstd,
                                   const double* acc, const void* residual_first, const void* add, int add_h,
                                   int add_w, void* dx, void* dres, float* dgamma, float* dbeta, float* dalpha,
                                   int accumulate, long M, int C, const void* next_x, const float* next_scale,
                                   const float* next_shift, const float* next_alpha, const float* next_mean,
                                   const float* next_invstd, double* next_acc, int dtype, void* stream);
int msml_bn_act_bwd_acc(const void* dy, const void* x, const float* scale, const float* shift,
                        const float* alpha, const float* save_mean, const float* save_invstd,
                        const void* residual_first, const void* add, void* dx, void* dres, float* dgamma,
                        float* dbeta, float* dalpha, int accumulate, long M, int C, double* acc, int dtype,
                        void* stream);
int msml_bn_act_bwd(const void* dy, const void* x, const float* scale, const float* shift,
                    const float* alpha, const float* save_mean, const float* save_invstd,
                    const void* residual_first, void* dx, void* dres, float* dgamma, float* dbeta,
                    float* dalpha, int accumulate, long M, int C, float* workspace, long ws_floats,
                    int dtype, void* stream);
/* db[c] = sum over pixels of dy (biased GCM convs, backbones/osb/unet.py:23-30);
 * workspace >= msml_bn_stats_rows(M,Cp)*2*Cp floats. */
int msml_bias_grad(const void* dy, long M, int Cp, int Creal, float* db, int accumulate,
                   float* workspace, long ws_floats, int dtype, void* stream);
/* out = a + b (GCM branch sum unet.py:37; gradient joins) */
int msml_add(const void* a, const void* b, void* out, long n, int dtype, void* stream);

/* ---------------------------------------------------------------- OSB tail + seg loss -------
 * msml_dap_fwd: DAP = PixelShuffle(3)->AvgPool2d(3) (backbones/osb/unet.py:158-161,223) == mean
 *   over 9-channel groups; x NHWC [N][H][W][Cp>=18] -> seg NCHW f32 [N][2][H][W] and (optional)
 *   mask[N][H][W] u8 = argmax over the 2 classes, ties -> 0 (train.py:357). */
int msml_dap_fwd(const void* x, float* seg, unsigned char* mask, int N, int H, int W, int Cp,
                 int dtype, void* stream);
int msml_dap_bwd(const float* dseg, void* dx, int N, int H, int W, int Cp, int dtype, void* stream);
/* StructureConsensuLossFunction(alpha, beta, 'idx', 'idx')(logit, msk, msk)
 * (tricks/consensus_loss.py:65-167, train.py:258).  logit NCHW f32 [N][2][H][W], msk int64
 * [N][H][W] in {0,1}; loss[1]; dlogit (optional) = d loss / d logit.  workspace >= 20*N floats. */
int msml_seg_consensus_loss(const float* logit, const long* msk, int N, int H, int W, float alpha,
                            float beta, float* loss, float* dlogit, float* workspace,
                            long ws_floats, void* stream);
/* the same with the reference's other reductions (tricks/consensus_loss.py:42-57,127-133,159-162): reduce_pixel_all != 0
 * divides the blob mean by H * W ('all') instead of the blob's pixel count ('idx'); reduce_pixel_kl_all != 0 averages the
 * consensus term over N * H * W instead of its non-zero entries. */
int msml_seg_consensus_loss_r(const float* logit, const long* msk, int N, int H, int W, float alpha, float beta,
                              int reduce_pixel_all, int reduce_pixel_kl_all, float* loss, float* dlogit,
                              float* workspace, long ws_floats, void* stream);

/* ---------------------------------------------------------------- classification head -------
 * kind: 0 = AMArcFace (headers/margin_losses.py:356-418), 1 = AMCosFace (:241-305).
 * msml_rownorm_fwd: F.normalize rows of w[R][E] (margin_losses.py:371, partial_fc.py:115) into
 *   dst[Rp][ld] (storage dtype, zero padded) + inv_norm[R].
 * msml_rownorm_bwd: dw (+)= (dy - y<y,dy>) * inv_norm  with dy f32 [R][ldy].
 * msml_gather_target / msml_margin_fwd / msml_margin_bwd: margin on the target logits, in place
 *   on the f32 cosine matrix [N][ld]; backward emits dcos in storage dtype [N][ldo].
 * msml_pfc_rowstats / msml_pfc_grad: the two local passes of PartialFC's distributed
 *   softmax-CE (partial_fc.py:132-167) around the max / sum / loss all-reduces; the margin is
 *   applied on the fly so the logits are never materialised. */
int msml_rownorm_fwd(const float* w, int R, int Rp, int E, void* dst, int ld, float* inv_norm,
                     int dtype, void* stream);
int msml_rownorm_bwd(const float* w, const float* inv_norm, const float* dy, int ldy, int R, int E,
                     float* dw, int accumulate, void* stream);
int msml_gather_target(const float* cosm, int ld, const long* label, int N, float* out,
                       void* stream);
int msml_margin_fwd(float* cosm, const long* label, int N, int C, int ld, int kind, float s,
                    float m, float a, float k, void* stream);
int msml_margin_bwd(const float* dlogit, int ldg, const long* label, const float* cos_t, int N,
                    int C, void* dcos, int ldo, int kind, float s, float m, float a, float k,
                    int dtype, void* stream);
int msml_pfc_rowstats(const float* cosm, int ld, int N, int C, const long* label, int kind, float s,
                      float m, float a, float k, float* rowmax, float* rowsum, void* stream);
int msml_pfc_grad(const float* cosm, int ld, int N, int C, const long* label, int kind, float s,
                  float m, float a, float k, const float* gmax, const float* gsum, float eps_ls,
                  float inv_n, void* dcos, int ldo, float* ptarget, int dtype, void* stream);
/* torch.optim.SGD(momentum, weight_decay) step on one flat, 16-B aligned f32 buffer
 * (train.py:179-191); clip_coef (device scalar or NULL) multiplies the gradient first.
 * msml_grad_norm_clip: out2[0] = global L2 norm of grad[n], out2[1] = min(1, max_norm /
 * (norm + 1e-6)) = torch.nn.utils.clip_grad_norm_'s factor (train.py:270,275), on device, no
 * host sync; workspace >= 1024 floats. */
int msml_sgd_momentum(float* w, const float* grad, float* mom, long n, float lr, float mu, float wd,
                      int first_step, const float* clip_coef, void* stream);
int msml_grad_norm_clip(const float* grad, long n, float max_norm, float* out2, float* workspace,
                        long ws_floats, void* stream);
/* msml_sgd_momentum_dev: msml_sgd_momentum with the learning rate read from DEVICE memory (lr[0]) and the momentum
 *   buffer always combined (mu * buf + g; a zero buffer gives torch's first step): a step captured into a hipGraph
 *   follows torch.optim.lr_scheduler.LambdaLR (train.py:193-196) without a re-capture.
 * msml_grad_norm_clip_scaled: the gradient buffer holds the SUM over W data-parallel ranks (scale = 1 / W): out2[0] =
 *   norm of the averaged gradient, out2[1] = scale * clip factor -- DistributedDataParallel's division by W
 *   (train.py:136-138) folded into the one coefficient the SGD kernel applies, no separate pass over the gradients. */
int msml_sgd_momentum_dev(float* w, const float* grad, float* mom, long n, const float* lr, float mu, float wd,
                          const float* coef, void* stream);
int msml_grad_norm_clip_scaled(const float* grad, long n, float max_norm, float scale, float* out2, float* workspace,
                               long ws_floats, void* stream);

/* Skinny GEMM with a huge K, split over K with deterministic slab reduction: out[M][coutp] (f32)
 * = a[M][K] (bf16, K % 32 == 0) . wp[kop][K]^T.  PartialFC dX = dcos . Wn (K = local classes,
 * headers/partial_fc.py:169 total_features.grad). */
long msml_gemm_splitk_workspace(int M, int coutp, int K);
int msml_gemm_splitk(const void* a, int M, int K, const void* wp, int kop, float* out, int coutp,
                     void* workspace, long ws_bytes, int dtype, void* stream);

/* Pack many weights (or sub-blocks of weights) in ONE launch.  table: device array of count x 16
 * int64 {w, dst, Afull, Bfull, a_off, A, b_off, B, R, S, transpose, C1, C1p, C2, C2p, KOp}; every
 * entry is packed as msml_pack_weight would pack the sub-block (a_off, A) x (b_off, B) of
 * w[Afull][Bfull][R][S].  Used once per training step for the forward and backward-data operands
 * of every conv / deconv / linear of the model. */
int msml_pack_weights_batched(const long* table, int count, int dtype, void* stream);
/* Same packing through LDS tiles (coalesced on both sides; the per-step refresh).  Only the real
 * elements are written: dst must carry its zero padding already (R*S <= 49).  tile_prefix: device
 * array [count + 1] = the exclusive prefix sum of msml_pack_tiles(A, B, R, S) over the entries
 * followed by the total; optionally [count] = -1 followed by total_tiles entry indices (tile -> entry),
 * which saves the per-block binary search. */
int msml_pack_tiles(int A, int B, int R, int S);
int msml_pack_weights_tiled(const long* table, const int* tile_prefix, int count, int total_tiles,
                            int dtype, void* stream);

/* dst[c][r] = src[r][c] (storage dtype), dst rows ld_d long, zero-filled for r >= R.  Wn^T for the
 * PartialFC dX GEMM (headers/partial_fc.py:169). */
int msml_transpose(const void* src, int R, int C, int ld_s, void* dst, int ld_d, int dtype,
                   void* stream);

/* Inference conv with the eval-mode BatchNorm folded into the epilogue (bf16 tensors only):
 *   res_first == 0: out = prelu(acc*scale[c] + shift[c], alpha[c]) + residual   (IBasicBlock)
 *   res_first == 1: out = prelu(acc*scale[c] + shift[c] + residual, alpha[c])   (resblock_bottle)
 * scale/shift from msml_bn_finalize(rows = 0); alpha, residual optional.  Replaces the
 * conv -> BatchNorm(eval) -> PReLU -> (+identity) chains of backbones/frb/iresnet.py:59-66,
 * backbones/fm/fmoperator.py:55-67 in eval mode (eval/verification.py:271). */
int msml_conv2d_fused(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                      const float* scale, const float* shift, const float* alpha,
                      const void* residual, int res_first, void* out, int coutp, int N, int H,
                      int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                      int transposed, void* stream);

/* ---------------------------------------------------------------- split-bf16 inference ("bf16x3") ---
 * f32-class accuracy on the bf16 MFMA (gfx950 has no TF32; exact-f32 MFMA is 1/16 of the bf16 rate):
 * a value is stored as hi = bf16(x), lo = bf16(x - hi); a pixel of an MSML_BF16X3 tensor holds 3*C bf16
 * channels [hi(C) | lo(C) | hi(C)], a packed weight is laid out [wh | wh | wl] along each tap's channels,
 * so x.w ~= xh.wh + xl.wh + xh.wl is ONE implicit GEMM over 3*C input channels on the unchanged bf16 main
 * loop.  Meets the reference's eval tolerances with fp16=True (embedding <= 1e-3, mask indices bit-exact).
 *
 * msml_conv2d_x3: msml_conv2d_fused on split tensors.  c0p / c1p / coutp are the LOGICAL padded channel
 *   counts (multiples of 32); in0 / in1 / residual / out hold 3x that many bf16 channels per pixel; wp from
 *   msml_pack_weight over the expanded weight ([wh|wh|wl] per segment, C1p = 3*c0p, C2p = 3*c1p).
 *   out = [prelu](acc*scale + shift [+ residual]) [+ residual]; scale NULL = 1, shift NULL = 0 (plain conv
 *   with bias: shift = bias).  Replaces the same call sites as msml_conv2d / msml_conv2d_fused in eval mode.
 * msml_x3_bn_act_fwd / msml_x3_fm_fuse_fwd / msml_x3_add: msml_bn_act_fwd / msml_fm_fuse_fwd / msml_add on
 *   split tensors of M pixels x C channels.  msml_x3_from_f32 / msml_x3_to_f32: [M][C] f32 <-> split. */
int msml_conv2d_x3(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                   const float* scale, const float* shift, const float* alpha, const void* residual,
                   int res_first, void* out, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                   int stride, int pad_h, int pad_w, int transposed, void* stream);
/* msml_conv2d_x3_border: the 3x3 / stride-1 / pad-1 forward case of msml_conv2d_x3 with a shift per BORDER CLASS of the
 *   output pixel: shift9 = float[9][coutp], row cy * 3 + cx, cy (cx) = 0 on the first row (column) of the map, 2 on the
 *   last, 1 inside; H, W >= 2.  Lets the caller fold an eval-mode BatchNorm IN FRONT of the conv into it -- bn1 -> conv1
 *   of IBasicBlock (reference backbones/frb/iresnet.py:58-60): conv(W, s x + t) = conv(W s, x) + the sum of W t over
 *   the taps that fall INSIDE the map (the zero padding follows the BatchNorm), which is a constant per class. */
int msml_conv2d_x3_border(const void* in0, int c0p, const void* wp, int kop, const float* scale,
                          const float* shift9, const float* alpha, const void* residual, int res_first,
                          void* out, int coutp, int N, int H, int W, void* stream);
int msml_x3_bn_act_fwd(const void* x, const float* scale, const float* shift, const float* alpha,
                       const void* residual, int res_first, void* y, long M, int C, void* stream);
int msml_x3_fm_fuse_fwd(const void* x, const void* yf, void* z, long M, int C, int act, int arith,
                        void* stream);
int msml_x3_add(const void* a, const void* b, void* out, long M, int C, void* stream);
int msml_x3_from_f32(const float* src, void* dst, long M, int C, void* stream);
int msml_x3_to_f32(const void* src, float* dst, long M, int C, void* stream);

/* ---------------------------------------------------------------- verification metrics -----------
 * Device side of eval/verification.py:54-199 (calculate_roc / calculate_val: 10-fold accuracy over a
 * threshold grid, TAR @ FAR).  msml_pair_sqdist: emb [2*n_pairs][E] f32, rows (2i, 2i+1) = pair i ->
 * dist[i] = || a/|a| - b/|b| ||^2 in f64 (:298-301,78-79).  msml_pair_hist: hist[nfolds][2][nthr + 1]
 * int32, hist[f][s][k] = number of pairs of KFold test fold f (shuffle=False) with issame == s whose
 * first threshold index with dist < thr[k] is k (k == nthr: none); every tp / fp / tn / fn of every
 * threshold and fold is a prefix sum of it.  thr: ascending f64 grid (np.arange(0, 4, 0.01 | 0.001)). */
int msml_pair_sqdist(const float* emb, int n_pairs, int E, double* dist, void* stream);
/* the same on the f64 sum of the orig + flip embedding passes (verification.py:283,299-300 accumulates both passes
 * in float64 arrays before normalising) */
int msml_pair_sqdist_f64(const double* emb, int n_pairs, int E, double* dist, void* stream);
int msml_pair_hist(const double* dist, const unsigned char* same, int n_pairs, const double* thr, int nthr,
                   int nfolds, int* hist, void* stream);

/* ---------------------------------------------------------------- device input pipeline ----------
 * Replaces the per-sample CPU augmentation of FaceByRandOccMask.__getitem__ (datasets/load_dataset.py:
 * 101-139) for the occluders that need no dataset assets: RandomRect (datasets/augment/rand_occ.py:103-139),
 * RandomEllipse (:148-203, analytic ellipse), RandomConnectedPolygon (:217-322, even-odd point-in-polygon test),
 * NoneOcc (:80-90), RandomBlock (:43-72, evaluation), the random
 * horizontal flip (load_dataset.py:119-123), the Gaussian light of _add_gauss_to_face (:183-201, _get_gauss
 * :282-339) and ToTensor + Normalize(0.5, 0.5).
 * msml_occ_draw: per-image descriptors desc[N][64] int32 {kind (0 none, 1 rect, 2 ellipse, 3 block, 4 polygon),
 *   x0|cx, y0|cy, w|aw, h|ah, r, g, b, flip, light cx / cy / scale (f32 bits), polygon vertex count, 0, 0, 0,
 *   up to 24 polygon vertices (x, y)} from a counter-based generator keyed by (seed, offset + image index):
 *   mode 0 = training mix {rect, ellipse, polygon, none}, 1 = rect (lo..hi percent), 2 = black block (lo..hi
 *   percent), 3 = none, 4 = polygon.
 * msml_occ_apply: src [N][H][W][3] uint8 (decoded RGB) -> img [N][3][H][W] f32 in [-1, 1] (occluded,
 *   flipped, lit when light != 0, normalised), msk [N][H][W] int64 (0 occluded / 1 clean,
 *   load_dataset.py:37), ori [N][3][H][W] (clean, flipped, normalised; optional). */
int msml_occ_draw(long seed, long offset, int N, int H, int W, int mode, int lo, int hi, int flip, int* desc,
                  void* stream);
int msml_occ_apply(const unsigned char* src, const int* desc, float* img, long* msk, float* ori, int N, int H,
                   int W, int light, void* stream);

/* Texture occluders of the reference's training mix: RandomGlasses / RandomGlassesList (datasets/augment/rand_occ.py:
 * 337-428), RandomScarf (:431-517), RandomRealObject (:520-600), selected as datasets/load_dataset.py:72-85,155-163.
 * The RGBA images come from the CALLER (atlas: every set's entries [num][h0][w0][4] uint8, preloaded from the user's
 * checkout exactly as the reference's constructors do; nothing ships with this library).
 *   meta[nsets][16] int32: byte offset of the set in `atlas`, entries, h0, w0, kind (5 glasses, 6 scarf, 7 object),
 *     wmin, wmax, hmin, hmax (resampled sizes the tables cover), wdir, hdir, 0...;
 *   dir / rtab: PIL's bicubic resampling coefficients (ImagingResample precompute_coeffs + normalize_coeffs_8bpc,
 *     22-bit fixed point), built on the host: rtab[dir[wdir + w' - wmin] + x * 10 + {0: first tap, 1: taps, 2..9: k}].
 * msml_occ_draw_tex: msml_occ_draw plus modes 5 (ms1m mix: uniform over rect, ellipse, polygon, glasses, scarf, object,
 *   none), 6 (casia mix: none with probability 1/2, else one of the six), 7 / 8 / 9 (glasses / scarf / object only);
 *   texture descriptors: kind, x, y of the paste, resampled w', h', set (word 13), entry (word 14).
 * msml_occ_resize: per image of a texture kind, Image.resize((w', h')) of its RGBA entry, bit for bit (RGBA -> RGBa,
 *   horizontal + vertical fixed-point pass, RGBa -> RGBA) into patch[n][patch_stride] (rows of w' RGBA pixels);
 *   lds_bytes >= (h0 * w0 + h0 * wmax) * 4 of the largest set.
 * msml_occ_apply_tex: msml_occ_apply with the paste: glasses replace the face where alpha > 10, scarf / object where
 *   alpha != 0, the mask is 0 where alpha != 0 (all three), cropped at the image border. */
int msml_occ_draw_tex(long seed, long offset, int N, int H, int W, int mode, int lo, int hi, int flip,
                      const int* meta, int nsets, int* desc, void* stream);
int msml_occ_resize(const unsigned char* atlas, const int* meta, const int* dir, const int* rtab, const int* desc,
                    unsigned char* patch, long patch_stride, int N, int lds_bytes, void* stream);
int msml_occ_apply_tex(const unsigned char* src, const int* desc, const unsigned char* patch, long patch_stride,
                       float* img, long* msk, float* ori, int N, int H, int W, int light, void* stream);

/* Backward-data of the OSB decoder's ConvTranspose2d(36 -> 18, k 4, s 2, p 1) on cat(seg, gcm) (backbones/osb/unet.py:
 * 140-156, autograd of deconv2..5) for BOTH input segments from one pass over dY (bf16):
 *   dxS[n, i, j, ci] = sum_{r, s, co} dy[n, 2i - 1 + r, 2j - 1 + s, co] * w[S * 18 + ci][co][r][s]
 * dy [N][2H][2H][32], dx0 / dx1 [N][H][H][32]; wp0 / wp1 = msml_pack_weight of the segment's rows of the
 * ConvTranspose2d weight (transpose 0, C1 = 18: [32 ci rows][16 taps x 32 co]).  H in {14, 28, 56};
 * MSML_ERR_UNSUPPORTED otherwise (callers then run msml_conv2d per segment). */
int msml_deconv4_bwd_data(const void* dy, const void* wp0, const void* wp1, void* dx0, void* dx1, int N, int H,
                          void* stream);

/* Backward-data conv fused with the backward REDUCE of the BatchNorm(+PReLU) that produced the
 * conv's input in the forward (IBasicBlock: bn1 -> conv1, bn2 -> prelu -> conv2,
 * backbones/frb/iresnet.py:59-64): the conv output dX is that BatchNorm's dy, so the epilogue
 * accumulates sum g, sum g*xhat, sum dy*min(z,0) per channel from the stored (bf16) dX and the
 * saved BatchNorm input bn_x (same [N][P][Q][coutp] layout), one partial row [3][coutp] per
 * workgroup.  bf16 only; MSML_ERR_UNSUPPORTED when the shape is not on the fast path (callers
 * fall back to msml_conv2d + msml_bn_act_bwd).  partial needs msml_conv2d_bnbwd_rows() rows;
 * *rows_used = rows written (all of them fully).  msml_bn_act_bwd_apply finishes the job:
 * finalize over `rows` + dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) [+ add];
 * coef_ws: 98*C floats of scratch (2*C coefficients + 32 folded partial rows). */
int msml_conv2d_bnbwd_rows(int coutp, int N, int P, int Q);
int msml_conv2d_bnbwd(const void* in0, int c0p, const void* wp, int kop, void* out, int coutp,
                      int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                      int pad_w, int transposed, const void* bn_x, const float* bn_scale,
                      const float* bn_shift, const float* bn_alpha, const float* bn_mean,
                      const float* bn_invstd, float* partial, int rows_cap, int* rows_used,
                      void* stream);
int msml_bn_act_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift,
                          const float* alpha, const float* save_mean, const float* save_invstd,
                          const float* partial, int rows, const void* add, void* dx,
                          float* dgamma, float* dbeta, float* dalpha, int accumulate, long M,
                          int C, float* coef_ws, int dtype, void* stream);

/* msml_bn_act_bwd_apply / _next with `add` given as the COMPACT input gradient of a 1x1 / stride-2
 * / pad-0 conv over an H x W map (downsample path of a stage's first block,
 * backbones/frb/iresnet.py:52-54,66): add[N][ceil(H/2)][ceil(W/2)][C] is summed into the pixels
 * with even (y, x); the dense gradient (3/4 zeros) is neither written nor re-read.  M = N*H*W < 2^24. */
int msml_bn_act_bwd_apply_s2(const void* dy, const void* x, const float* scale, const float* shift,
                             const float* alpha, const float* save_mean, const float* save_invstd,
                             const float* partial, int rows, const void* add, int H, int W, void* dx,
                             float* dgamma, float* dbeta, float* dalpha, int accumulate, long M,
                             int C, float* coef_ws, int dtype, void* stream);
int msml_bn_act_bwd_apply_next_s2(const void* dy, const void* x, const float* scale,
                                  const float* shift, const float* alpha, const float* save_mean,
                                  const float* save_invstd, const float* partial, int rows,
                                  const void* add, int H, int W, void* dx, float* dgamma, float* dbeta,
                                  float* dalpha, int accumulate, long M, int C, float* coef_ws,
                                  const void* next_x, const float* next_mean,
                                  const float* next_invstd, float* next_partial, int dtype,
                                  void* stream);

/* Training-mode BatchNorm(+PReLU) in FRONT of a 3x3 / stride-1 / pad-1 conv (IBasicBlock:
 * bn1 -> conv1, bn2 -> prelu -> conv2, backbones/frb/iresnet.py:58-63, backbones/osb/unet.py:82-87)
 * applied while the conv's input image sits in LDS: the conv reads the BatchNorm's INPUT in0 and
 * computes on X = bf16(PReLU(in0 * in_scale + in_shift)) (in_alpha NULL: no PReLU; scale / shift
 * from msml_bn_finalize), zero padding applied to X as in the unfused graph, so the result equals
 * msml_bn_act_fwd followed by msml_conv2d bit for bit while X is never written to HBM.
 * msml_conv_wgrad_bnin is the matching weight gradient (u = dY, v = the BatchNorm input).
 * bf16 only, and only the shapes of the halo-tile kernels (the *_applies queries return 1);
 * MSML_ERR_UNSUPPORTED otherwise -- callers then materialise X with msml_bn_act_fwd. */
int msml_conv2d_bnin_applies(int c0p, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                             int stride, int pad_h, int pad_w, int want_stats);
int msml_conv2d_bnin(const void* in0, int c0p, const float* in_scale, const float* in_shift,
                     const float* in_alpha, const void* wp, int kop, void* out, int coutp,
                     float* stats, int N, int H, int W, int P, int Q, int R, int S, int stride,
                     int pad_h, int pad_w, void* stream);
int msml_conv_wgrad_bnin_applies(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q,
                                 int R, int S, int stride, int pad_h, int pad_w);
int msml_conv_wgrad_bnin(const void* u, int up, const void* v, int vp, const float* x_scale,
                         const float* x_shift, const float* x_alpha, float* dw, int A, int Breal,
                         int Btot, int boff, int N, int H, int W, int P, int Q, int R, int S,
                         int stride, int pad_h, int pad_w, int accumulate, void* workspace,
                         long ws_bytes, void* stream);

/* Name of the kernel the conv entry points launch for a shape (profiling labels only). */
const char* msml_conv2d_kernel(int c0p, int c1p, int coutp, int N, int H, int W, int P, int Q,
                               int R, int S, int stride, int pad_h, int pad_w, int transposed,
                               int in_dtype, int out_dtype, int want_stats);

/* Stem im2col (iresnet.py:209 conv1 / unet.py:193 on the 3-channel image): x NCHW f32 ->
 * out[N][P][Q][KP], k = (r*S + s)*C + c, zeros for k >= R*S*C and outside the image; the stem then
 * runs as a 1x1 conv over KP channels with the weight reordered to [Cout][R][S][C]. */
int msml_stem_im2col(const float* x, void* out, int N, int C, int H, int W, int P, int Q, int R,
                     int S, int stride, int pad, int KP, int dtype, void* stream);

/* msml_bn_act_bwd_apply that also reduces the backward sums of the activation-free BatchNorm whose
 * output gradient the written dx is (chained IBasicBlocks: block i+1's bn1 input gradient is block
 * i's output gradient, i.e. the dy of block i's bn3, iresnet.py:65): next_x = that BatchNorm's
 * saved input, next_partial[msml_bn_act_bwd_apply_rows(M, C)][3][C]. */
int msml_bn_act_bwd_apply_rows(long M, int C);
int msml_bn_act_bwd_apply_next(const void* dy, const void* x, const float* scale, const float* shift,
                               const float* alpha, const float* save_mean, const float* save_invstd,
                               const float* partial, int rows, const void* add, void* dx,
                               float* dgamma, float* dbeta, float* dalpha, int accumulate, long M,
                               int C, float* coef_ws, const void* next_x, const float* next_mean,
                               const float* next_invstd, float* next_partial, int dtype, void* stream);

/* Training-mode BatchNorm (+ PReLU) -> 3x3 / stride-1 / pad-1 conv in ONE launch with accumulator-mode statistics on
 * both sides (bn1 -> conv1 and bn2 -> prelu -> conv2 of IBasicBlock, backbones/frb/iresnet.py:58-62): coefficients from
 * acc_in (double[8][2][c0p], the producer's sums) in the kernel prologue, the normalised input applied in LDS and written
 * to act_out (NHWC like in0; the weight gradient reads it), coef_out = float[4][c0p] (scale, shift, mean, invstd), running
 * statistics updated, the output's sums added to acc_out (zero-initialised double[8][2][coutp]).  Bit-identical to
 * msml_bn_fin_act_fwd + msml_conv2d_acc.  Shapes: msml_conv2d_bnin_acc_applies -- 1: served by the halo-tile conv, 2: by the
 * weights-stationary 64-channel kernel (experiment builds only, msml_has_experiments), 3: by the persistent 128-channel halo
 * tile (round 6: 64 k input channels, 128 output channels, at least two rounds of tiles), 0: not covered. */
int msml_conv2d_bnin_acc_applies(int c0p, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                                 int pad_h, int pad_w);
int msml_conv2d_bnin_acc(const void* in0, int c0p, const double* acc_in, double count, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         float* coef_out, const float* in_alpha, void* act_out, const void* wp, int kop, void* out,
                         int coutp, double* acc_out, int N, int H, int W, int P, int Q, int R, int S, int stride,
                         int pad_h, int pad_w, void* stream);

/* BatchNorm BACKWARD -> 3x3 / stride-1 backward-data conv -> sums of the next BatchNorm backward, ONE launch in
 * accumulator mode (the backward of conv2 / conv1 of IBasicBlock with bn3 / bn2 in front, backbones/frb/iresnet.py:59-65):
 * in0 = dy of the upper BatchNorm, up_x its saved input, up_scale ... up_invstd its saved coefficients, up_acc the three
 * sums double[8][3][c0p] its producer accumulated; dc = the BatchNorm's input gradient is formed in LDS, written to dc_out
 * (NHWC like in0; the weight gradient reads it) and convolved; dgamma / dbeta / dalpha (+)= its parameter gradients; the
 * epilogue adds the sums of the LOWER BatchNorm (bn_*, acc) exactly like msml_conv2d_bnbwd_acc.  Bit-identical to
 * msml_bn_fin_bwd_apply + msml_conv2d_bnbwd_acc.  Shapes: msml_conv2d_bnbwd_in_acc_applies. */
int msml_conv2d_bnbwd_in_acc_applies(int c0p, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                                     int pad_h, int pad_w);
int msml_conv2d_bnbwd_in_acc(const void* in0, int c0p, const void* up_x, const float* up_scale, const float* up_shift,
                             const float* up_alpha, const float* up_mean, const float* up_invstd, const double* up_acc,
                             float* dgamma, float* dbeta, float* dalpha, int accumulate, void* dc_out, const void* wp,
                             int kop, void* out, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                             int pad_h, int pad_w, const void* bn_x, const float* bn_scale, const float* bn_shift,
                             const float* bn_alpha, const float* bn_mean, const float* bn_invstd, double* acc,
                             void* stream);

/* Block-level entry point: every launch of one IBasicBlock forward (backbones/frb/iresnet.py:56-67; the OSB encoder's
 * copy backbones/osb/unet.py:80-91) in the bf16 training path with accumulator-mode statistics, enqueued by ONE call:
 * bn1 -> conv1 -> bn2 + PReLU -> conv2 (stride) [-> downsample conv 1x1 -> its BatchNorm] -> bn3 + identity, i.e. the
 * sequence msml_bn_fin_act_fwd / msml_conv2d_acc the host otherwise issues launch by launch (same kernels, same order,
 * bit-identical results).  ptrs / ints / flts: tables indexed by the enums at the top of csrc/block.hip
 * (msml_iblock_fwd_tables returns their lengths); every pointer is a device pointer borrowed for the enqueue,
 * coefficient blocks are float[4][C] (scale, shift, mean, invstd), accumulators zero-initialised double[8][2][C]. */
int msml_iblock_fwd_tables(int* nptr, int* nint, int* nflt);
int msml_iblock_fwd(const void* const* ptrs, const int* ints, const float* flts, void* stream);

/* Box calibration probe (bench.py `calibration`, not part of the hot path): `wgs` workgroups of four waves run `iters`
 * rounds of 16 register-resident v_mfma_f32_16x16x32_bf16 on the random bf16 operands in seed[4096]; out[wgs * 256]
 * receives the accumulator sums.  2 * 16 * 16 * 32 * 16 * iters FLOP per wave.  The reference has no counterpart: its
 * benchmark prints images/sec only (train.py:303-318, utils/utils_callbacks.py). */
int msml_probe_mfma(const void* seed, float* out, int wgs, int iters, void* stream);
/* The same with every operand re-read from LDS in the mix of the halo-tile conv's main loop: `wgs` workgroups of eight
 * waves (two per SIMD), per round 9 ds_read_b128 fragments feed 14 MFMAs; iters even; out[wgs * 512].
 * 2 * 16 * 16 * 32 * 14 * iters FLOP per wave. */
int msml_probe_mfma_lds(const void* seed, float* out, int wgs, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSML_HIP_H */
