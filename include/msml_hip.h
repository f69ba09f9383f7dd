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
 *     multiple of 8; pad channels hold exact zeros.  `dtype` selects the STORAGE type of
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

enum { MSML_F32 = 0, MSML_BF16 = 1 };

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

#ifdef __cplusplus
}
#endif
#endif /* MSML_HIP_H */
