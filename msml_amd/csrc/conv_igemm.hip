// Implicit-GEMM convolution on MFMA (gfx950), NHWC activations, K-contiguous packed weights.
//
//   out[m][ko] = sum_{seg} sum_{r,s} sum_{c} in_seg[gather(m, r, s)][c] * wp[ko][seg,r,s,c]
//
// GEMM view: M = N*P*Q output pixels (rows), N = packed output channels (cols),
// K = sum over segments of R*S*Cp.  One kernel serves
//   conv forward / deconv backward-data      (normal gather:      iy = oy*stride - pad + r)
//   conv backward-data / deconv forward      (transposed gather:  oy_in = (oy + pad - r)/stride)
//   Linear layers and the PartialFC GEMMs    (1x1 or HxW "valid" windows)
// Two input segments implement channel concatenation (cat(yf, yo), cat(seg, gcm)) without
// materialising it.  Reference call sites: backbones/frb/iresnet.py:56-67 (IBasicBlock convs),
// backbones/fm/fmoperator.py:285-286 (same_conv on the concat), backbones/osb/unet.py:207-221
// (GCM convs, deconvs), backbones/frb/iresnet.py:232 (fc), headers/partial_fc.py:98,169.
//
// Tiling: 256 threads = 4 waves; workgroup tile BM x BN, K-step 32; wave tile made of
// 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 for the exact-f32
// parity mode).  Tiles are staged through LDS (double-buffered, 16-B chunks XOR-swizzled so
// both the ds_write_b128 staging and the ds_read_b128 fragment reads are conflict-free).
// Pixels are the MFMA row dimension, so an accumulator register's 32 lanes hold 32 consecutive
// output channels of one pixel: stores are 64-128 B contiguous per pixel row.
#include <stdlib.h>

#include "common.h"

struct ConvArgs {
  const void* in[2];
  int cp[2];        // padded channels per segment
  int ksteps[2];    // K-steps (of 32) per segment
  int nseg;
  int N, H, W, P, Q;
  int R, S, stride_shift, stride, pad_h, pad_w, transposed;
  const void* wp;
  int Ktot;
  void* out;
  int coutp;        // output channel stride (and valid column bound)
  const float* bias;
  float* stats;     // [tiles_m][2][coutp] partial (sum, sumsq) or null
  int stats_acc;    // accumulator mode (common.h): stats is double[MSML_ACC_ROWS][2][coutp]
  long M;
};

template <typename T>
struct Elem;
template <>
struct Elem<unsigned short> {           // bf16 storage
  static constexpr int CH = 8;          // elements per 16-B chunk
  static constexpr int CPR = 4;         // chunks per 32-element K row
  __device__ static __forceinline__ int swz(int row) { return (row >> 2) & 3; }
};
template <>
struct Elem<float> {
  static constexpr int CH = 4;
  static constexpr int CPR = 8;
  __device__ static __forceinline__ int swz(int row) { return (row >> 1) & 7; }
};

// acc[tm][tn] += A(32 rows x 32 k) * B(32 cols x 32 k)^T for one K-step, fragments from LDS.
// As/Bs: [rows][CPR] chunks of 16 B (u32x4), swizzled.
template <typename T, int TM, int TN>
__device__ __forceinline__ void mma_step(const u32x4* __restrict__ As, const u32x4* __restrict__ Bs,
                                         int arow0, int brow0, int lane, f32x16 (&acc)[TM][TN]) {
  const int r = lane & 31, h = lane >> 5;
  constexpr int CPR = Elem<T>::CPR;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      u32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) {
        int row = arow0 + i * 32 + r;
        a[i] = As[row * CPR + ((kk * 2 + h) ^ Elem<T>::swz(row))];
      }
#pragma unroll
      for (int j = 0; j < TN; j++) {
        int row = brow0 + j * 32 + r;
        b[j] = Bs[row * CPR + ((kk * 2 + h) ^ Elem<T>::swz(row))];
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
  } else {
    // f32: a K window of 8 = two 16-B chunks; lane half h takes chunk h and feeds its 4 floats
    // to 4 consecutive 32x32x2 MFMAs (any K permutation is valid as long as A and B agree).
#pragma unroll
    for (int w = 0; w < 4; w++) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) {
        int row = arow0 + i * 32 + r;
        a[i] = __builtin_bit_cast(f32x4, As[row * CPR + ((w * 2 + h) ^ Elem<T>::swz(row))]);
      }
#pragma unroll
      for (int j = 0; j < TN; j++) {
        int row = brow0 + j * 32 + r;
        b[j] = __builtin_bit_cast(f32x4, Bs[row * CPR + ((w * 2 + h) ^ Elem<T>::swz(row))]);
      }
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
  }
}

template <typename TIN, typename TOUT, int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(256) k_conv_igemm(const ConvArgs p) {
  constexpr int CH = Elem<TIN>::CH, CPR = Elem<TIN>::CPR;
  constexpr int RPT = 256 / CPR;            // rows covered per pass of the 256 threads
  constexpr int NA = BM / RPT, NB = BN >= RPT ? BN / RPT : 1;
  constexpr bool BPART = BN < RPT;            // only the first BN rows of threads stage B
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32;
  static_assert(WGM * WGN == 4 && NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1, "tile config");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* As = reinterpret_cast<u32x4*>(smem);                 // [2][BM][CPR]
  u32x4* Bs = As + 2 * BM * CPR;                              // [2][BN][CPR]
  MSML_LDS_REGION(As, 2 * BM * CPR * 16);
  MSML_LDS_REGION(Bs, 2 * BN * CPR * 16);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cc = t % CPR, row0 = t / CPR;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // per-row gather bases
  int nb[NA], y0[NA], x0[NA];
  bool rok[NA];
  const int PQ = p.P * p.Q;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    long m = m0 + row0 + i * RPT;
    rok[i] = m < p.M;
    int mm = rok[i] ? (int)m : 0;
    int n = mm / PQ;
    int rem = mm - n * PQ;
    int oy = rem / p.Q, ox = rem - oy * p.Q;
    nb[i] = n * p.H * p.W;
    if (p.transposed) {
      y0[i] = oy + p.pad_h;
      x0[i] = ox + p.pad_w;
    } else {
      y0[i] = oy * p.stride - p.pad_h;
      x0[i] = ox * p.stride - p.pad_w;
    }
  }
  const TIN* wrow[NB];
#pragma unroll
  for (int i = 0; i < NB; i++)
    wrow[i] = reinterpret_cast<const TIN*>(p.wp) + (long)(n0 + row0 + i * RPT) * p.Ktot + cc * CH;

  // K iterator state (identical for all threads sharing cc)
  int seg = 0, ks_in_seg = 0, r = 0, s = 0, cidx = cc;
  int cpk = p.cp[0] / CH;
  const TIN* inp = reinterpret_cast<const TIN*>(p.in[0]);
  int cpseg = p.cp[0];
  auto normalize = [&]() {
    while (cidx >= cpk) {
      cidx -= cpk;
      if (++s == p.S) { s = 0; ++r; }
    }
  };
  normalize();
  const int total_steps = p.ksteps[0] + (p.nseg > 1 ? p.ksteps[1] : 0);

  u32x4 ra[NA], rb[NB];
  auto gload = [&](int kstep) {
    const bool tap_ok = r < p.R;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      int iy, ix;
      bool ok = rok[i] && tap_ok;
      if (p.transposed) {
        int ty = y0[i] - r, tx = x0[i] - s;
        ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & (p.stride - 1)) == 0;
        iy = ty >> p.stride_shift;
        ix = tx >> p.stride_shift;
        ok = ok && iy < p.H && ix < p.W;
      } else {
        iy = y0[i] + r;
        ix = x0[i] + s;
        ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      }
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) {
        long off = (long)(nb[i] + iy * p.W + ix) * cpseg + cidx * CH;
        v = *reinterpret_cast<const u32x4*>(inp + off);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; i++)
      if (!BPART || row0 < BN) rb[i] = *reinterpret_cast<const u32x4*>(wrow[i] + (long)kstep * 32);
  };
  auto advance = [&]() {
    if (++ks_in_seg == p.ksteps[seg] && seg + 1 < p.nseg) {
      seg++;
      ks_in_seg = 0;
      r = 0; s = 0; cidx = cc;
      cpseg = p.cp[seg];
      cpk = cpseg / CH;
      inp = reinterpret_cast<const TIN*>(p.in[seg]);
    } else {
      cidx += CPR;
    }
    normalize();
  };
  auto lstore = [&](int buf) {
    u32x4* a = As + buf * BM * CPR;
    u32x4* b = Bs + buf * BN * CPR;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      int row = row0 + i * RPT;
      a[row * CPR + (cc ^ Elem<TIN>::swz(row))] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      int row = row0 + i * RPT;
      if (!BPART || row0 < BN) b[row * CPR + (cc ^ Elem<TIN>::swz(row))] = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int wm = wave / WGN, wn = wave % WGN;
  const int arow0 = wm * WTM, brow0 = wn * WTN;

  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < total_steps; step++) {
    const bool more = step + 1 < total_steps;
    if (more) {
      advance();
      gload(step + 1);
    }
    mma_step<TIN, TM, TN>(As + cur * BM * CPR, Bs + cur * BN * CPR, arow0, brow0, lane, acc);
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---------------- epilogue: bias, store, optional per-channel (sum, sumsq) partials -------
  const int h = lane >> 5, c32 = lane & 31;
  TOUT* outp = reinterpret_cast<TOUT*>(p.out);
  float* red = reinterpret_cast<float*>(smem);     // reuse LDS: [WGM][2][BN]
  MSML_LDS_REGION(red, WGM * 2 * BN * 4);
  if (p.stats) __syncthreads();
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int col = n0 + brow0 + j * 32 + c32;
    const bool cok = col < p.coutp;
    const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int row = arow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        long m = m0 + row;
        float v = acc[i][j][e] + bv;
        if (m < p.M && cok) {
          store1<TOUT>(outp + m * p.coutp + col, v);
          s1 += v;
          s2 += v * v;
        }
      }
    }
    if (p.stats) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        red[(wm * 2 + 0) * BN + brow0 + j * 32 + c32] = s1;
        red[(wm * 2 + 1) * BN + brow0 + j * 32 + c32] = s2;
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    for (int i = t; i < 2 * BN; i += 256) {
      int which = i / BN, c = i % BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; w++) v += red[(w * 2 + which) * BN + c];
      int col = n0 + c;
      if (col < p.coutp) stats_emit(p.stats, p.stats_acc, blockIdx.x, which, p.coutp, col, v);
    }
  }
}

template <typename TIN, typename TOUT, int BM, int BN, int WGM, int WGN>
static int launch(const ConvArgs& a, hipStream_t stream) {
  constexpr int CPR = Elem<TIN>::CPR;
  size_t lds = (size_t)2 * (BM + BN) * CPR * 16;
  dim3 grid(cdiv(a.M, BM), cdiv(a.coutp, BN));
  hipLaunchKernelGGL((k_conv_igemm<TIN, TOUT, BM, BN, WGM, WGN>), grid, dim3(256), lds, stream, a);
  return 0;
}

bool msml_conv_fast_dispatch(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                             const float* bias, void* out, int coutp, float* stats, int N, int H,
                             int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                             int transposed, int in_dtype, int out_dtype, int bn, hipStream_t st,
                             const float* scale, const float* alpha, const void* residual, int res_first,
                             const BnBwdFuse* bnb, int* bnb_rows);

bool msml_conv_line_applies(int c0p, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                            int pad_w);
bool msml_conv_line_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out, int coutp,
                             int N, int H, int W, int R, int S, int transposed, hipStream_t st,
                             const float* scale = nullptr, const void* residual = nullptr);

bool msml_deconv4_applies(int c0p, int c1p, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                          int pad_h, int pad_w, int transposed);
bool msml_deconv4_dispatch(const void* in0, const void* in1, const void* wp, int kop, const float* bias, void* out, int N,
                           int H, hipStream_t st);

thread_local int msml_tl_stats_acc = 0;
thread_local int msml_tl_bias9 = 0;

extern "C" int msml_conv_tile_m(int coutp) { return coutp <= 64 ? 256 : 128; }
extern "C" int msml_conv_tile_n(int coutp) { return coutp <= 32 ? 32 : (coutp <= 64 ? 64 : 128); }

extern "C" int msml_conv2d(const void* in0, int c0p, const void* in1, int c1p, const void* wp,
                           int kop, const float* bias, void* out, int coutp, float* stats,
                           int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                           int pad_w, int transposed, int in_dtype, int out_dtype, void* stream) {
  MSML_CHECK(in0 && wp && out, MSML_ERR_SHAPE, "conv2d: null pointer");
  MSML_CHECK(N > 0 && H > 0 && W > 0 && P > 0 && Q > 0 && R > 0 && S > 0, MSML_ERR_SHAPE,
             "conv2d: bad dims N=%d H=%d W=%d P=%d Q=%d R=%d S=%d", N, H, W, P, Q, R, S);
  MSML_CHECK(c0p > 0 && c0p % 8 == 0 && c1p >= 0 && c1p % 8 == 0 && coutp > 0 && coutp % 8 == 0,
             MSML_ERR_SHAPE, "conv2d: channel counts must be multiples of 8 (c0p=%d c1p=%d coutp=%d)",
             c0p, c1p, coutp);
  MSML_CHECK(stride == 1 || stride == 2 || stride == 4, MSML_ERR_UNSUPPORTED,
             "conv2d: stride %d (power of two <= 4 only)", stride);
  MSML_CHECK((c1p == 0) == (in1 == nullptr), MSML_ERR_SHAPE, "conv2d: in1/c1p mismatch");
  if (!transposed) {
    MSML_CHECK((H + 2 * pad_h - R) / stride + 1 == P && (W + 2 * pad_w - S) / stride + 1 == Q,
               MSML_ERR_SHAPE, "conv2d: P,Q inconsistent with H,W,R,S,stride,pad");
  } else {
    MSML_CHECK((H - 1) * stride - 2 * pad_h + R <= P + stride - 1 && (H - 1) * stride - 2 * pad_h + R >= P - (stride - 1) &&
                   (W - 1) * stride - 2 * pad_w + S <= Q + stride - 1 && (W - 1) * stride - 2 * pad_w + S >= Q - (stride - 1),
               MSML_ERR_SHAPE, "conv2d(transposed): P,Q inconsistent with H,W,R,S,stride,pad");
  }
  ConvArgs a;
  a.in[0] = in0; a.in[1] = in1;
  a.cp[0] = c0p; a.cp[1] = c1p;
  a.nseg = in1 ? 2 : 1;
  a.ksteps[0] = (R * S * c0p + 31) / 32;
  a.ksteps[1] = in1 ? (R * S * c1p + 31) / 32 : 0;
  a.Ktot = 32 * (a.ksteps[0] + a.ksteps[1]);
  a.N = N; a.H = H; a.W = W; a.P = P; a.Q = Q; a.R = R; a.S = S;
  a.stride = stride;
  a.stride_shift = stride == 1 ? 0 : (stride == 2 ? 1 : 2);
  a.pad_h = pad_h; a.pad_w = pad_w; a.transposed = transposed;
  a.wp = wp; a.out = out; a.coutp = coutp; a.bias = bias; a.stats = stats;
  a.stats_acc = stats ? msml_tl_stats_acc : 0;
  a.M = (long)N * P * Q;
  const int bn = msml_conv_tile_n(coutp);
  MSML_CHECK(kop >= cdiv(coutp, bn) * bn, MSML_ERR_SHAPE,
             "conv2d: packed weight has %d rows, need %d", kop, cdiv(coutp, bn) * bn);
  hipStream_t st = (hipStream_t)stream;
  // 7x1 / 1x7 line convs of the OSB's Global-Convolution modules (and their backward-data convs): conv_line.hip
  if (in_dtype == MSML_BF16 && out_dtype == MSML_BF16 && !in1 && !stats && !getenv("MSML_NO_FAST_CONV") &&
      msml_conv_line_applies(c0p, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w) &&
      msml_conv_line_dispatch(in0, c0p, wp, kop, bias, out, coutp, N, H, W, R, S, transposed, st)) {
    MSML_LAUNCH_OK("conv2d(line)");
    return MSML_OK;
  }
  // 4x4 / stride-2 transposed convs of the OSB decoder on cat(seg, gcm): conv_d4.hip
  if (in_dtype == MSML_BF16 && out_dtype == MSML_BF16 && in1 && !stats && !getenv("MSML_NO_FAST_CONV") &&
      msml_deconv4_applies(c0p, c1p, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed) &&
      msml_deconv4_dispatch(in0, in1, wp, kop, bias, out, N, H, st)) {
    MSML_LAUNCH_OK("conv2d(deconv4)");
    return MSML_OK;
  }
  if ((out_dtype == MSML_BF16 || out_dtype == MSML_F32) && !getenv("MSML_NO_FAST_CONV") &&
      msml_conv_fast_dispatch(in0, c0p, in1, c1p, wp, kop, bias, out, coutp, stats, N, H, W, P, Q, R,
                              S, stride, pad_h, pad_w, transposed, in_dtype, out_dtype, bn, st, nullptr, nullptr,
                              nullptr, 0, nullptr, nullptr)) {
    MSML_LAUNCH_OK("conv2d(fast)");
    return MSML_OK;
  }
#define CONV_CASE(TI, TO)                                                          \
  if (bn == 128) launch<TI, TO, 128, 128, 2, 2>(a, st);                            \
  else if (bn == 64) launch<TI, TO, 256, 64, 4, 1>(a, st);                         \
  else launch<TI, TO, 256, 32, 4, 1>(a, st);
  if (in_dtype == MSML_F32 && out_dtype == MSML_F32) { CONV_CASE(float, float) }
  else if (in_dtype == MSML_BF16 && out_dtype == MSML_BF16) { CONV_CASE(unsigned short, unsigned short) }
  else if (in_dtype == MSML_BF16 && out_dtype == MSML_F32) { CONV_CASE(unsigned short, float) }
  else {
    msml_set_error("conv2d: unsupported dtype pair in=%d out=%d", in_dtype, out_dtype);
    return MSML_ERR_DTYPE;
  }
#undef CONV_CASE
  MSML_LAUNCH_OK("conv2d");
  return MSML_OK;
}


// msml_conv2d whose per-channel (sum, sumsq) of the output go to an ACCUMULATOR (common.h): acc is a zero-initialised
// double[MSML_ACC_ROWS][2][coutp]; consumed by msml_bn_fin_act_fwd without a finalize launch in between.
extern "C" int msml_conv2d_acc(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                               const float* bias, void* out, int coutp, double* acc, int N, int H, int W, int P,
                               int Q, int R, int S, int stride, int pad_h, int pad_w, int transposed, int in_dtype,
                               int out_dtype, void* stream) {
  MSML_CHECK(acc, MSML_ERR_SHAPE, "conv2d_acc: null accumulator");
  msml_tl_stats_acc = 1;
  const int rc = msml_conv2d(in0, c0p, in1, c1p, wp, kop, bias, out, coutp, reinterpret_cast<float*>(acc), N, H, W, P, Q,
                             R, S, stride, pad_h, pad_w, transposed, in_dtype, out_dtype, stream);
  msml_tl_stats_acc = 0;
  return rc;
}


// Inference variant with the eval-mode BatchNorm (+ PReLU, + residual) folded into the conv
// epilogue (bf16 fast path only; see msml_hip.h).
extern "C" int msml_conv2d_fused(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                                 const float* scale, const float* shift, const float* alpha,
                                 const void* residual, int res_first, void* out, int coutp, int N, int H,
                                 int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                                 int transposed, void* stream) {
  MSML_CHECK(in0 && wp && out && shift, MSML_ERR_SHAPE, "conv2d_fused: null pointer");
  MSML_CHECK(c0p % 32 == 0 && c1p % 32 == 0 && coutp % 8 == 0, MSML_ERR_SHAPE, "conv2d_fused: channel padding");
  const int bn = msml_conv_tile_n(coutp);
  MSML_CHECK(kop >= cdiv(coutp, bn) * bn, MSML_ERR_SHAPE, "conv2d_fused: packed weight rows");
  // 7x1 / 1x7 line convs (the OSB's Global-Convolution modules: the backward-data conv of conv_l1 takes conv_r1's input
  // gradient as its residual, functional.py _ConvTee): conv_line.hip with scale / shift / residual in its epilogue
  if (!in1 && !alpha && !(residual && res_first) && !getenv("MSML_NO_FAST_CONV") &&
      msml_conv_line_applies(c0p, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w) &&
      msml_conv_line_dispatch(in0, c0p, wp, kop, shift, out, coutp, N, H, W, R, S, transposed, (hipStream_t)stream, scale,
                              residual)) {
    MSML_LAUNCH_OK("conv2d_fused(line)");
    return MSML_OK;
  }
  MSML_CHECK(msml_conv_fast_dispatch(in0, c0p, in1, c1p, wp, kop, shift, out, coutp, nullptr, N, H, W, P, Q,
                                     R, S, stride, pad_h, pad_w, transposed, MSML_BF16, MSML_BF16, bn,
                                     (hipStream_t)stream, scale, alpha, residual, res_first, nullptr, nullptr),
             MSML_ERR_UNSUPPORTED, "conv2d_fused: shape not supported by the fast kernel");
  MSML_LAUNCH_OK("conv2d_fused");
  return MSML_OK;
}


// Split-bf16 inference conv (msml_hip.h "bf16x3"): the fast kernel sees 3 * c0p (+ 3 * c1p) plain bf16
// input channels; its X3 epilogue writes the three output planes.
extern "C" int msml_conv2d_x3(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                              const float* scale, const float* shift, const float* alpha, const void* residual,
                              int res_first, void* out, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                              int stride, int pad_h, int pad_w, int transposed, void* stream) {
  MSML_CHECK(in0 && wp && out, MSML_ERR_SHAPE, "conv2d_x3: null pointer");
  MSML_CHECK(c0p > 0 && c0p % 32 == 0 && c1p % 32 == 0 && coutp > 0 && coutp % 32 == 0, MSML_ERR_SHAPE,
             "conv2d_x3: channel padding c0p=%d c1p=%d coutp=%d", c0p, c1p, coutp);
  const int bn = msml_conv_tile_n(coutp);
  MSML_CHECK(kop >= cdiv(coutp, bn) * bn, MSML_ERR_SHAPE, "conv2d_x3: packed weight rows");
  MSML_CHECK(msml_conv_fast_dispatch(in0, 3 * c0p, in1, 3 * c1p, wp, kop, shift, out, coutp, nullptr, N, H, W, P, Q,
                                     R, S, stride, pad_h, pad_w, transposed, MSML_BF16, MSML_BF16X3, bn,
                                     (hipStream_t)stream, scale, alpha, residual, res_first, nullptr, nullptr),
             MSML_ERR_UNSUPPORTED, "conv2d_x3: shape not supported by the fast kernel");
  MSML_LAUNCH_OK("conv2d_x3");
  return MSML_OK;
}


// The same conv with a BORDER-CLASS shift: shift9 = float[9][coutp], row (cy * 3 + cx), cy / cx = 0 on the first row /
// column of the output map, 2 on the last, 1 inside (3x3 / stride-1 / pad-1 forward only).  This is what folding an
// eval-mode BatchNorm IN FRONT of the conv into the conv needs (bn1 -> conv1 of IBasicBlock, backbones/frb/iresnet.py:58-60):
// conv(W, s x + t) = conv(W s, x) + sum over the taps INSIDE the map of W t -- the zero padding is applied after the
// BatchNorm, so the constant part depends on which taps of a border pixel fall outside.
extern "C" int msml_conv2d_x3_border(const void* in0, int c0p, const void* wp, int kop, const float* scale,
                                     const float* shift9, const float* alpha, const void* residual, int res_first, void* out,
                                     int coutp, int N, int H, int W, void* stream) {
  MSML_CHECK(in0 && wp && out && shift9, MSML_ERR_SHAPE, "conv2d_x3_border: null pointer");
  MSML_CHECK(c0p > 0 && c0p % 32 == 0 && coutp > 0 && coutp % 32 == 0 && H >= 2 && W >= 2, MSML_ERR_SHAPE,
             "conv2d_x3_border: c0p=%d coutp=%d H=%d W=%d", c0p, coutp, H, W);
  const int bn = msml_conv_tile_n(coutp);
  MSML_CHECK(kop >= cdiv(coutp, bn) * bn, MSML_ERR_SHAPE, "conv2d_x3_border: packed weight rows");
  msml_tl_bias9 = 1;
  const bool ok = msml_conv_fast_dispatch(in0, 3 * c0p, nullptr, 0, wp, kop, shift9, out, coutp, nullptr, N, H, W, H, W, 3, 3, 1,
                                          1, 1, 0, MSML_BF16, MSML_BF16X3, bn, (hipStream_t)stream, scale, alpha, residual,
                                          res_first, nullptr, nullptr);
  msml_tl_bias9 = 0;
  MSML_CHECK(ok, MSML_ERR_UNSUPPORTED, "conv2d_x3_border: shape not supported by the fast kernel");
  MSML_LAUNCH_OK("conv2d_x3_border");
  return MSML_OK;
}


// Backward-data conv whose output is the gradient of a training-mode BatchNorm(+PReLU) output:
// the epilogue also produces that BatchNorm's backward partial sums (see msml_hip.h).
extern "C" int msml_conv2d_bnbwd_rows(int coutp, int N, int P, int Q) {
  // upper bound of partial rows any kernel choice writes: one per pixel tile, x4 parity classes
  const long m = (long)N * P * Q;
  return 4 * (cdiv(m, 128) + 4);
}

extern "C" int msml_conv2d_bnbwd(const void* in0, int c0p, const void* wp, int kop, void* out, int coutp,
                                 int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                                 int pad_w, int transposed, const void* bn_x, const float* bn_scale,
                                 const float* bn_shift, const float* bn_alpha, const float* bn_mean,
                                 const float* bn_invstd, float* partial, int rows_cap, int* rows_used,
                                 void* stream) {
  MSML_CHECK(in0 && wp && out && bn_x && bn_scale && bn_shift && bn_mean && bn_invstd && partial && rows_used,
             MSML_ERR_SHAPE, "conv2d_bnbwd: null pointer");
  MSML_CHECK(c0p % 32 == 0 && coutp % 8 == 0, MSML_ERR_SHAPE, "conv2d_bnbwd: channel padding");
  MSML_CHECK(rows_cap >= msml_conv2d_bnbwd_rows(coutp, N, P, Q), MSML_ERR_WORKSPACE,
             "conv2d_bnbwd: partial buffer has %d rows, need %d", rows_cap, msml_conv2d_bnbwd_rows(coutp, N, P, Q));
  const int bn = msml_conv_tile_n(coutp);
  MSML_CHECK(kop >= cdiv(coutp, bn) * bn, MSML_ERR_SHAPE, "conv2d_bnbwd: packed weight rows");
  BnBwdFuse f{(const unsigned short*)bn_x, bn_scale, bn_shift, bn_alpha, bn_mean, bn_invstd, partial, msml_tl_stats_acc};
  MSML_CHECK(msml_conv_fast_dispatch(in0, c0p, nullptr, 0, wp, kop, nullptr, out, coutp, nullptr, N, H, W, P, Q,
                                     R, S, stride, pad_h, pad_w, transposed, MSML_BF16, MSML_BF16, bn,
                                     (hipStream_t)stream, nullptr, nullptr, nullptr, 0, &f, rows_used),
             MSML_ERR_UNSUPPORTED, "conv2d_bnbwd: shape not supported by the fast kernel");
  MSML_LAUNCH_OK("conv2d_bnbwd");
  return MSML_OK;
}

// msml_conv2d_bnbwd with the three backward sums added into an accumulator (zero-initialised double[8][3][coutp])
// instead of partial rows; consumed by msml_bn_fin_bwd_apply.
extern "C" int msml_conv2d_bnbwd_acc(const void* in0, int c0p, const void* wp, int kop, void* out, int coutp, int N,
                                     int H, int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                                     int transposed, const void* bn_x, const float* bn_scale, const float* bn_shift,
                                     const float* bn_alpha, const float* bn_mean, const float* bn_invstd, double* acc,
                                     void* stream) {
  MSML_CHECK(acc, MSML_ERR_SHAPE, "conv2d_bnbwd_acc: null accumulator");
  int used = 0;
  msml_tl_stats_acc = 1;
  const int rc = msml_conv2d_bnbwd(in0, c0p, wp, kop, out, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed,
                                   bn_x, bn_scale, bn_shift, bn_alpha, bn_mean, bn_invstd, reinterpret_cast<float*>(acc),
                                   1 << 30, &used, stream);
  msml_tl_stats_acc = 0;
  return rc;
}

int msml_conv_halo2_tiling(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                           int pad_h, int pad_w, int transposed, int x3);
int msml_conv_s2r_applies(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                          int pad_w, int transposed);
int msml_conv_halo_persist_shape(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                                 int pad_h, int pad_w);
bool msml_conv_halo_applies(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                            int stride, int pad_h, int pad_w, bool want_stats);
bool msml_conv_ws_applies(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                          int stride, int pad_h, int pad_w, bool want_stats);

// Forward conv whose input is X = PReLU(in0 * in_scale + in_shift) -- a training-mode BatchNorm
// (+PReLU) in front of the conv -- applied to the halo image while it sits in LDS, so X is never
// written to HBM (zero padding applies to X, as in the unfused graph).  Only the shapes of the
// halo-tile kernels (conv_ws.hip, conv_halo.hip): 3x3 / stride 1 / pad 1, see
// msml_conv2d_bnin_applies; MSML_ERR_UNSUPPORTED otherwise.  stats as in msml_conv2d.
bool msml_conv_halo_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out,
                             int coutp, float* stats, int N, int H, int W, int P, int Q, int R, int S,
                             int stride, int pad_h, int pad_w, int transposed, hipStream_t st,
                             const float* scale, const float* alpha, const void* residual, int res_first,
                             const BnBwdFuse* bnb, int* bnb_rows, const BnIn* xin, int x3 = 0, const BnBwdIn* bin = nullptr);
bool msml_conv_ws_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out,
                           int coutp, float* stats, int N, int H, int W, int P, int Q, int R, int S,
                           int stride, int pad_h, int pad_w, int transposed, hipStream_t st,
                           const float* scale, const float* alpha, const void* residual, int res_first,
                           const BnBwdFuse* bnb, int* bnb_rows, const BnIn* xin);

extern "C" int msml_conv2d_bnin_applies(int c0p, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                                        int stride, int pad_h, int pad_w, int want_stats) {
  if (getenv("MSML_NO_FAST_CONV") || c0p > 1024 || (long)N * P * Q >= (1L << 24)) return 0;
  const int bn = msml_conv_tile_n(coutp), kop = cdiv(coutp, bn) * bn;
#ifdef MSML_EXPERIMENTS      // (the weights-stationary kernel takes an input transform in experiment builds only)
  if (msml_conv_ws_applies(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, want_stats != 0)) return 1;
#endif
  return msml_conv_halo_applies(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, want_stats != 0) ? 1 : 0;
}

extern "C" int msml_conv2d_bnin(const void* in0, int c0p, const float* in_scale, const float* in_shift,
                                const float* in_alpha, const void* wp, int kop, void* out, int coutp,
                                float* stats, int N, int H, int W, int P, int Q, int R, int S, int stride,
                                int pad_h, int pad_w, void* stream) {
  MSML_CHECK(in0 && wp && out && in_scale && in_shift, MSML_ERR_SHAPE, "conv2d_bnin: null pointer");
  MSML_CHECK(N > 0 && H > 0 && W > 0 && c0p > 0 && c0p % 32 == 0 && coutp > 0 && coutp % 8 == 0, MSML_ERR_SHAPE,
             "conv2d_bnin: bad dims N=%d H=%d W=%d c0p=%d coutp=%d", N, H, W, c0p, coutp);
  const int bn = msml_conv_tile_n(coutp);
  MSML_CHECK(kop >= cdiv(coutp, bn) * bn, MSML_ERR_SHAPE, "conv2d_bnin: packed weight rows");
  MSML_CHECK(msml_conv2d_bnin_applies(c0p, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, stats != nullptr),
             MSML_ERR_UNSUPPORTED, "conv2d_bnin: shape not covered by the halo-tile kernels");
  const BnIn xin{in_scale, in_shift, in_alpha};
  hipStream_t st = (hipStream_t)stream;
  const bool ok =
      msml_conv_ws_dispatch(in0, c0p, wp, kop, nullptr, out, coutp, stats, N, H, W, P, Q, R, S, stride, pad_h, pad_w,
                            0, st, nullptr, nullptr, nullptr, 0, nullptr, nullptr, &xin) ||
      msml_conv_halo_dispatch(in0, c0p, wp, kop, nullptr, out, coutp, stats, N, H, W, P, Q, R, S, stride, pad_h,
                              pad_w, 0, st, nullptr, nullptr, nullptr, 0, nullptr, nullptr, &xin);
  MSML_CHECK(ok, MSML_ERR_UNSUPPORTED, "conv2d_bnin: launch refused");
  MSML_LAUNCH_OK("conv2d_bnin");
  return MSML_OK;
}

// 1 when msml_conv2d_bnin_acc serves the shape on the halo-tile conv, 2 on the weights-stationary 64-channel kernel
// (round 5: the in-LDS transform of that kernel keeps a lane's coefficients in registers and staggers the two waves of a
// SIMD, conv_ws.hip), 0 otherwise.
extern "C" int msml_conv2d_bnin_acc_applies(int c0p, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                                            int stride, int pad_h, int pad_w) {
  if (getenv("MSML_NO_FAST_CONV") || c0p > 1024 || c0p % 8 || 256 % (c0p / 8) || (long)N * P * Q >= (1L << 24)) return 0;
  const int bn = msml_conv_tile_n(coutp), kop = cdiv(coutp, bn) * bn;
#ifdef MSML_EXPERIMENTS      // (the weights-stationary kernel's prologue transform measured slower: experiment builds only)
  if (msml_conv_ws_applies(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, true)) return 2;
#endif
  // 3: the persistent 128-channel tile takes the launch (round 6: its prologue transform pays from 64 input channels on --
  // the one-slab shape 64 -> 128 @ 56x56, conv1 of a stage's first block, has no one-round kernel to fall back to)
  static const bool xfp = !(getenv("MSML_BNIN_ACC_PERSIST") && atoi(getenv("MSML_BNIN_ACC_PERSIST")) == 0);
  const char* m16e = getenv("MSML_HALO_M16");
  if (xfp && (!m16e || atoi(m16e) >= 2) &&
      msml_conv_halo_persist_shape(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w))
    return 3;
  return msml_conv_halo_applies(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, true) ? 1 : 0;
}

// Training-mode BatchNorm (+ PReLU) -> 3x3 / stride-1 / pad-1 conv in ONE launch, accumulator-mode statistics on both
// sides: the BatchNorm's coefficients are derived from `acc_in` (the f64 sums its producer accumulated) in the conv
// kernel's prologue, the normalised input is applied per slab in LDS AND written to `act_out` (the weight gradient reads
// it), `coef_out` = float[4][c0p] (scale, shift, mean, invstd) and the running statistics are written by one workgroup,
// the conv output's (sum, sumsq) go to `acc_out`.  Replaces msml_bn_fin_act_fwd + msml_conv2d_acc for the shapes of
// msml_conv2d_bnin_acc_applies, bit for bit (same coefficient arithmetic, same rounding of the activation, same MFMA
// order); reference: bn1 -> conv1 and bn2 -> prelu -> conv2 of IBasicBlock, backbones/frb/iresnet.py:58-62.
extern "C" int msml_conv2d_bnin_acc(const void* in0, int c0p, const double* acc_in, double count, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, float momentum,
                                    float eps, float* coef_out, const float* in_alpha, void* act_out, const void* wp,
                                    int kop, void* out, int coutp, double* acc_out, int N, int H, int W, int P, int Q,
                                    int R, int S, int stride, int pad_h, int pad_w, void* stream) {
  MSML_CHECK(in0 && wp && out && acc_in && coef_out && act_out && acc_out, MSML_ERR_SHAPE, "conv2d_bnin_acc: null pointer");
  MSML_CHECK(N > 0 && H > 0 && W > 0 && c0p > 0 && coutp > 0 && count > 0.0, MSML_ERR_SHAPE, "conv2d_bnin_acc: bad dims");
  const int bn = msml_conv_tile_n(coutp);
  MSML_CHECK(kop >= cdiv(coutp, bn) * bn, MSML_ERR_SHAPE, "conv2d_bnin_acc: packed weight rows");
  MSML_CHECK(msml_conv2d_bnin_acc_applies(c0p, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w), MSML_ERR_UNSUPPORTED,
             "conv2d_bnin_acc: shape not covered by the halo-tile kernel");
  BnIn xin{nullptr, nullptr, in_alpha};
  xin.scale = coef_out;                                // (non-null marks the transform; the kernel fills its own table)
  xin.shift = coef_out + c0p;
  xin.acc = acc_in; xin.count = count; xin.gamma = gamma; xin.beta = beta;
  xin.rmean = running_mean; xin.rvar = running_var; xin.momentum = momentum; xin.eps = eps;
  xin.coef_out = coef_out; xin.store = (unsigned short*)act_out;
  msml_tl_stats_acc = 1;
  const bool ok = msml_conv_ws_dispatch(in0, c0p, wp, kop, nullptr, out, coutp, reinterpret_cast<float*>(acc_out), N, H, W,
                                        P, Q, R, S, stride, pad_h, pad_w, 0, (hipStream_t)stream, nullptr, nullptr, nullptr,
                                        0, nullptr, nullptr, &xin) ||
                  msml_conv_halo_dispatch(in0, c0p, wp, kop, nullptr, out, coutp, reinterpret_cast<float*>(acc_out), N, H,
                                          W, P, Q, R, S, stride, pad_h, pad_w, 0, (hipStream_t)stream, nullptr, nullptr,
                                          nullptr, 0, nullptr, nullptr, &xin);
  msml_tl_stats_acc = 0;
  MSML_CHECK(ok, MSML_ERR_UNSUPPORTED, "conv2d_bnin_acc: launch refused");
  MSML_LAUNCH_OK("conv2d_bnin_acc");
  return MSML_OK;
}

// 1 when msml_conv2d_bnbwd_in_acc serves the shape: 3x3 / stride-1 backward-data conv on the halo-tile kernel, <= 512
// input channels (the LDS coefficient table), accumulator-mode sums on both BatchNorms.
extern "C" int msml_conv2d_bnbwd_in_acc_applies(int c0p, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                                                int stride, int pad_h, int pad_w) {
#ifndef MSML_EXPERIMENTS     // measured slower than the two launches: compiled into experiment builds only
  return 0;
#endif
  if (getenv("MSML_NO_FAST_CONV") || c0p > 512 || c0p % 8 || 256 % (c0p / 8) || coutp % 8 || 256 % (coutp / 8) ||
      (long)N * P * Q >= (1L << 24))
    return 0;
  const int bn = msml_conv_tile_n(coutp), kop = cdiv(coutp, bn) * bn;
  return msml_conv_halo_applies(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, false) ? 1 : 0;
}

// BatchNorm backward -> backward-data conv -> (sums of the next BatchNorm backward) in ONE launch, accumulator mode:
// in0 = dy, the gradient of the UPPER BatchNorm's output (bn3 of an IBasicBlock for conv2's backward-data, bn2 (+ PReLU)
// for conv1's: backbones/frb/iresnet.py:59-65); up_* describe that BatchNorm (saved input up_x, coefficients, the three
// sums up_acc = double[8][3][c0p] its producer accumulated); its input gradient dc = scale * (g - s0 / n - xhat * s1 / n)
// is formed per slab in LDS, written through to dc_out (the weight gradient reads it) and convolved; one workgroup adds
// dgamma / dbeta / dalpha.  The conv epilogue then accumulates the sums of the LOWER BatchNorm (bn_*: msml_conv2d_bnbwd_acc).
// Replaces msml_bn_fin_bwd_apply (no add / next) + msml_conv2d_bnbwd_acc bit for bit.
extern "C" int msml_conv2d_bnbwd_in_acc(const void* in0, int c0p, const void* up_x, const float* up_scale,
                                        const float* up_shift, const float* up_alpha, const float* up_mean,
                                        const float* up_invstd, const double* up_acc, float* dgamma, float* dbeta,
                                        float* dalpha, int accumulate, void* dc_out, const void* wp, int kop, void* out,
                                        int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                                        int pad_h, int pad_w, const void* bn_x, const float* bn_scale,
                                        const float* bn_shift, const float* bn_alpha, const float* bn_mean,
                                        const float* bn_invstd, double* acc, void* stream) {
  MSML_CHECK(in0 && up_x && up_scale && up_shift && up_mean && up_invstd && up_acc && dc_out && wp && out && bn_x &&
                 bn_scale && bn_shift && bn_mean && bn_invstd && acc, MSML_ERR_SHAPE, "conv2d_bnbwd_in_acc: null pointer");
  const int bn = msml_conv_tile_n(coutp);
  MSML_CHECK(kop >= cdiv(coutp, bn) * bn, MSML_ERR_SHAPE, "conv2d_bnbwd_in_acc: packed weight rows");
  MSML_CHECK(msml_conv2d_bnbwd_in_acc_applies(c0p, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w), MSML_ERR_UNSUPPORTED,
             "conv2d_bnbwd_in_acc: shape not covered by the halo-tile kernel");
  BnBwdFuse f{(const unsigned short*)bn_x, bn_scale, bn_shift, bn_alpha, bn_mean, bn_invstd, reinterpret_cast<float*>(acc), 1};
  BnBwdIn b;
  b.x = (const unsigned short*)up_x; b.scale = up_scale; b.shift = up_shift; b.alpha = up_alpha; b.mean = up_mean;
  b.invstd = up_invstd; b.acc = up_acc; b.count = (double)N * H * W; b.dgamma = dgamma; b.dbeta = dbeta; b.dalpha = dalpha;
  b.accumulate = accumulate; b.store = (unsigned short*)dc_out;
  int rows = 0;
  const bool ok = msml_conv_halo_dispatch(in0, c0p, wp, kop, nullptr, out, coutp, nullptr, N, H, W, P, Q, R, S, stride,
                                          pad_h, pad_w, 1, (hipStream_t)stream, nullptr, nullptr, nullptr, 0, &f, &rows,
                                          nullptr, 0, &b);
  MSML_CHECK(ok, MSML_ERR_UNSUPPORTED, "conv2d_bnbwd_in_acc: launch refused");
  MSML_LAUNCH_OK("conv2d_bnbwd_in_acc");
  return MSML_OK;
}

// Name of the kernel msml_conv2d / msml_conv2d_fused / msml_conv2d_bnbwd launches for a shape
// (profiling labels: bench.py's roofline names the kernel it measured).
extern "C" const char* msml_conv2d_kernel(int c0p, int c1p, int coutp, int N, int H, int W, int P, int Q,
                                          int R, int S, int stride, int pad_h, int pad_w, int transposed,
                                          int in_dtype, int out_dtype, int want_stats) {
  const int bn = msml_conv_tile_n(coutp);
  const bool fast = in_dtype == MSML_BF16 && !getenv("MSML_NO_FAST_CONV") && c0p % 32 == 0 && c1p % 32 == 0 &&
                    (c1p == 0 || ((R * S * (c0p / 32)) & 1) == 0) && (long)N * P * Q < (1L << 24);
  if (fast && out_dtype == MSML_BF16 && !want_stats &&
      msml_deconv4_applies(c0p, c1p, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed))
    return "k_deconv4_fwd<14x16 input tile, 4 parity classes, weights resident>";
  if (fast && c1p == 0 && out_dtype == MSML_BF16 && !want_stats &&
      msml_conv_line_applies(c0p, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w))
    return "k_conv_line<full lines in LDS, weights resident, persistent>";
  if (fast && c1p == 0 && out_dtype == MSML_BF16 &&
      msml_conv_ws_applies(c0p, cdiv(coutp, bn) * bn, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, want_stats != 0))
    return "k_conv_ws<64 -> 64 channels, weights resident, persistent>";    // (or k_conv_s2r<stride 1> without fused sums / residual)
  if (fast && c1p == 0 && out_dtype == MSML_BF16 &&
      msml_conv_halo_persist_shape(c0p, cdiv(coutp, bn) * bn, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w))
    return "k_conv_halo_p<14x14 px x 128 ch, persistent>";       // (training launches: no affine epilogue, accumulator-mode sums)
  if (fast && c1p == 0 && out_dtype == MSML_BF16 &&
      msml_conv_halo_applies(c0p, cdiv(coutp, bn) * bn, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, want_stats != 0))
    return coutp % 256 == 0 ? "k_conv_halo<14x14 px x 256 ch, 8 waves>" : "k_conv_halo<14x14 px x 128 ch, 8 waves>";
  if (fast && c1p == 0 && out_dtype == MSML_BF16 &&
      msml_conv_s2r_applies(c0p, cdiv(coutp, bn) * bn, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed))
    return transposed ? "k_conv_s2r<64 -> 64 stride-2 backward-data, weights in registers, persistent>"
                      : "k_conv_s2r<64 -> 64 stride-2 forward, 4 parity planes, weights in registers, persistent>";
  if (fast && c1p == 0 && out_dtype == MSML_BF16) {
    const int t2 = msml_conv_halo2_tiling(c0p, cdiv(coutp, bn) * bn, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed, 0);
    if (t2 == 3) return "k_conv_halo2<mosaic of six 4x4 images x 128 ch>";
    if (t2 == 2) return stride == 1 ? "k_conv_halo2<mosaic of four 7x7 images x 128 ch>"
                                    : (transposed ? "k_conv_halo2<stride-2 backward-data, 4 classes, 7x7 mosaic x 128 ch>"
                                                  : "k_conv_halo2<stride-2 forward, 4 parity planes, 7x7 mosaic x 128 ch>");
    if (t2 == 1) return transposed ? "k_conv_halo2<stride-2 backward-data, 4 output classes, 14x14 px>"
                                   : "k_conv_halo2<stride-2 forward, 4 parity planes, 14x14 px>";
  }
  if (fast) return bn == 128 ? "k_conv_fast<128 x 128, 4 waves>" : (bn == 64 ? "k_conv_fast<256 x 64, 4 waves>" : "k_conv_fast<256 x 32, 4 waves>");
  return bn == 128 ? "k_conv_igemm<128 x 128>" : (bn == 64 ? "k_conv_igemm<256 x 64>" : "k_conv_igemm<256 x 32>");
}
