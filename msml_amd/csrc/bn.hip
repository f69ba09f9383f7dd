// BatchNorm (train + eval) / PReLU / residual kernels on NHWC [M pixels][C] tensors, C % 8 == 0.
//
// Reference call sites: nn.BatchNorm2d + nn.PReLU + `out += identity` in IBasicBlock
// (backbones/frb/iresnet.py:56-67, backbones/osb/unet.py:80-91), resblock_bottle
// (backbones/fm/fmoperator.py:53-68), stems (iresnet.py:209-211), bn2 / BatchNorm1d features
// (iresnet.py:225,233).  BN is per-GPU (no SyncBN: train.py:135-137).
//
// HBM-bound streaming kernels; per-channel reductions are two-level: every workgroup reduces
// its pixel slab in registers -> LDS -> one partial row in a workspace (plain stores), and a
// tiny finalize kernel sums the rows in f64 in a fixed order (deterministic; no float atomics).
#include <stdlib.h>

#include "common.h"

#define RED_ROWS_MAX 1024      // partial rows produced by the standalone reduction kernels

// ------------------------------------------------------------------ per-channel reduction ---
// Generic slab reducer: thread (cx, py) accumulates NQ quantities x 8 channels over pixels
// py, py+PY, ... of the block's slab, then LDS-reduces over py.  F: functor
//   D ld(long pix, int c8)                   loads one pixel chunk,
//   void f(const D& d, float (&q)[NQ][8])    adds its contribution.
// PIPE: four pixels per trip (see below) -- for launches with few workgroups
template <int NQ, bool PIPE, typename L, typename F>
__device__ __forceinline__ void slab_reduce(long M, int C, float* __restrict__ partial, L ld, F f, int acc_mode = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);            // [PY][NQ][C]
  const int C8 = C / 8;
  const int PY = 256 / C8 > 0 ? 256 / C8 : 1;                 // pixel lanes per block
  MSML_LDS_REGION(red, PY * NQ * C * 4);
  const int t = threadIdx.x;
  const int cx = t % C8, py = t / C8;
  float q[NQ][8];
#pragma unroll
  for (int a = 0; a < NQ; a++)
#pragma unroll
    for (int j = 0; j < 8; j++) q[a][j] = 0.f;
  const long per = (M + gridDim.x - 1) / gridDim.x;
  const long beg = blockIdx.x * per;
  long end = beg + per;
  if (end > M) end = M;
  // PIPE: four pixels per trip, all their loads issued before the first accumulation (a rolled loop
  // waits for every pixel's loads in turn: with fewer workgroups than ~2 per CU the launch is bound
  // by that latency chain -- 256 x 14 x 14 x 256: 38.9 -> 36.0 us; large tensors lose occupancy to
  // the extra registers and keep the rolled loop).  Same accumulation order (ascending pixels).
  if (py < PY) {
    long pix = beg + py;
    if (PIPE)
    for (; pix + 3 * PY < end; pix += 4 * PY) {
      auto d0 = ld(pix, cx), d1 = ld(pix + PY, cx), d2 = ld(pix + 2 * PY, cx), d3 = ld(pix + 3 * PY, cx);
      f(d0, q);
      f(d1, q);
      f(d2, q);
      f(d3, q);
    }
    for (; pix < end; pix += PY) f(ld(pix, cx), q);
  }
  // C8 may exceed 256 only if C > 2048: not supported by the launchers
  if (py < PY) {
#pragma unroll
    for (int a = 0; a < NQ; a++)
#pragma unroll
      for (int j = 0; j < 8; j++) red[(py * NQ + a) * C + cx * 8 + j] = q[a][j];
  }
  __syncthreads();
  for (int i = t; i < NQ * C; i += blockDim.x) {
    float s = 0.f;
    for (int y = 0; y < PY; y++) s += red[y * NQ * C + i];
    if (acc_mode)          // accumulator mode (common.h): partial is a zero-initialised double[MSML_ACC_ROWS][NQ][C]
      unsafeAtomicAdd(reinterpret_cast<double*>(partial) + (long)(blockIdx.x & (MSML_ACC_ROWS - 1)) * NQ * C + i, (double)s);
    else
      partial[(long)blockIdx.x * NQ * C + i] = s;
  }
}

static inline int red_rows(long M, int C) {
  int py = 256 / (C / 8) > 0 ? 256 / (C / 8) : 1;
  static const long ppt = getenv("MSML_RED_PPT") ? atol(getenv("MSML_RED_PPT")) : 16;
  long rows = (M + (long)py * ppt - 1) / ((long)py * ppt);     // >= ppt pixels per thread (4 / 8 measured
  // slower end to end: more partial rows for the finalize)
  if (rows < 1) rows = 1;
  if (rows > RED_ROWS_MAX) rows = RED_ROWS_MAX;
  return (int)rows;
}
static inline size_t red_lds(int NQ, int C) {
  int py = 256 / (C / 8) > 0 ? 256 / (C / 8) : 1;
  return (size_t)py * NQ * C * sizeof(float);
}

#define RED_PIPE_MAX_ROWS 512
template <typename T, bool PIPE>
__global__ void __launch_bounds__(256) k_bn_stats(const T* __restrict__ x, long M, int C,
                                                  float* __restrict__ partial, int acc_mode) {
  slab_reduce<2, PIPE>(M, C, partial, [&](long pix, int c8) { return load8<T>(x + pix * C + c8 * 8); },
                 [&](const Vec8& v, float(&q)[2][8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      q[0][j] += v.v[j];
      q[1][j] += v.v[j] * v.v[j];
    }
  }, acc_mode);
}

extern "C" int msml_bn_stats_rows(long M, int C) { return red_rows(M, C); }

extern "C" int msml_bn_stats(const void* x, long M, int C, float* partial, int dtype, void* stream) {
  MSML_CHECK(x && partial && M > 0 && C > 0 && C % 8 == 0 && C <= 2048, MSML_ERR_SHAPE,
             "bn_stats: bad shape M=%ld C=%d", M, C);
  int rows = red_rows(M, C);
  MSML_DISPATCH_DTYPE(dtype, "bn_stats",
                      if (rows <= RED_PIPE_MAX_ROWS)
                        (k_bn_stats<DT, true>)<<<rows, 256, red_lds(2, C), (hipStream_t)stream>>>((const DT*)x, M, C, partial, msml_tl_stats_acc);
                      else
                        (k_bn_stats<DT, false>)<<<rows, 256, red_lds(2, C), (hipStream_t)stream>>>((const DT*)x, M, C, partial, msml_tl_stats_acc);)
  MSML_LAUNCH_OK("bn_stats");
  return MSML_OK;
}

// msml_bn_stats into an accumulator (zero-initialised double[8][2][C], common.h) for msml_bn_fin_act_fwd
extern "C" int msml_bn_stats_acc(const void* x, long M, int C, double* acc, int dtype, void* stream) {
  MSML_CHECK(acc, MSML_ERR_SHAPE, "bn_stats_acc: null accumulator");
  msml_tl_stats_acc = 1;
  const int rc = msml_bn_stats(x, M, C, reinterpret_cast<float*>(acc), dtype, stream);
  msml_tl_stats_acc = 0;
  return rc;
}

// ------------------------------------------------------------------ finalize (forward) -------
// partial: [rows][2][C] (sum, sumsq).  rows == 0 -> eval mode: coefficients from running stats.
// Training also updates the running statistics exactly like nn.BatchNorm (momentum, unbiased
// variance) and stores mean / invstd for the backward.
// One workgroup per FIN_CPB = 8 channels: 128 row-lanes x 8 channels (these kernels are chains of
// dependent loads on a few KB of partial rows -- latency, not bandwidth: with 32 channels x 32
// row-lanes per workgroup a 512-row reduce took 10-30 us and the 266 finalize launches of a step
// 2.6 ms), f64 accumulation, fixed-order two-level LDS tree (deterministic).
#define FIN_CPB 8
#define FIN_LANES (1024 / FIN_CPB)
template <int NQ>
__device__ __forceinline__ void fin_reduce(const float* __restrict__ partial, int rows, int C, int c,
                                           bool cok, double (&out)[NQ]) {
  __shared__ double red[FIN_LANES][NQ][FIN_CPB + 1];
  __shared__ double red2[16][NQ][FIN_CPB + 1];
  const int cx = threadIdx.x & (FIN_CPB - 1), ry = threadIdx.x / FIN_CPB;
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;
  if (cok) {
    // four independent load streams per thread
    double a1[NQ], a2[NQ], a3[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) a1[q] = a2[q] = a3[q] = 0.0;
    int r = ry;
    for (; r + 3 * FIN_LANES < rows; r += 4 * FIN_LANES)
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        acc[q] += (double)partial[((long)r * NQ + q) * C + c];
        a1[q] += (double)partial[((long)(r + FIN_LANES) * NQ + q) * C + c];
        a2[q] += (double)partial[((long)(r + 2 * FIN_LANES) * NQ + q) * C + c];
        a3[q] += (double)partial[((long)(r + 3 * FIN_LANES) * NQ + q) * C + c];
      }
    for (; r < rows; r += FIN_LANES)
#pragma unroll
      for (int q = 0; q < NQ; q++) acc[q] += (double)partial[((long)r * NQ + q) * C + c];
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = (acc[q] + a1[q]) + (a2[q] + a3[q]);
  }
#pragma unroll
  for (int q = 0; q < NQ; q++) red[ry][q][cx] = acc[q];
  __syncthreads();
  if (ry < 16) {                                       // 16 groups of FIN_LANES / 16 row-lanes
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      double s = 0.0;
#pragma unroll
      for (int y = 0; y < FIN_LANES / 16; y++) s += red[ry * (FIN_LANES / 16) + y][q][cx];
      red2[ry][q][cx] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    double s = 0.0;
    if (ry == 0)
#pragma unroll
      for (int y = 0; y < 16; y++) s += red2[y][q][cx];
    out[q] = s;
  }
}

__global__ void __launch_bounds__(1024) k_bn_finalize(const float* __restrict__ partial, int rows, int C, double count,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                              float eps, float* __restrict__ scale, float* __restrict__ shift,
                              float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x * FIN_CPB + (threadIdx.x & (FIN_CPB - 1));
  const bool cok = c < C;
  float mean = 0.f, invstd = 1.f;
  if (rows > 0) {
    double s[2];
    fin_reduce<2>(partial, rows, C, c, cok, s);
    if (threadIdx.x >= FIN_CPB || !cok) return;
    double m = s[0] / count;
    double var = s[1] / count - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
  } else {
    if (threadIdx.x >= FIN_CPB || !cok) return;
    mean = rmean[c];
    invstd = 1.0f / sqrtf(rvar[c] + eps);
  }
  float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * invstd;
  shift[c] = b - mean * g * invstd;
  if (save_mean) {
    save_mean[c] = mean;
    save_invstd[c] = invstd;
  }
}

extern "C" int msml_bn_finalize(const float* partial, int rows, int C, double count,
                                const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float momentum, float eps, float* scale,
                                float* shift, float* save_mean, float* save_invstd, void* stream) {
  MSML_CHECK(C > 0 && scale && shift && rows >= 0, MSML_ERR_SHAPE, "bn_finalize: bad args");
  MSML_CHECK(rows > 0 ? (partial && count > 0) : (running_mean && running_var), MSML_ERR_SHAPE,
             "bn_finalize: train needs partials, eval needs running stats");
  k_bn_finalize<<<cdiv(C, FIN_CPB), 1024, 0, (hipStream_t)stream>>>(
      partial, rows, C, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift,
      save_mean, save_invstd);
  MSML_LAUNCH_OK("bn_finalize");
  return MSML_OK;
}

// ------------------------------------------------------------------ apply (forward) ----------
// y = prelu(x * scale[c] + shift[c]) + residual      (alpha == null: no PReLU; residual optional)
// The grid is sized so that (threads in grid) % (C/8) == 0: a thread's 8-channel chunk is then
// loop-invariant and its coefficients live in registers (one load each, not one per pixel).
struct Coef8 {
  float v[8];
};
__device__ __forceinline__ Coef8 ldc8(const float* p, int c0, float dflt) {
  Coef8 r;
  if (p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p + c0);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + c0 + 4);
#pragma unroll
    for (int i = 0; i < 4; i++) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = dflt;
  }
  return r;
}

// STATS: also emit per-channel (sum, sumsq) partial rows of the (storage-rounded) OUTPUT, one row
// pair per workgroup -- the statistics the next block's leading BatchNorm needs (the block output
// is otherwise re-read by a k_bn_stats pass).  Needs C8 | 256.
template <typename T, bool STATS>
__global__ void __launch_bounds__(256) k_bn_act_fwd(const T* __restrict__ x, const float* __restrict__ scale,
                                                    const float* __restrict__ shift,
                                                    const float* __restrict__ alpha,
                                                    const T* __restrict__ residual, int res_first,
                                                    T* __restrict__ y, long n8, int C8, float* __restrict__ stats) {
  const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const int c0 = (int)(tid % C8) * 8;
  const Coef8 sc = ldc8(scale, c0, 1.f), sh = ldc8(shift, c0, 0.f), al = ldc8(alpha, c0, 1.f);
  float q1[8], q2[8];
#pragma unroll
  for (int j = 0; j < 8; j++) q1[j] = q2[j] = 0.f;
  for (long i = tid; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 v = load8<T>(x + i * 8);
    Vec8 r;
    if (residual) r = load8<T>(residual + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float z = v.v[j] * sc.v[j] + sh.v[j];
      if (residual && res_first) z += r.v[j];
      if (alpha) z = z > 0.f ? z : z * al.v[j];
      if (residual && !res_first) z += r.v[j];
      v.v[j] = z;
    }
    store8<T>(y + i * 8, v);
    if (STATS) {
      const Vec8 vr = round8<T>(v);                    // what the consumer will read back
#pragma unroll
      for (int j = 0; j < 8; j++) {
        q1[j] += vr.v[j];
        q2[j] += vr.v[j] * vr.v[j];
      }
    }
  }
  if (STATS) {
    // threads t, t + C8, ... of the block share a channel chunk: fixed-order sum through LDS
    __shared__ float red[2][256][9];
    const int t = threadIdx.x, C = C8 * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      red[0][t][j] = q1[j];
      red[1][t][j] = q2[j];
    }
    __syncthreads();
    for (int i = t; i < 2 * C; i += 256) {
      const int which = i / C, c = i % C, cx = c >> 3, j = c & 7;
      float sum = 0.f;
      for (int k = cx; k < 256; k += C8) sum += red[which][k][j];
      stats[((long)blockIdx.x * 2 + which) * C + c] = sum;
    }
  }
}

static inline int ew_grid(long n8) {
  static const long cap = getenv("MSML_EW_GRID") ? atol(getenv("MSML_EW_GRID")) : 768;     // 3 per CU: with the accumulator fold in every
  // workgroup's prologue 768 beats 1024 by 0.25 ms per step (640 / 896 equal, 512 / 1536 / 2048 slower)
  long b = (n8 + 255) / 256;
  return (int)(b < cap ? b : cap);
}
// grid whose total thread count is a multiple of C8 (C8 <= 256 or a multiple of 256)
static inline int ew_grid_c(long n8, int C8) {
  int g = ew_grid(n8);
  if (C8 > 256) {
    int m = C8 / 256;
    g = (g + m - 1) / m * m;
  }
  return g;
}

extern "C" int msml_bn_act_fwd(const void* x, const float* scale, const float* shift,
                               const float* alpha, const void* residual, int res_first, void* y,
                               long M, int C, int dtype, void* stream) {
  MSML_CHECK(x && y && scale && shift && M > 0 && C > 0 && C % 8 == 0, MSML_ERR_SHAPE,
             "bn_act_fwd: bad shape M=%ld C=%d", M, C);
  MSML_CHECK(256 % (C / 8) == 0 || (C / 8) % 256 == 0, MSML_ERR_UNSUPPORTED,
             "bn_act_fwd: C/8 = %d must divide 256 or be a multiple of it", C / 8);
  long n8 = M * (C / 8);
  MSML_DISPATCH_DTYPE(dtype, "bn_act_fwd",
                      (k_bn_act_fwd<DT, false>)<<<ew_grid_c(n8, C / 8), 256, 0, (hipStream_t)stream>>>(
                          (const DT*)x, scale, shift, alpha, (const DT*)residual, res_first, (DT*)y, n8,
                          C / 8, nullptr);)
  MSML_LAUNCH_OK("bn_act_fwd");
  return MSML_OK;
}

// Same, plus the (sum, sumsq) partial rows of the output: stats[msml_bn_act_fwd_stats_rows(M, C)][2][C]
// in the row format of msml_bn_stats (feed it to msml_bn_finalize of the next BatchNorm).
extern "C" int msml_bn_act_fwd_stats_rows(long M, int C) { return ew_grid_c(M * (C / 8), C / 8); }

extern "C" int msml_bn_act_fwd_stats(const void* x, const float* scale, const float* shift, const float* alpha,
                                     const void* residual, int res_first, void* y, long M, int C,
                                     float* stats, int dtype, void* stream) {
  MSML_CHECK(x && y && scale && shift && stats && M > 0 && C > 0 && C % 8 == 0, MSML_ERR_SHAPE,
             "bn_act_fwd_stats: bad shape M=%ld C=%d", M, C);
  MSML_CHECK(256 % (C / 8) == 0, MSML_ERR_UNSUPPORTED, "bn_act_fwd_stats: C/8 = %d must divide 256", C / 8);
  long n8 = M * (C / 8);
  MSML_DISPATCH_DTYPE(dtype, "bn_act_fwd_stats",
                      (k_bn_act_fwd<DT, true>)<<<ew_grid_c(n8, C / 8), 256, 0, (hipStream_t)stream>>>(
                          (const DT*)x, scale, shift, alpha, (const DT*)residual, res_first, (DT*)y, n8,
                          C / 8, stats);)
  MSML_LAUNCH_OK("bn_act_fwd_stats");
  return MSML_OK;
}

// ------------------------------------------------------------------ finalize + apply, one launch ----
// Accumulator mode (common.h): the producer of x added its per-workgroup (sum, sumsq) into acc[MSML_ACC_ROWS][2][C]
// (f64 atomics); every workgroup folds the rows and derives (scale, shift) of all C channels into LDS -- 32 KB of L2
// reads and C rsqrt per workgroup -- and workgroup 0 also writes them (with mean / invstd for the backward) and
// updates the running statistics exactly like k_bn_finalize.  The 5-7 us finalize launch between the producer and
// this kernel disappears (266 of them per training step of ires50-MSML).  STATS: the (sum, sumsq) of the OUTPUT go to
// acc_out the same way (next block's leading BatchNorm).
template <typename T, bool STATS>
__global__ void __launch_bounds__(256) k_bn_fin_act_fwd(const double* __restrict__ acc, double count,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                                        float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                        float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                        const T* __restrict__ x, const float* __restrict__ alpha,
                                                        const T* __restrict__ residual, int res_first,
                                                        T* __restrict__ y, long n8, int C8, double* __restrict__ acc_out) {
  extern __shared__ float cs[];                        // [2][C]: scale, shift
  const int C = C8 * 8, t = threadIdx.x;
  MSML_LDS_REGION(cs, 2 * C * 4);
  for (int c = t; c < C; c += 256) {
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int r = 0; r < MSML_ACC_ROWS; r++) {
      s += acc[(r * 2 + 0) * C + c];
      ss += acc[(r * 2 + 1) * C + c];
    }
    const double m = s / count;
    double var = ss / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd, sh = b - mean * g * invstd;
    cs[c] = sc;
    cs[C + c] = sh;
    if (blockIdx.x == 0) {
      scale[c] = sc;
      shift[c] = sh;
      save_mean[c] = mean;
      save_invstd[c] = invstd;
      if (rmean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
      }
    }
  }
  __syncthreads();
  const long tid = blockIdx.x * (long)blockDim.x + t;
  const int c0 = (int)(tid % C8) * 8;
  Coef8 sc, sh;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    sc.v[j] = cs[c0 + j];
    sh.v[j] = cs[C + c0 + j];
  }
  const Coef8 al = ldc8(alpha, c0, 1.f);
  float q1[8], q2[8];
#pragma unroll
  for (int j = 0; j < 8; j++) q1[j] = q2[j] = 0.f;
  for (long i = tid; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 v = load8<T>(x + i * 8);
    Vec8 r;
    if (residual) r = load8<T>(residual + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float z = v.v[j] * sc.v[j] + sh.v[j];
      if (residual && res_first) z += r.v[j];
      if (alpha) z = z > 0.f ? z : z * al.v[j];
      if (residual && !res_first) z += r.v[j];
      v.v[j] = z;
    }
    store8<T>(y + i * 8, v);
    if (STATS) {
      const Vec8 vr = round8<T>(v);                    // what the consumer will read back
#pragma unroll
      for (int j = 0; j < 8; j++) {
        q1[j] += vr.v[j];
        q2[j] += vr.v[j] * vr.v[j];
      }
    }
  }
  if (STATS) {
    __shared__ float red[2][256][9];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      red[0][t][j] = q1[j];
      red[1][t][j] = q2[j];
    }
    __syncthreads();
    for (int i = t; i < 2 * C; i += 256) {
      const int which = i / C, c = i % C, cx = c >> 3, j = c & 7;
      float sum = 0.f;
      for (int k = cx; k < 256; k += C8) sum += red[which][k][j];
      stats_emit(reinterpret_cast<float*>(acc_out), 1, blockIdx.x, which, C, c, sum);
    }
  }
}

extern "C" int msml_bn_fin_act_fwd(const double* acc, double count, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, float* scale,
                                   float* shift, float* save_mean, float* save_invstd, const void* x,
                                   const float* alpha, const void* residual, int res_first, void* y, long M, int C,
                                   double* acc_out, int dtype, void* stream) {
  MSML_CHECK(acc && x && y && scale && shift && save_mean && save_invstd && M > 0 && C > 0 && C % 8 == 0 && count > 0,
             MSML_ERR_SHAPE, "bn_fin_act_fwd: bad arguments M=%ld C=%d", M, C);
  MSML_CHECK(256 % (C / 8) == 0, MSML_ERR_UNSUPPORTED, "bn_fin_act_fwd: C/8 = %d must divide 256", C / 8);
  const long n8 = M * (C / 8);
  const size_t lds = (size_t)2 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (acc_out) {
    MSML_DISPATCH_DTYPE(dtype, "bn_fin_act_fwd",
                        (k_bn_fin_act_fwd<DT, true>)<<<ew_grid_c(n8, C / 8), 256, lds, st>>>(
                            acc, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean,
                            save_invstd, (const DT*)x, alpha, (const DT*)residual, res_first, (DT*)y, n8, C / 8, acc_out);)
  } else {
    MSML_DISPATCH_DTYPE(dtype, "bn_fin_act_fwd",
                        (k_bn_fin_act_fwd<DT, false>)<<<ew_grid_c(n8, C / 8), 256, lds, st>>>(
                            acc, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean,
                            save_invstd, (const DT*)x, alpha, (const DT*)residual, res_first, (DT*)y, n8, C / 8, nullptr);)
  }
  MSML_LAUNCH_OK("bn_fin_act_fwd");
  return MSML_OK;
}

// ------------------------------------------------------------------ backward ------------------
// With z = x*scale + shift, g = dy * prelu'(z), xhat = (x - mean) * invstd:
//   reduce: s1 = sum g, s2 = sum g * xhat, s3 = sum dy * min(z, 0)   (d alpha)
//   apply:  dx = gamma * invstd * (g - s1/n - xhat * s2/n)
template <typename T, bool PIPE>
__global__ void __launch_bounds__(256) k_bn_bwd_reduce(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const T* __restrict__ res, long M, int C,
                                                       float* __restrict__ partial, int acc_mode) {
  const int c0h = (threadIdx.x % (C / 8)) * 8;
  const Coef8 sc = ldc8(scale, c0h, 1.f), sh = ldc8(shift, c0h, 0.f), al = ldc8(alpha, c0h, 1.f);
  const Coef8 mu = ldc8(mean, c0h, 0.f), is = ldc8(invstd, c0h, 1.f);
  struct Px {
    Vec8 g, v, rr;
  };
  slab_reduce<3, PIPE>(M, C, partial, [&](long pix, int c8) {
    Px d;
    d.g = load8<T>(dy + pix * C + c8 * 8);
    d.v = load8<T>(x + pix * C + c8 * 8);
    if (res) d.rr = load8<T>(res + pix * C + c8 * 8);
    return d;
  }, [&](const Px& d, float(&q)[3][8]) {
    const Vec8 &g = d.g, &v = d.v, &rr = d.rr;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float gg = g.v[j];
      if (alpha) {
        float z = v.v[j] * sc.v[j] + sh.v[j];
        if (res) z += rr.v[j];
        if (z <= 0.f) {
          q[2][j] += gg * z;
          gg *= al.v[j];
        }
      }
      float xh = (v.v[j] - mu.v[j]) * is.v[j];
      q[0][j] += gg;
      q[1][j] += gg * xh;
    }
  }, acc_mode);
}

// Many partial rows (a fused conv epilogue writes one per workgroup: up to ~50k) are first folded
// to FOLD_ROWS rows by FOLD_ROWS x C/32 workgroups; each fold block owns a contiguous chunk, so
// the result does not depend on scheduling.
#define FOLD_ROWS 32
#define FOLD_MIN_ROWS 512
template <int NQ>
__global__ void __launch_bounds__(1024) k_fold_rows(const float* __restrict__ partial, int rows, int C,
                                                    float* __restrict__ out) {
  const int c = blockIdx.x * FIN_CPB + (threadIdx.x & (FIN_CPB - 1));
  const bool cok = c < C;
  const int chunk = (rows + FOLD_ROWS - 1) / FOLD_ROWS;
  const int r0 = blockIdx.y * chunk;
  int nr = rows - r0;
  if (nr > chunk) nr = chunk;
  if (nr < 0) nr = 0;
  double s[NQ];
  fin_reduce<NQ>(partial + (long)r0 * NQ * C, nr, C, c, cok, s);
  if (threadIdx.x >= FIN_CPB || !cok) return;
#pragma unroll
  for (int q = 0; q < NQ; q++) out[((long)blockIdx.y * NQ + q) * C + c] = (float)s[q];
}

__global__ void __launch_bounds__(1024) k_bn_bwd_finalize(const float* __restrict__ partial, int rows, int C, double count,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                  float* __restrict__ dalpha, float* __restrict__ coef, int accumulate) {
  const int c = blockIdx.x * FIN_CPB + (threadIdx.x & (FIN_CPB - 1));
  const bool cok = c < C;
  double s[3];
  fin_reduce<3>(partial, rows, C, c, cok, s);
  if (threadIdx.x >= FIN_CPB || !cok) return;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s[0];
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s[1];
  if (dalpha) dalpha[c] = (accumulate ? dalpha[c] : 0.f) + (float)s[2];
  coef[c] = (float)(s[0] / count);
  coef[C + c] = (float)(s[1] / count);
}

// NEXT: the dx this kernel writes is the dy of ANOTHER BatchNorm without activation (the previous
// IBasicBlock's bn3, whose output was this BatchNorm's input); its backward sums
// (sum dx, sum dx * xhat_n, 0) are accumulated here from the stored dx and that BatchNorm's saved
// input nx -- one partial row [3][C] per workgroup, no separate reduce pass over dx.
template <typename T, bool NEXT, bool ADD_S2 = false>
__global__ void __launch_bounds__(256) k_bn_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift,
                                                      const float* __restrict__ alpha,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ invstd,
                                                      const float* __restrict__ coef,
                                                      const T* __restrict__ res, const T* __restrict__ add,
                                                      T* __restrict__ dx, T* __restrict__ dres, long n8,
                                                      int C8, const T* __restrict__ nx,
                                                      const float* __restrict__ nmean,
                                                      const float* __restrict__ ninvstd,
                                                      float* __restrict__ npartial, int aH, int aW) {
  const int C = C8 * 8;
  const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const int c0 = (int)(tid % C8) * 8;
  const Coef8 sc = ldc8(scale, c0, 1.f), sh = ldc8(shift, c0, 0.f), al = ldc8(alpha, c0, 1.f);
  const Coef8 mu = ldc8(mean, c0, 0.f), is = ldc8(invstd, c0, 1.f);
  const Coef8 k1 = ldc8(coef, c0, 0.f), k2 = ldc8(coef + C, c0, 0.f);
  // ADD_S2: `add` is the COMPACT input gradient of a 1x1 / stride-2 conv, [N][ceil(H/2)][ceil(W/2)][C]:
  // it lands on the pixels with even y and x, the other three quarters of the dense gradient are zero
  const float rcpW = ADD_S2 ? 1.0f / (float)aW : 0.f, rcpH = ADD_S2 ? 1.0f / (float)aH : 0.f;
  const int aPw = (aW + 1) >> 1, aPh = (aH + 1) >> 1;
  const Coef8 nmu = ldc8(NEXT ? nmean : nullptr, c0, 0.f), nis = ldc8(NEXT ? ninvstd : nullptr, c0, 1.f);
  float nq0[8], nq1[8];
#pragma unroll
  for (int j = 0; j < 8; j++) nq0[j] = nq1[j] = 0.f;
  for (long i = tid; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 g = load8<T>(dy + i * 8);
    Vec8 v = load8<T>(x + i * 8);
    Vec8 rr, ad;
    if (res) rr = load8<T>(res + i * 8);
    bool has_add = add != nullptr;
    if (ADD_S2) {
      // pixel -> (n, y, x) with float reciprocals (exact below 2^24 pixels, checked by the host) + fix-up
      const int pix = (int)(i / C8);
      int row = (int)((float)pix * rcpW), xx = pix - row * aW;
      if (xx < 0) { row--; xx += aW; } else if (xx >= aW) { row++; xx -= aW; }
      int nn = (int)((float)row * rcpH), yy = row - nn * aH;
      if (yy < 0) { nn--; yy += aH; } else if (yy >= aH) { nn++; yy -= aH; }
      has_add = !((xx | yy) & 1);
      if (has_add) ad = load8<T>(add + (((long)nn * aPh + (yy >> 1)) * aPw + (xx >> 1)) * C + c0);
    } else if (add) {
      ad = load8<T>(add + i * 8);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float gg = g.v[j];
      if (alpha) {
        float z = v.v[j] * sc.v[j] + sh.v[j];
        if (res) z += rr.v[j];
        if (z <= 0.f) gg *= al.v[j];
      }
      float xh = (v.v[j] - mu.v[j]) * is.v[j];
      // scale[c] == gamma * invstd
      v.v[j] = sc.v[j] * (gg - k1.v[j] - xh * k2.v[j]);
      if (has_add) v.v[j] += ad.v[j];
      g.v[j] = gg;
    }
    store8<T>(dx + i * 8, v);
    if (dres) store8<T>(dres + i * 8, g);
    if (NEXT) {
      const Vec8 o = round8<T>(v), xn = load8<T>(nx + i * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        nq0[j] += o.v[j];
        nq1[j] += o.v[j] * ((xn.v[j] - nmu.v[j]) * nis.v[j]);
      }
    }
  }
  if (NEXT) {
    __shared__ float red[2][256][9];
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      red[0][t][j] = nq0[j];
      red[1][t][j] = nq1[j];
    }
    __syncthreads();
    for (int i = t; i < 3 * C; i += 256) {
      const int q = i / C, c = i % C, cx = c >> 3, j = c & 7;
      float sum = 0.f;
      if (q < 2)
        for (int k = cx; k < 256; k += C8) sum += red[q][k][j];
      npartial[((long)blockIdx.x * 3 + q) * C + c] = sum;
    }
  }
}

extern "C" int msml_bn_act_bwd(const void* dy, const void* x, const float* scale, const float* shift,
                               const float* alpha, const float* save_mean, const float* save_invstd,
                               const void* residual_first, void* dx, void* dres, float* dgamma,
                               float* dbeta, float* dalpha, int accumulate, long M, int C,
                               float* workspace, long ws_floats, int dtype, void* stream) {
  MSML_CHECK(dy && x && dx && scale && shift && save_mean && save_invstd && workspace && M > 0 &&
                 C > 0 && C % 8 == 0 && C <= 2048,
             MSML_ERR_SHAPE, "bn_act_bwd: bad args M=%ld C=%d", M, C);
  int rows = red_rows(M, C);
  long need = (long)rows * 3 * C + 2 * C;
  MSML_CHECK(ws_floats >= need, MSML_ERR_WORKSPACE, "bn_act_bwd: workspace %ld < %ld floats", ws_floats, need);
  float* partial = workspace;
  float* coef = workspace + (long)rows * 3 * C;
  hipStream_t st = (hipStream_t)stream;
  long n8 = M * (C / 8);
  MSML_DISPATCH_DTYPE(
      dtype, "bn_act_bwd",
      if (rows <= RED_PIPE_MAX_ROWS)
        (k_bn_bwd_reduce<DT, true>)<<<rows, 256, red_lds(3, C), st>>>((const DT*)dy, (const DT*)x, scale, shift, alpha,
                                                                      save_mean, save_invstd,
                                                                      (const DT*)residual_first, M, C, partial, 0);
      else
        (k_bn_bwd_reduce<DT, false>)<<<rows, 256, red_lds(3, C), st>>>((const DT*)dy, (const DT*)x, scale, shift, alpha,
                                                                       save_mean, save_invstd,
                                                                       (const DT*)residual_first, M, C, partial, 0);
      MSML_LAUNCH_OK("bn_bwd_reduce");
      k_bn_bwd_finalize<<<cdiv(C, FIN_CPB), 1024, 0, st>>>(partial, rows, C, (double)M, dgamma, dbeta, dalpha, coef, accumulate);
      MSML_LAUNCH_OK("bn_bwd_finalize");
      (k_bn_bwd_apply<DT, false>)<<<ew_grid_c(n8, C / 8), 256, 0, st>>>(
          (const DT*)dy, (const DT*)x, scale, shift, alpha, save_mean, save_invstd, coef,
          (const DT*)residual_first, (const DT*)nullptr, (DT*)dx, (DT*)dres, n8, C / 8, (const DT*)nullptr,
          nullptr, nullptr, nullptr, 0, 0);)
  MSML_LAUNCH_OK("bn_bwd_apply");
  return MSML_OK;
}

// ------------------------------------------------------------------ backward, accumulator mode -------
// k_bn_bwd_finalize + k_bn_bwd_apply in ONE launch: the three backward sums arrive in acc[MSML_ACC_ROWS][3][C] (f64
// atomics of the producer: a backward-data conv's fused epilogue, the NEXT pass of another apply, or k_bn_bwd_reduce),
// every workgroup folds them and keeps k1 = s0 / n, k2 = s1 / n of all channels in LDS, workgroup 0 writes the
// parameter gradients.  NEXT / ADD_S2 as in k_bn_bwd_apply; NEXT adds into nacc (zero-initialised, same format).
// NACT (round 6): the NEXT BatchNorm is followed by a PReLU (the stems: conv -> bn -> prelu, iresnet.py:209-211,
// unet.py:193-195, whose output the first IBasicBlock's bn1 normalises): its three sums are formed through the PReLU
// mask of z = nx * nscale + nshift -- sum g', sum g' * xhat, sum dx * min(z, 0) with g' = dx * (z > 0 ? 1 : nalpha).
template <typename T, bool NEXT, bool ADD_S2, bool NACT = false>
__global__ void __launch_bounds__(256) k_bn_fin_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ alpha, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const double* __restrict__ acc,
                                                          double count, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, float* __restrict__ dalpha,
                                                          int accumulate, const T* __restrict__ res,
                                                          const T* __restrict__ add, T* __restrict__ dx,
                                                          T* __restrict__ dres, long n8, int C8,
                                                          const T* __restrict__ nx, const float* __restrict__ nmean,
                                                          const float* __restrict__ ninvstd, double* __restrict__ nacc,
                                                          int aH, int aW, const float* __restrict__ nscale = nullptr,
                                                          const float* __restrict__ nshift = nullptr,
                                                          const float* __restrict__ nalpha = nullptr) {
  static_assert(!NACT || NEXT, "NACT qualifies the NEXT BatchNorm");
  extern __shared__ float ck[];                        // [2][C]: k1, k2
  const int C = C8 * 8, t = threadIdx.x;
  MSML_LDS_REGION(ck, 2 * C * 4);
  for (int c = t; c < C; c += 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < MSML_ACC_ROWS; r++) {
      s0 += acc[(r * 3 + 0) * C + c];
      s1 += acc[(r * 3 + 1) * C + c];
      s2 += acc[(r * 3 + 2) * C + c];
    }
    ck[c] = (float)(s0 / count);
    ck[C + c] = (float)(s1 / count);
    if (blockIdx.x == 0) {
      if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s0;
      if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s1;
      if (dalpha) dalpha[c] = (accumulate ? dalpha[c] : 0.f) + (float)s2;
    }
  }
  __syncthreads();
  const long tid = blockIdx.x * (long)blockDim.x + t;
  const int c0 = (int)(tid % C8) * 8;
  const Coef8 sc = ldc8(scale, c0, 1.f), sh = ldc8(shift, c0, 0.f), al = ldc8(alpha, c0, 1.f);
  const Coef8 mu = ldc8(mean, c0, 0.f), is = ldc8(invstd, c0, 1.f);
  Coef8 k1, k2;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    k1.v[j] = ck[c0 + j];
    k2.v[j] = ck[C + c0 + j];
  }
  const float rcpW = ADD_S2 ? 1.0f / (float)aW : 0.f, rcpH = ADD_S2 ? 1.0f / (float)aH : 0.f;
  const int aPw = (aW + 1) >> 1, aPh = (aH + 1) >> 1;
  const Coef8 nmu = ldc8(NEXT ? nmean : nullptr, c0, 0.f), nis = ldc8(NEXT ? ninvstd : nullptr, c0, 1.f);
  const Coef8 nsc = ldc8(NACT ? nscale : nullptr, c0, 1.f), nsh = ldc8(NACT ? nshift : nullptr, c0, 0.f),
              nal = ldc8(NACT ? nalpha : nullptr, c0, 1.f);
  float nq0[8], nq1[8], nq2[NACT ? 8 : 1];
#pragma unroll
  for (int j = 0; j < 8; j++) nq0[j] = nq1[j] = 0.f;
#pragma unroll
  for (int j = 0; j < (NACT ? 8 : 1); j++) nq2[j] = 0.f;
  for (long i = tid; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 g = load8<T>(dy + i * 8);
    Vec8 v = load8<T>(x + i * 8);
    Vec8 rr, ad;
    if (res) rr = load8<T>(res + i * 8);
    bool has_add = add != nullptr;
    if (ADD_S2) {
      const int pix = (int)(i / C8);
      int row = (int)((float)pix * rcpW), xx = pix - row * aW;
      if (xx < 0) { row--; xx += aW; } else if (xx >= aW) { row++; xx -= aW; }
      int nn = (int)((float)row * rcpH), yy = row - nn * aH;
      if (yy < 0) { nn--; yy += aH; } else if (yy >= aH) { nn++; yy -= aH; }
      has_add = !((xx | yy) & 1);
      if (has_add) ad = load8<T>(add + (((long)nn * aPh + (yy >> 1)) * aPw + (xx >> 1)) * C + c0);
    } else if (add) {
      ad = load8<T>(add + i * 8);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float gg = g.v[j];
      if (alpha) {
        float z = v.v[j] * sc.v[j] + sh.v[j];
        if (res) z += rr.v[j];
        if (z <= 0.f) gg *= al.v[j];
      }
      float xh = (v.v[j] - mu.v[j]) * is.v[j];
      v.v[j] = sc.v[j] * (gg - k1.v[j] - xh * k2.v[j]);
      if (has_add) v.v[j] += ad.v[j];
      g.v[j] = gg;
    }
    store8<T>(dx + i * 8, v);
    if (dres) store8<T>(dres + i * 8, g);
    if (NEXT) {
      const Vec8 o = round8<T>(v), xn = load8<T>(nx + i * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float gn = o.v[j];
        if (NACT) {
          const float z = xn.v[j] * nsc.v[j] + nsh.v[j];
          if (z <= 0.f) {
            nq2[j] += gn * z;
            gn *= nal.v[j];
          }
        }
        nq0[j] += gn;
        nq1[j] += gn * ((xn.v[j] - nmu.v[j]) * nis.v[j]);
      }
    }
  }
  if (NEXT) {
    constexpr int NQN = NACT ? 3 : 2;
    __shared__ float red[NQN][256][9];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      red[0][t][j] = nq0[j];
      red[1][t][j] = nq1[j];
      if (NACT) red[NQN - 1][t][j] = nq2[j];
    }
    __syncthreads();
    for (int i = t; i < NQN * C; i += 256) {
      const int q = i / C, c = i % C, cx = c >> 3, j = c & 7;
      float sum = 0.f;
      for (int k = cx; k < 256; k += C8) sum += red[q][k][j];
      bnb_emit(reinterpret_cast<float*>(nacc), 1, blockIdx.x, q, C, c, sum);
    }
  }
}

// residual_first / dres: the PReLU-after-the-sum form of msml_bn_act_bwd (res_first blocks); add / add_h / add_w, next_*:
// as msml_bn_act_bwd_apply[_next][_s2].  acc: double[8][3][C] filled by the producer; next_acc: zero-initialised.
static int bn_fin_bwd_apply_impl(const void* dy, const void* x, const float* scale, const float* shift,
                                 const float* alpha, const float* save_mean, const float* save_invstd,
                                 const double* acc, const void* residual_first, const void* add, int add_h,
                                 int add_w, void* dx, void* dres, float* dgamma, float* dbeta, float* dalpha,
                                 int accumulate, long M, int C, const void* next_x, const float* next_mean,
                                 const float* next_invstd, double* next_acc, const float* next_scale,
                                 const float* next_shift, const float* next_alpha, int dtype, void* stream) {
  MSML_CHECK(dy && x && dx && scale && shift && save_mean && save_invstd && acc && M > 0 && C > 0 && C % 8 == 0,
             MSML_ERR_SHAPE, "bn_fin_bwd_apply: bad args M=%ld C=%d", M, C);
  MSML_CHECK(256 % (C / 8) == 0, MSML_ERR_UNSUPPORTED, "bn_fin_bwd_apply: C/8 = %d must divide 256", C / 8);
  const bool s2 = add_h > 0;
  MSML_CHECK(!s2 || (add && add_w > 0 && M % ((long)add_h * add_w) == 0 && M < (1L << 24)), MSML_ERR_SHAPE,
             "bn_fin_bwd_apply: stride-2 add needs M = N*H*W < 2^24 (M=%ld H=%d W=%d)", M, add_h, add_w);
  MSML_CHECK(!next_acc || (next_x && next_mean && next_invstd), MSML_ERR_SHAPE, "bn_fin_bwd_apply: next_* incomplete");
  hipStream_t st = (hipStream_t)stream;
  const long n8 = M * (C / 8);
  const size_t lds = (size_t)2 * C * sizeof(float);
#define BN_FIN_LAUNCH(NEXT_, S2_)                                                                                   \
  (k_bn_fin_bwd_apply<DT, NEXT_, S2_>)<<<ew_grid_c(n8, C / 8), 256, lds, st>>>(                                     \
      (const DT*)dy, (const DT*)x, scale, shift, alpha, save_mean, save_invstd, acc, (double)M, dgamma, dbeta, dalpha, \
      accumulate, (const DT*)residual_first, (const DT*)add, (DT*)dx, (DT*)dres, n8, C / 8, (const DT*)next_x,      \
      next_mean, next_invstd, next_acc, add_h, add_w);
#define BN_FIN_LAUNCH_ACT(S2_)                                                                                      \
  (k_bn_fin_bwd_apply<DT, true, S2_, true>)<<<ew_grid_c(n8, C / 8), 256, lds, st>>>(                                \
      (const DT*)dy, (const DT*)x, scale, shift, alpha, save_mean, save_invstd, acc, (double)M, dgamma, dbeta, dalpha, \
      accumulate, (const DT*)residual_first, (const DT*)add, (DT*)dx, (DT*)dres, n8, C / 8, (const DT*)next_x,      \
      next_mean, next_invstd, next_acc, add_h, add_w, next_scale, next_shift, next_alpha);
  const bool nact = next_acc && next_alpha;
  MSML_CHECK(!nact || (next_scale && next_shift), MSML_ERR_SHAPE, "bn_fin_bwd_apply: next_scale / next_shift missing");
#ifndef MSML_EXPERIMENTS     // (measured not faster, msml_amd/ops.py STEM_BWD_SUMS: instantiated in experiment builds only)
  MSML_CHECK(!nact, MSML_ERR_UNSUPPORTED, "bn_fin_bwd_apply_next_act: experiment builds only (msml_has_experiments)");
#undef BN_FIN_LAUNCH_ACT
#define BN_FIN_LAUNCH_ACT(S2_)
#endif
  MSML_DISPATCH_DTYPE(dtype, "bn_fin_bwd_apply",
                      if (nact) { if (s2) { BN_FIN_LAUNCH_ACT(true) } else { BN_FIN_LAUNCH_ACT(false) } }
                      else if (next_acc) { if (s2) { BN_FIN_LAUNCH(true, true) } else { BN_FIN_LAUNCH(true, false) } }
                      else { if (s2) { BN_FIN_LAUNCH(false, true) } else { BN_FIN_LAUNCH(false, false) } })
#undef BN_FIN_LAUNCH
#undef BN_FIN_LAUNCH_ACT
  MSML_LAUNCH_OK("bn_fin_bwd_apply");
  return MSML_OK;
}

extern "C" int msml_bn_fin_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift,
                                     const float* alpha, const float* save_mean, const float* save_invstd,
                                     const double* acc, const void* residual_first, const void* add, int add_h,
                                     int add_w, void* dx, void* dres, float* dgamma, float* dbeta, float* dalpha,
                                     int accumulate, long M, int C, const void* next_x, const float* next_mean,
                                     const float* next_invstd, double* next_acc, int dtype, void* stream) {
  return bn_fin_bwd_apply_impl(dy, x, scale, shift, alpha, save_mean, save_invstd, acc, residual_first, add, add_h, add_w, dx,
                               dres, dgamma, dbeta, dalpha, accumulate, M, C, next_x, next_mean, next_invstd, next_acc,
                               nullptr, nullptr, nullptr, dtype, stream);
}

// msml_bn_fin_bwd_apply whose NEXT BatchNorm is followed by a PReLU (next_alpha != null): the written dx is the gradient of
// PReLU(next_x * next_scale + next_shift), and next_acc receives all three sums of that BatchNorm + PReLU.
extern "C" int msml_bn_fin_bwd_apply_next_act(const void* dy, const void* x, const float* scale, const float* shift,
                                              const float* alpha, const float* save_mean, const float* save_invstd,
                                              const double* acc, const void* residual_first, const void* add, int add_h,
                                              int add_w, void* dx, void* dres, float* dgamma, float* dbeta, float* dalpha,
                                              int accumulate, long M, int C, const void* next_x, const float* next_scale,
                                              const float* next_shift, const float* next_alpha, const float* next_mean,
                                              const float* next_invstd, double* next_acc, int dtype, void* stream) {
  MSML_CHECK(next_x && next_scale && next_shift && next_alpha && next_mean && next_invstd && next_acc, MSML_ERR_SHAPE,
             "bn_fin_bwd_apply_next_act: next_* incomplete");
  return bn_fin_bwd_apply_impl(dy, x, scale, shift, alpha, save_mean, save_invstd, acc, residual_first, add, add_h, add_w, dx,
                               dres, dgamma, dbeta, dalpha, accumulate, M, C, next_x, next_mean, next_invstd, next_acc,
                               next_scale, next_shift, next_alpha, dtype, stream);
}

// msml_bn_act_bwd (no producer-side sums): reduce pass into the accumulator, then the fused finalize + apply
// (`add`, optional: another gradient path that joins at the BatchNorm input, summed into dx by the apply pass).
extern "C" int msml_bn_act_bwd_acc(const void* dy, const void* x, const float* scale, const float* shift,
                                   const float* alpha, const float* save_mean, const float* save_invstd,
                                   const void* residual_first, const void* add, void* dx, void* dres, float* dgamma,
                                   float* dbeta, float* dalpha, int accumulate, long M, int C, double* acc, int dtype,
                                   void* stream) {
  MSML_CHECK(dy && x && dx && scale && shift && save_mean && save_invstd && acc && M > 0 && C > 0 && C % 8 == 0 &&
                 C <= 2048, MSML_ERR_SHAPE, "bn_act_bwd_acc: bad args M=%ld C=%d", M, C);
  const int rows = red_rows(M, C);
  hipStream_t st = (hipStream_t)stream;
  MSML_DISPATCH_DTYPE(
      dtype, "bn_act_bwd_acc",
      if (rows <= RED_PIPE_MAX_ROWS)
        (k_bn_bwd_reduce<DT, true>)<<<rows, 256, red_lds(3, C), st>>>((const DT*)dy, (const DT*)x, scale, shift, alpha,
                                                                      save_mean, save_invstd, (const DT*)residual_first,
                                                                      M, C, reinterpret_cast<float*>(acc), 1);
      else
        (k_bn_bwd_reduce<DT, false>)<<<rows, 256, red_lds(3, C), st>>>((const DT*)dy, (const DT*)x, scale, shift, alpha,
                                                                       save_mean, save_invstd, (const DT*)residual_first,
                                                                       M, C, reinterpret_cast<float*>(acc), 1);)
  MSML_LAUNCH_OK("bn_bwd_reduce(acc)");
  return msml_bn_fin_bwd_apply(dy, x, scale, shift, alpha, save_mean, save_invstd, acc, residual_first, add, 0, 0,
                               dx, dres, dgamma, dbeta, dalpha, accumulate, M, C, nullptr, nullptr, nullptr, nullptr,
                               dtype, stream);
}

// Second half of msml_bn_act_bwd for callers that already hold the partial sums (a backward-data
// conv with the fused reduce, msml_conv2d_bnbwd): finalize + apply, with an optional tensor
// `add` summed into dx (the other gradient path that joins at the BatchNorm input).
static int bn_bwd_apply_impl(const void* dy, const void* x, const float* scale, const float* shift,
                             const float* alpha, const float* save_mean, const float* save_invstd,
                             const float* partial, int rows, const void* add, void* dx, float* dgamma,
                             float* dbeta, float* dalpha, int accumulate, long M, int C, float* coef_ws,
                             const void* next_x, const float* next_mean, const float* next_invstd,
                             float* next_partial, int dtype, void* stream, int add_h = 0, int add_w = 0) {
  MSML_CHECK(dy && x && dx && scale && shift && save_mean && save_invstd && partial && coef_ws && rows > 0 &&
                 M > 0 && C > 0 && C % 8 == 0 && C <= 2048,
             MSML_ERR_SHAPE, "bn_act_bwd_apply: bad args M=%ld C=%d rows=%d", M, C, rows);
  const bool s2 = add_h > 0;
  MSML_CHECK(!s2 || (add && add_w > 0 && M % ((long)add_h * add_w) == 0 && M < (1L << 24)), MSML_ERR_SHAPE,
             "bn_act_bwd_apply: stride-2 add needs M = N*H*W < 2^24 (M=%ld H=%d W=%d)", M, add_h, add_w);
  hipStream_t st = (hipStream_t)stream;
  long n8 = M * (C / 8);
  if (rows > FOLD_MIN_ROWS) {
    float* folded = coef_ws + 2 * C;
    k_fold_rows<3><<<dim3(cdiv(C, FIN_CPB), FOLD_ROWS), 1024, 0, st>>>(partial, rows, C, folded);
    MSML_LAUNCH_OK("bn_bwd_fold");
    partial = folded;
    rows = FOLD_ROWS;
  }
  k_bn_bwd_finalize<<<cdiv(C, FIN_CPB), 1024, 0, st>>>(partial, rows, C, (double)M, dgamma, dbeta, dalpha, coef_ws, accumulate);
  MSML_LAUNCH_OK("bn_bwd_finalize");
#define BN_APPLY_LAUNCH(NEXT_, S2_, NX, NM, NI, NP)                                                          \
  (k_bn_bwd_apply<DT, NEXT_, S2_>)<<<ew_grid_c(n8, C / 8), 256, 0, st>>>(                                    \
      (const DT*)dy, (const DT*)x, scale, shift, alpha, save_mean, save_invstd, coef_ws, (const DT*)nullptr, \
      (const DT*)add, (DT*)dx, (DT*)nullptr, n8, C / 8, (const DT*)(NX), NM, NI, NP, add_h, add_w);
  if (next_partial) {
    MSML_CHECK(next_x && next_mean && next_invstd && 256 % (C / 8) == 0, MSML_ERR_SHAPE,
               "bn_act_bwd_apply_next: bad args C=%d", C);
    MSML_DISPATCH_DTYPE(dtype, "bn_act_bwd_apply_next",
                        if (s2) { BN_APPLY_LAUNCH(true, true, next_x, next_mean, next_invstd, next_partial) }
                        else { BN_APPLY_LAUNCH(true, false, next_x, next_mean, next_invstd, next_partial) })
  } else {
    MSML_DISPATCH_DTYPE(dtype, "bn_act_bwd_apply",
                        if (s2) { BN_APPLY_LAUNCH(false, true, nullptr, nullptr, nullptr, nullptr) }
                        else { BN_APPLY_LAUNCH(false, false, nullptr, nullptr, nullptr, nullptr) })
  }
#undef BN_APPLY_LAUNCH
  MSML_LAUNCH_OK("bn_bwd_apply");
  return MSML_OK;
}

// Second half of msml_bn_act_bwd for callers that already hold the partial sums (a backward-data
// conv with the fused reduce, msml_conv2d_bnbwd): finalize + apply, with an optional tensor
// `add` summed into dx (the other gradient path that joins at the BatchNorm input).
extern "C" int msml_bn_act_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift,
                                     const float* alpha, const float* save_mean, const float* save_invstd,
                                     const float* partial, int rows, const void* add, void* dx,
                                     float* dgamma, float* dbeta, float* dalpha, int accumulate, long M,
                                     int C, float* coef_ws, int dtype, void* stream) {
  return bn_bwd_apply_impl(dy, x, scale, shift, alpha, save_mean, save_invstd, partial, rows, add, dx, dgamma,
                           dbeta, dalpha, accumulate, M, C, coef_ws, nullptr, nullptr, nullptr, nullptr, dtype,
                           stream);
}

// ... and, in the same pass, the backward sums of the activation-free BatchNorm whose OUTPUT
// gradient this dx is (next_x = that BatchNorm's saved input): next_partial[rows][3][C] with
// rows = msml_bn_act_bwd_apply_rows(M, C), ready for msml_bn_act_bwd_apply of that BatchNorm.
extern "C" int msml_bn_act_bwd_apply_rows(long M, int C) { return ew_grid_c(M * (C / 8), C / 8); }

extern "C" int msml_bn_act_bwd_apply_next(const void* dy, const void* x, const float* scale, const float* shift,
                                          const float* alpha, const float* save_mean, const float* save_invstd,
                                          const float* partial, int rows, const void* add, void* dx,
                                          float* dgamma, float* dbeta, float* dalpha, int accumulate, long M,
                                          int C, float* coef_ws, const void* next_x, const float* next_mean,
                                          const float* next_invstd, float* next_partial, int dtype,
                                          void* stream) {
  MSML_CHECK(next_partial, MSML_ERR_SHAPE, "bn_act_bwd_apply_next: next_partial is null");
  return bn_bwd_apply_impl(dy, x, scale, shift, alpha, save_mean, save_invstd, partial, rows, add, dx, dgamma,
                           dbeta, dalpha, accumulate, M, C, coef_ws, next_x, next_mean, next_invstd,
                           next_partial, dtype, stream);
}

// Same two entry points with `add` given as the COMPACT input gradient of a 1x1 / stride-2 / pad-0
// conv over an H x W map (the downsample path of the first block of a stage,
// backbones/frb/iresnet.py:52-54,66): add[N][ceil(H/2)][ceil(W/2)][C] lands on the pixels with even
// (y, x); the dense gradient -- three quarters zeros -- is neither written nor re-read.
extern "C" int msml_bn_act_bwd_apply_s2(const void* dy, const void* x, const float* scale, const float* shift,
                                        const float* alpha, const float* save_mean, const float* save_invstd,
                                        const float* partial, int rows, const void* add, int H, int W, void* dx,
                                        float* dgamma, float* dbeta, float* dalpha, int accumulate, long M,
                                        int C, float* coef_ws, int dtype, void* stream) {
  MSML_CHECK(add && H > 0 && W > 0, MSML_ERR_SHAPE, "bn_act_bwd_apply_s2: bad add H=%d W=%d", H, W);
  return bn_bwd_apply_impl(dy, x, scale, shift, alpha, save_mean, save_invstd, partial, rows, add, dx, dgamma,
                           dbeta, dalpha, accumulate, M, C, coef_ws, nullptr, nullptr, nullptr, nullptr, dtype,
                           stream, H, W);
}

extern "C" int msml_bn_act_bwd_apply_next_s2(const void* dy, const void* x, const float* scale,
                                             const float* shift, const float* alpha, const float* save_mean,
                                             const float* save_invstd, const float* partial, int rows,
                                             const void* add, int H, int W, void* dx, float* dgamma, float* dbeta,
                                             float* dalpha, int accumulate, long M, int C, float* coef_ws,
                                             const void* next_x, const float* next_mean,
                                             const float* next_invstd, float* next_partial, int dtype,
                                             void* stream) {
  MSML_CHECK(add && H > 0 && W > 0 && next_partial, MSML_ERR_SHAPE, "bn_act_bwd_apply_next_s2: bad args");
  return bn_bwd_apply_impl(dy, x, scale, shift, alpha, save_mean, save_invstd, partial, rows, add, dx, dgamma,
                           dbeta, dalpha, accumulate, M, C, coef_ws, next_x, next_mean, next_invstd,
                           next_partial, dtype, stream, H, W);
}

// ------------------------------------------------------------------ bias gradient ------------
// db[c] = sum over pixels of dy[.., c]  (GCM convs carry a bias: backbones/osb/unet.py:23-30)
__global__ void __launch_bounds__(1024) k_colsum_finalize(const float* __restrict__ partial, int rows, int C,
                                                          int stride_q, float* __restrict__ out, int Creal,
                                                          int accumulate) {
  constexpr int CS_LANES = 32;                         // 32 row-lanes x 32 channels
  // rows of [stride_q][C]; quantity 0 is the column sum.  Same 32 x 32 layout as fin_reduce.
  __shared__ double red[CS_LANES][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double acc = 0.0;
  if (c < Creal)
    for (int r = ry; r < rows; r += CS_LANES) acc += (double)partial[(long)r * stride_q * C + c];
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < Creal) {
    double s = 0.0;
    for (int y = 0; y < CS_LANES; y++) s += red[y][cx];
    out[c] = (accumulate ? out[c] : 0.f) + (float)s;
  }
}

extern "C" int msml_bias_grad(const void* dy, long M, int Cp, int Creal, float* db, int accumulate,
                              float* workspace, long ws_floats, int dtype, void* stream) {
  MSML_CHECK(dy && db && workspace && M > 0 && Cp > 0 && Cp % 8 == 0 && Creal <= Cp, MSML_ERR_SHAPE,
             "bias_grad: bad args");
  int rows = red_rows(M, Cp);
  MSML_CHECK(ws_floats >= (long)rows * 2 * Cp, MSML_ERR_WORKSPACE, "bias_grad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  MSML_DISPATCH_DTYPE(dtype, "bias_grad",
                      if (rows <= RED_PIPE_MAX_ROWS)
                        (k_bn_stats<DT, true>)<<<rows, 256, red_lds(2, Cp), st>>>((const DT*)dy, M, Cp, workspace, 0);
                      else
                        (k_bn_stats<DT, false>)<<<rows, 256, red_lds(2, Cp), st>>>((const DT*)dy, M, Cp, workspace, 0);)
  MSML_LAUNCH_OK("bias_grad");
  k_colsum_finalize<<<cdiv(Creal, 32), 1024, 0, st>>>(workspace, rows, Cp, 2, db, Creal, accumulate);
  MSML_LAUNCH_OK("bias_grad_finalize");
  return MSML_OK;
}

// ------------------------------------------------------------------ plain element-wise -------
// out = a + b (residual joins in backward, GCM left+right branch sum: unet.py:37)
template <typename T>
__global__ void __launch_bounds__(256) k_add(const T* __restrict__ a, const T* __restrict__ b,
                                             T* __restrict__ o, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8;
       i += (long)gridDim.x * blockDim.x) {
    Vec8 x = load8<T>(a + i * 8), y = load8<T>(b + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) x.v[j] += y.v[j];
    store8<T>(o + i * 8, x);
  }
}

extern "C" int msml_add(const void* a, const void* b, void* out, long n, int dtype, void* stream) {
  MSML_CHECK(a && b && out && n > 0 && n % 8 == 0, MSML_ERR_SHAPE, "add: n=%ld", n);
  MSML_DISPATCH_DTYPE(dtype, "add",
                      k_add<DT><<<ew_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                          (const DT*)a, (const DT*)b, (DT*)out, n / 8);)
  MSML_LAUNCH_OK("add");
  return MSML_OK;
}
