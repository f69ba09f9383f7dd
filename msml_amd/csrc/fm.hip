// Feature-Masking fused element-wise kernels (backbones/fm/fmoperator.py:288,304-310).
//   M = act(x);  z = arith(yf, M) + yf
// HBM-bound: forward streams 3 tensors, backward 5.  16 B per lane per access, grid-stride.
#include "common.h"

__device__ __forceinline__ float fm_act(float x, int act) {
  if (act == MSML_ACT_SIGMOID) return 1.f / (1.f + __expf(-x));
  // tanh(x) = 1 - 2/(exp(2x)+1); exact at +-inf, no cancellation blow-up near 0 beyond 1 ulp of 1
  return tanhf(x);
}

template <typename T, int ACT, int ARITH>
__global__ void __launch_bounds__(256) k_fm_fwd(const T* __restrict__ x, const T* __restrict__ yf,
                                                T* __restrict__ z, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8;
       i += (long)gridDim.x * blockDim.x) {
    Vec8 a = load8<T>(x + i * 8);
    Vec8 f = load8<T>(yf + i * 8);
    Vec8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float m = fm_act(a.v[j], ACT);
      float y = f.v[j];
      float r;
      if (ARITH == MSML_ARITH_ADD) r = y + m;
      else if (ARITH == MSML_ARITH_SUB) r = y - m;
      else if (ARITH == MSML_ARITH_MUL) r = y * m;
      else r = y / m;
      o.v[j] = r + y;
    }
    store8<T>(z + i * 8, o);
  }
}

// dz -> (dx, dyf).  dM/dx: sigmoid' = M(1-M), tanh' = 1-M^2.
//   add: dyf = 2 dz,        dM = dz
//   sub: dyf = 2 dz,        dM = -dz
//   mul: dyf = dz (M + 1),  dM = dz yf
//   div: dyf = dz (1/M + 1), dM = -dz yf / M^2
template <typename T, int ACT, int ARITH>
__global__ void __launch_bounds__(256) k_fm_bwd(const T* __restrict__ dz, const T* __restrict__ x,
                                                const T* __restrict__ yf, T* __restrict__ dx,
                                                T* __restrict__ dyf, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8;
       i += (long)gridDim.x * blockDim.x) {
    Vec8 g = load8<T>(dz + i * 8);
    Vec8 a = load8<T>(x + i * 8);
    Vec8 f = load8<T>(yf + i * 8);
    Vec8 ox, of;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float m = fm_act(a.v[j], ACT);
      float dm_dx = (ACT == MSML_ACT_SIGMOID) ? m * (1.f - m) : 1.f - m * m;
      float y = f.v[j], d = g.v[j];
      float dM, dY;
      if (ARITH == MSML_ARITH_ADD) { dY = 2.f * d; dM = d; }
      else if (ARITH == MSML_ARITH_SUB) { dY = 2.f * d; dM = -d; }
      else if (ARITH == MSML_ARITH_MUL) { dY = d * (m + 1.f); dM = d * y; }
      else { float im = 1.f / m; dY = d * (im + 1.f); dM = -d * y * im * im; }
      ox.v[j] = dM * dm_dx;
      of.v[j] = dY;
    }
    store8<T>(dx + i * 8, ox);
    store8<T>(dyf + i * 8, of);
  }
}

static inline int fm_grid(long n8) {
  long b = (n8 + 255) / 256;
  return (int)(b < 2048 ? b : 2048);   // 256 CUs x 8 blocks, grid-stride beyond
}

#define FM_CASE(K, ACTV, ARV, ...)                                              \
  if (act == ACTV && arith == ARV) {                                            \
    K<DT, ACTV, ARV><<<fm_grid(n8), 256, 0, (hipStream_t)stream>>>(__VA_ARGS__); \
    launched = true;                                                            \
  }
#define FM_ALL(K, ...)                                        \
  FM_CASE(K, MSML_ACT_TANH, MSML_ARITH_ADD, __VA_ARGS__)      \
  FM_CASE(K, MSML_ACT_TANH, MSML_ARITH_SUB, __VA_ARGS__)      \
  FM_CASE(K, MSML_ACT_TANH, MSML_ARITH_MUL, __VA_ARGS__)      \
  FM_CASE(K, MSML_ACT_TANH, MSML_ARITH_DIV, __VA_ARGS__)      \
  FM_CASE(K, MSML_ACT_SIGMOID, MSML_ARITH_ADD, __VA_ARGS__)   \
  FM_CASE(K, MSML_ACT_SIGMOID, MSML_ARITH_SUB, __VA_ARGS__)   \
  FM_CASE(K, MSML_ACT_SIGMOID, MSML_ARITH_MUL, __VA_ARGS__)   \
  FM_CASE(K, MSML_ACT_SIGMOID, MSML_ARITH_DIV, __VA_ARGS__)

extern "C" int msml_fm_fuse_fwd(const void* x, const void* yf, void* z, long n, int act,
                                int arith, int dtype, void* stream) {
  MSML_CHECK(n > 0 && n % 8 == 0, MSML_ERR_SHAPE, "fm_fuse_fwd: n=%ld must be a positive multiple of 8", n);
  MSML_CHECK(act >= 0 && act <= 1 && arith >= 0 && arith <= 3, MSML_ERR_UNSUPPORTED,
             "fm_fuse_fwd: act=%d arith=%d", act, arith);
  long n8 = n / 8;
  bool launched = false;
  MSML_DISPATCH_DTYPE(dtype, "fm_fuse_fwd",
                      FM_ALL(k_fm_fwd, (const DT*)x, (const DT*)yf, (DT*)z, n8))
  (void)launched;
  MSML_LAUNCH_OK("fm_fuse_fwd");
  return MSML_OK;
}

extern "C" int msml_fm_fuse_bwd(const void* dz, const void* x, const void* yf, void* dx, void* dyf,
                                long n, int act, int arith, int dtype, void* stream) {
  MSML_CHECK(n > 0 && n % 8 == 0, MSML_ERR_SHAPE, "fm_fuse_bwd: n=%ld must be a positive multiple of 8", n);
  MSML_CHECK(act >= 0 && act <= 1 && arith >= 0 && arith <= 3, MSML_ERR_UNSUPPORTED,
             "fm_fuse_bwd: act=%d arith=%d", act, arith);
  long n8 = n / 8;
  bool launched = false;
  MSML_DISPATCH_DTYPE(dtype, "fm_fuse_bwd",
                      FM_ALL(k_fm_bwd, (const DT*)dz, (const DT*)x, (const DT*)yf, (DT*)dx,
                             (DT*)dyf, n8))
  (void)launched;
  MSML_LAUNCH_OK("fm_fuse_bwd");
  return MSML_OK;
}
