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

// ------------------------------------------------------------------------------------------------
// Peer-guided branch of the FM operators (backbones/fm/fmoperator.py:293-308) and small element-wise
// helpers of the same family: the mask M = act(x) as a tensor (input of conv_m), products m_bar * yf,
// 1 - M ('invert'), the MSE distillation loss, and dropout (backbones/frb/iresnet.py:231).
template <typename T>
__global__ void __launch_bounds__(256) k_act_fwd(const T* __restrict__ x, T* __restrict__ m, long n8, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 a = load8<T>(x + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) a.v[j] = fm_act(a.v[j], act);
    store8<T>(m + i * 8, a);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) k_act_bwd(const T* __restrict__ dm, const T* __restrict__ x, T* __restrict__ dx,
                                                 long n8, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 g = load8<T>(dm + i * 8), a = load8<T>(x + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float m = fm_act(a.v[j], act);
      g.v[j] *= (act == MSML_ACT_SIGMOID) ? m * (1.f - m) : 1.f - m * m;
    }
    store8<T>(dx + i * 8, g);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) k_mul_fwd(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o,
                                                 long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 u = load8<T>(a + i * 8), v = load8<T>(b + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) u.v[j] *= v.v[j];
    store8<T>(o + i * 8, u);
  }
}
// da = g * b, db = g * a (either output may be null)
template <typename T>
__global__ void __launch_bounds__(256) k_mul_bwd(const T* __restrict__ g, const T* __restrict__ a, const T* __restrict__ b,
                                                 T* __restrict__ da, T* __restrict__ db, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 q = load8<T>(g + i * 8), u = load8<T>(a + i * 8), v = load8<T>(b + i * 8), x, y;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      x.v[j] = q.v[j] * v.v[j];
      y.v[j] = q.v[j] * u.v[j];
    }
    if (da) store8<T>(da + i * 8, x);
    if (db) store8<T>(db + i * 8, y);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) k_axpb(const T* __restrict__ x, T* __restrict__ y, long n8, float a, float b) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 u = load8<T>(x + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) u.v[j] = a * u.v[j] + b;
    store8<T>(y + i * 8, u);
  }
}

// sum (a - b)^2: one partial per block (fixed order inside the block), summed by k_mse_final.
template <typename T>
__global__ void __launch_bounds__(256) k_mse_partial(const T* __restrict__ a, const T* __restrict__ b, long n8,
                                                     double* __restrict__ part) {
  double s = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 u = load8<T>(a + i * 8), v = load8<T>(b + i * 8);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float d = u.v[j] - v.v[j];
      q += d * d;
    }
    s += (double)q;
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void k_mse_final(const double* __restrict__ part, int nb, double count, float* __restrict__ loss) {
  double s = 0.0;
  for (int i = 0; i < nb; i++) s += part[i];
  loss[0] = (float)(s / count);
}
// da = (2 / count) * g * (a - b), db = -da
template <typename T>
__global__ void __launch_bounds__(256) k_mse_bwd(const T* __restrict__ a, const T* __restrict__ b,
                                                 const float* __restrict__ g, float k, T* __restrict__ da,
                                                 T* __restrict__ db, long n8) {
  const float c = k * g[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 u = load8<T>(a + i * 8), v = load8<T>(b + i * 8), x, y;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      x.v[j] = c * (u.v[j] - v.v[j]);
      y.v[j] = -x.v[j];
    }
    if (da) store8<T>(da + i * 8, x);
    if (db) store8<T>(db + i * 8, y);
  }
}

// y = x * keep / (1 - p), keep decided per element by a counter-based hash of (seed, element index): the
// backward applies the same call to dy.
__device__ __forceinline__ unsigned int drop_hash(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + i * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return (unsigned int)((z ^ (z >> 31)) >> 32);
}
template <typename T>
__global__ void __launch_bounds__(256) k_dropout(const T* __restrict__ x, T* __restrict__ y, long n8, float p,
                                                 unsigned long long seed, const long* __restrict__ seed_ptr) {
  if (seed_ptr) seed = (unsigned long long)seed_ptr[0];     // device-resident seed: a graph replay draws a new mask
  const unsigned int thr = (unsigned int)((double)p * 4294967296.0);
  const float scale = 1.f / (1.f - p);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    Vec8 u = load8<T>(x + i * 8);
#pragma unroll
    for (int j = 0; j < 8; j++) u.v[j] = drop_hash(seed, (unsigned long long)(i * 8 + j)) >= thr ? u.v[j] * scale : 0.f;
    store8<T>(y + i * 8, u);
  }
}

#define EW_CHECK(name) MSML_CHECK(n > 0 && n % 8 == 0, MSML_ERR_SHAPE, name ": n=%ld must be a positive multiple of 8", n)

extern "C" int msml_fm_act_fwd(const void* x, void* m, long n, int act, int dtype, void* stream) {
  EW_CHECK("fm_act_fwd");
  MSML_DISPATCH_DTYPE(dtype, "fm_act_fwd",
                      (k_act_fwd<DT>)<<<fm_grid(n / 8), 256, 0, (hipStream_t)stream>>>((const DT*)x, (DT*)m, n / 8, act);)
  MSML_LAUNCH_OK("fm_act_fwd");
  return MSML_OK;
}
extern "C" int msml_fm_act_bwd(const void* dm, const void* x, void* dx, long n, int act, int dtype, void* stream) {
  EW_CHECK("fm_act_bwd");
  MSML_DISPATCH_DTYPE(dtype, "fm_act_bwd", (k_act_bwd<DT>)<<<fm_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                                               (const DT*)dm, (const DT*)x, (DT*)dx, n / 8, act);)
  MSML_LAUNCH_OK("fm_act_bwd");
  return MSML_OK;
}
extern "C" int msml_mul_fwd(const void* a, const void* b, void* out, long n, int dtype, void* stream) {
  EW_CHECK("mul_fwd");
  MSML_DISPATCH_DTYPE(dtype, "mul_fwd", (k_mul_fwd<DT>)<<<fm_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                                            (const DT*)a, (const DT*)b, (DT*)out, n / 8);)
  MSML_LAUNCH_OK("mul_fwd");
  return MSML_OK;
}
extern "C" int msml_mul_bwd(const void* g, const void* a, const void* b, void* da, void* db, long n, int dtype,
                            void* stream) {
  EW_CHECK("mul_bwd");
  MSML_DISPATCH_DTYPE(dtype, "mul_bwd", (k_mul_bwd<DT>)<<<fm_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                                            (const DT*)g, (const DT*)a, (const DT*)b, (DT*)da, (DT*)db, n / 8);)
  MSML_LAUNCH_OK("mul_bwd");
  return MSML_OK;
}
extern "C" int msml_axpb(const void* x, void* y, long n, float a, float b, int dtype, void* stream) {
  EW_CHECK("axpb");
  MSML_DISPATCH_DTYPE(dtype, "axpb",
                      (k_axpb<DT>)<<<fm_grid(n / 8), 256, 0, (hipStream_t)stream>>>((const DT*)x, (DT*)y, n / 8, a, b);)
  MSML_LAUNCH_OK("axpb");
  return MSML_OK;
}
extern "C" int msml_mse_fwd(const void* a, const void* b, long n, double count, float* loss, double* workspace,
                            long ws_doubles, int dtype, void* stream) {
  EW_CHECK("mse_fwd");
  const int nb = fm_grid(n / 8);
  MSML_CHECK(loss && workspace && ws_doubles >= nb && count > 0, MSML_ERR_WORKSPACE, "mse_fwd: workspace of %ld doubles, need %d",
             ws_doubles, nb);
  MSML_DISPATCH_DTYPE(dtype, "mse_fwd",
                      (k_mse_partial<DT>)<<<nb, 256, 0, (hipStream_t)stream>>>((const DT*)a, (const DT*)b, n / 8, workspace);)
  k_mse_final<<<1, 1, 0, (hipStream_t)stream>>>(workspace, nb, count, loss);
  MSML_LAUNCH_OK("mse_fwd");
  return MSML_OK;
}
extern "C" int msml_mse_bwd(const void* a, const void* b, const float* g, double count, void* da, void* db, long n,
                            int dtype, void* stream) {
  EW_CHECK("mse_bwd");
  MSML_DISPATCH_DTYPE(dtype, "mse_bwd", (k_mse_bwd<DT>)<<<fm_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                                            (const DT*)a, (const DT*)b, g, (float)(2.0 / count), (DT*)da, (DT*)db, n / 8);)
  MSML_LAUNCH_OK("mse_bwd");
  return MSML_OK;
}
extern "C" int msml_dropout(const void* x, void* y, long n, float p, long seed, int dtype, void* stream) {
  EW_CHECK("dropout");
  MSML_CHECK(p >= 0.f && p < 1.f, MSML_ERR_SHAPE, "dropout: p=%f", p);
  MSML_DISPATCH_DTYPE(dtype, "dropout", (k_dropout<DT>)<<<fm_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                                            (const DT*)x, (DT*)y, n / 8, p, (unsigned long long)seed, nullptr);)
  MSML_LAUNCH_OK("dropout");
  return MSML_OK;
}
extern "C" int msml_dropout_dev(const void* x, void* y, long n, float p, const long* seed, int dtype, void* stream) {
  EW_CHECK("dropout_dev");
  MSML_CHECK(seed && p >= 0.f && p < 1.f, MSML_ERR_SHAPE, "dropout_dev: p=%f", p);
  MSML_DISPATCH_DTYPE(dtype, "dropout_dev", (k_dropout<DT>)<<<fm_grid(n / 8), 256, 0, (hipStream_t)stream>>>(
                                                (const DT*)x, (DT*)y, n / 8, p, 0ULL, seed);)
  MSML_LAUNCH_OK("dropout_dev");
  return MSML_OK;
}
