// Shared device/host helpers for libmsml_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/msml_hip.h"

// BatchNorm statistics in ACCUMULATOR mode (msml_conv2d_acc / msml_bn_fin_act_fwd, bn.hip): the `stats` pointer of a
// producer is a zero-initialised double[MSML_ACC_ROWS][2][C] and every workgroup adds its per-channel (sum, sumsq)
// with f64 atomics into row (workgroup index % MSML_ACC_ROWS) instead of storing a partial row of its own.  The
// f64 sum of f32 partials is exact unless two partials differ by more than 2^29, so the result does not depend on
// the order of the adds (and equals the fixed-order f64 sum of the row format).  The mode travels from the C entry
// point to the launch sites in a thread-local (set and reset inside one call: the library stays reentrant).
#ifndef MSML_ACC_ROWS
#define MSML_ACC_ROWS 8
#endif
extern thread_local int msml_tl_stats_acc;
// Border-class bias (msml_conv2d_x3_border, conv_igemm.hip): `bias` of a 3x3 / stride-1 / pad-1 forward launch points at
// float[9][coutp], row (cy * 3 + cx) with cy / cx = 0 on the first row / column of the map, 2 on the last, 1 inside.
// Travels from the C entry point to the launch sites like msml_tl_stats_acc.
extern thread_local int msml_tl_bias9;
__device__ __forceinline__ int border_class(int y, int x, int H, int W) {
  return ((y == 0) ? 0 : (y == H - 1 ? 2 : 1)) * 3 + ((x == 0) ? 0 : (x == W - 1 ? 2 : 1));
}
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

void msml_set_error(const char* fmt, ...);

#define MSML_CHECK(cond, code, ...)          \
  do {                                       \
    if (!(cond)) {                           \
      msml_set_error(__VA_ARGS__);           \
      return (code);                         \
    }                                        \
  } while (0)

// call after every launch: reports launch-configuration errors without synchronising
#define MSML_LAUNCH_OK(name)                                                     \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      msml_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return MSML_ERR_LAUNCH;                                                    \
    }                                                                            \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- LDS high-water guard --------------------------------------------------------------------
// Every kernel with dynamic LDS derives its regions (tile rings, transpose tiles, reduction scratch, coefficient tables)
// from one `extern __shared__` base, and the launch site computes the byte count as a max() of hand-derived terms; a
// region that outgrows the allocation is silently dropped / zero-read by the hardware (round 3 shipped such a launch:
// conv_fast.hip, `rlds`).  In a -DMSML_LDS_GUARD build (tools/build_variant.py --all MSML_LDS_GUARD) every region
// declares its extent right where it is derived, and the kernel TRAPS when the region ends beyond the LDS the dispatch
// was given: group_segment_size of the AQL packet = static + dynamic bytes of THIS launch.  No cost in the product build.
#ifdef MSML_LDS_GUARD
__device__ __forceinline__ void msml_lds_region(const void* p, size_t bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned int have = reinterpret_cast<const unsigned int*>(__builtin_amdgcn_dispatch_ptr())[7];   // group_segment_size
  const unsigned int off = (unsigned int)(size_t)(__attribute__((address_space(3))) const char*)p;
  if ((size_t)off + bytes > (size_t)have) __builtin_trap();
#endif
}
#define MSML_LDS_REGION(p, bytes) msml_lds_region((p), (size_t)(bytes))
#else
#define MSML_LDS_REGION(p, bytes) ((void)0)
#endif

// ---- storage element helpers ------------------------------------------------------------
// Activations are stored either as f32 or bf16; arithmetic is always f32.
__device__ __forceinline__ float bf2f(unsigned short v) {
  return __uint_as_float(((unsigned int)v) << 16);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
  // plain cast -> v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// 8 consecutive channels of one pixel, as f32
struct Vec8 {
  float v[8];
};

template <typename T>
__device__ __forceinline__ Vec8 load8(const T* p);
template <>
__device__ __forceinline__ Vec8 load8<float>(const float* p) {
  Vec8 r;
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    r.v[i] = a[i];
    r.v[4 + i] = b[i];
  }
  return r;
}
template <>
__device__ __forceinline__ Vec8 load8<unsigned short>(const unsigned short* p) {
  Vec8 r;
  u32x4 a = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    r.v[2 * i] = __uint_as_float(a[i] << 16);
    r.v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u);
  }
  return r;
}

template <typename T>
__device__ __forceinline__ void store8(T* p, const Vec8& r);
template <>
__device__ __forceinline__ void store8<float>(float* p, const Vec8& r) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[i] = r.v[i];
    b[i] = r.v[4 + i];
  }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <>
__device__ __forceinline__ void store8<unsigned short>(unsigned short* p, const Vec8& r) {
  u32x4 a;
#pragma unroll
  for (int i = 0; i < 4; i++)
    a[i] = (unsigned int)f2bf(r.v[2 * i]) | ((unsigned int)f2bf(r.v[2 * i + 1]) << 16);
  *reinterpret_cast<u32x4*>(p) = a;
}

template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load1<unsigned short>(const unsigned short* p) { return bf2f(*p); }
template <typename T>
__device__ __forceinline__ void store1(T* p, float v);
template <>
__device__ __forceinline__ void store1<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store1<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }

// wave (64-lane) sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 32 lanes of each wave half, in every lane of the half: four DPP adds inside the 16-lane rows (quad
// swaps, then the half-row and row mirrors: VALU only) and ONE cross-row exchange through the LDS crossbar.  A
// __shfl_xor tree is five ds_bpermute per value; at 128 values per wave and 8-12 waves per CU that tree alone was a
// fixed ~15 us at the end of every statistics launch (the crossbar moves 128 B per clock).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_half_sum(float v) {
  v = dpp_add<0xB1>(v);                              // quad_perm [1, 0, 3, 2]
  v = dpp_add<0x4E>(v);                              // quad_perm [2, 3, 0, 1]
  v = dpp_add<0x141>(v);                             // row_half_mirror
  v = dpp_add<0x140>(v);                             // row_mirror
  return v + __shfl_xor(v, 16, 64);
}

// ... and of the three backward sums (rows [3][C])
__device__ __forceinline__ void bnb_emit(float* partial, int acc_mode, long wg, int q, int C, int col, float v) {
  if (acc_mode)
    unsafeAtomicAdd(reinterpret_cast<double*>(partial) + ((wg & (MSML_ACC_ROWS - 1)) * 3 + q) * C + col, (double)v);
  else
    partial[(wg * 3 + q) * C + col] = v;
}

// one per-channel partial of a producer workgroup: a row store, or an f64 atomic add in accumulator mode
__device__ __forceinline__ void stats_emit(float* stats, int acc_mode, long wg, int which, int C, int col, float v) {
  if (acc_mode)
    unsafeAtomicAdd(reinterpret_cast<double*>(stats) + ((wg & (MSML_ACC_ROWS - 1)) * 2 + which) * C + col, (double)v);
  else
    stats[(wg * 2 + which) * C + col] = v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// dtype dispatch: DT is unsigned short for bf16 storage, float for f32 storage
#define MSML_DISPATCH_DTYPE(dtype, name, ...)                       \
  if ((dtype) == MSML_F32) {                                        \
    typedef float DT;                                               \
    __VA_ARGS__                                                     \
  } else if ((dtype) == MSML_BF16) {                                \
    typedef unsigned short DT;                                      \
    __VA_ARGS__                                                     \
  } else {                                                          \
    msml_set_error("%s: unsupported dtype %d", name, (int)(dtype)); \
    return MSML_ERR_DTYPE;                                          \
  }


// value of a Vec8 after a round trip through storage type T
template <typename T>
__device__ __forceinline__ Vec8 round8(const Vec8& v);
template <>
__device__ __forceinline__ Vec8 round8<float>(const Vec8& v) { return v; }
template <>
__device__ __forceinline__ Vec8 round8<unsigned short>(const Vec8& v) {
  Vec8 r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.v[j] = bf2f(f2bf(v.v[j]));
  return r;
}

// ---- BatchNorm(+PReLU) applied to a conv operand while it sits in LDS ----------------------------
// The halo kernels (conv_halo / conv_ws / wgrad_halo) keep the input image in LDS for all 9 taps, so
// a training-mode BatchNorm in front of the conv can be applied there (each wave transforms the 16-B
// chunks it DMA'd itself, right after its own vmcnt wait) and the normalised activation is never
// written to HBM.  Zero-padding pixels are skipped: padding applies AFTER the BatchNorm.
// Arithmetic and rounding are those of k_bn_act_fwd (bn.hip): y = bf16(PReLU(x * scale + shift)).
struct BnIn {
  const float* scale;      // nullptr: no input transform
  const float* shift;
  const float* alpha;      // nullptr: no PReLU
  // Accumulator mode (acc != nullptr, conv_halo.hip only): the coefficients are DERIVED in the kernel's prologue from the
  // producer's f64 sums double[MSML_ACC_ROWS][2][C] exactly as k_bn_fin_act_fwd derives them (bn.hip), workgroup (0, 0)
  // writes coef_out = float[4][C] (scale, shift, mean, invstd) and updates the running statistics, and the normalised
  // image is also WRITTEN OUT (store, same NHWC shape as the input: the weight gradient reads it) -- the BatchNorm apply
  // launch in front of the conv disappears.
  const double* acc = nullptr;
  double count = 0.0;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* rmean = nullptr;
  float* rvar = nullptr;
  float momentum = 0.f, eps = 0.f;
  float* coef_out = nullptr;
  unsigned short* store = nullptr;
};
// tab: LDS table [3][C] (scale, shift, alpha) filled by bn_in_fill
__device__ __forceinline__ void bn_in_fill(const BnIn& f, float* tab, int c0, int C, int t, int nt) {
  for (int i = t; i < C; i += nt) {
    tab[i] = f.scale[c0 + i];
    tab[C + i] = f.shift[c0 + i];
    tab[2 * C + i] = f.alpha ? f.alpha[c0 + i] : 1.f;
  }
}
// Accumulator mode: the table from the producer's f64 sums (the arithmetic of k_bn_fin_act_fwd's prologue, bn.hip --
// bit-identical coefficients); `writer` (one workgroup of the launch) also stores them and updates the running statistics.
__device__ __forceinline__ void bn_in_fill_acc(const BnIn& f, float* tab, int C, int t, int nt, bool writer) {
  for (int c = t; c < C; c += nt) {
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int r = 0; r < MSML_ACC_ROWS; r++) {
      s += f.acc[(r * 2 + 0) * C + c];
      ss += f.acc[(r * 2 + 1) * C + c];
    }
    const double m = s / f.count;
    double var = ss / f.count - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m;
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    const float sc = g * invstd, sh = b - mean * g * invstd;
    tab[c] = sc;
    tab[C + c] = sh;
    tab[2 * C + c] = f.alpha ? f.alpha[c] : 1.f;
    if (writer) {
      f.coef_out[c] = sc;
      f.coef_out[C + c] = sh;
      f.coef_out[2 * C + c] = mean;
      f.coef_out[3 * C + c] = invstd;
      if (f.rmean) {
        const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
        f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * mean;
        f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unbiased;
      }
    }
  }
}
__device__ __forceinline__ void bn_in_chunk(char* lds16, const float* tab, int C, int ch, bool has_alpha) {
  // two halves of 4 channels, not unrolled: the callers sit at their register limit
#pragma unroll 1
  for (int hf = 0; hf < 2; hf++) {
    u32x2 raw = *reinterpret_cast<const u32x2*>(lds16 + hf * 8);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + ch + hf * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(tab + C + ch + hf * 4);
    const f32x4 al = *reinterpret_cast<const f32x4*>(tab + 2 * C + ch + hf * 4);
    float z[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float x = (j & 1) ? __uint_as_float(raw[j >> 1] & 0xffff0000u) : __uint_as_float(raw[j >> 1] << 16);
      float y = x * sc[j] + sh[j];
      if (has_alpha) y = y > 0.f ? y : y * al[j];
      z[j] = y;
    }
    raw[0] = (unsigned int)f2bf(z[0]) | ((unsigned int)f2bf(z[1]) << 16);
    raw[1] = (unsigned int)f2bf(z[2]) | ((unsigned int)f2bf(z[3]) << 16);
    *reinterpret_cast<u32x2*>(lds16 + hf * 8) = raw;
  }
}

// bn_in_chunk with the lane's coefficients in registers (the 16x16x32 halo conv: a lane transforms the SAME 8 channels in
// every chunk of a slab, so the table is read once per slab instead of once per chunk); same arithmetic and rounding.
__device__ __forceinline__ void bn_in_chunk_r(char* lds16, const f32x4 (&sc)[2], const f32x4 (&sh)[2], const f32x4 (&al)[2],
                                              bool has_alpha) {
#pragma unroll
  for (int hf = 0; hf < 2; hf++) {
    u32x2 raw = *reinterpret_cast<const u32x2*>(lds16 + hf * 8);
    float z[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float x = (j & 1) ? __uint_as_float(raw[j >> 1] & 0xffff0000u) : __uint_as_float(raw[j >> 1] << 16);
      float y = x * sc[hf][j] + sh[hf][j];
      if (has_alpha) y = y > 0.f ? y : y * al[hf][j];
      z[j] = y;
    }
    raw[0] = (unsigned int)f2bf(z[0]) | ((unsigned int)f2bf(z[1]) << 16);
    raw[1] = (unsigned int)f2bf(z[2]) | ((unsigned int)f2bf(z[3]) << 16);
    *reinterpret_cast<u32x2*>(lds16 + hf * 8) = raw;
  }
}

// ---- BatchNorm backward-reduce fused into a backward-data conv epilogue ---------------------
// The conv's output dX is the gradient dy of a training-mode BatchNorm(+PReLU) output; the conv
// epilogue accumulates that BatchNorm's backward sums from the bf16-rounded dX it stores and the
// saved BatchNorm input x (same NHWC shape as dX):  q0 = sum g, q1 = sum g * xhat,
// q2 = sum dy * min(z, 0)  with z = x * scale + shift, g = dy * prelu'(z), xhat = (x - mean) * invstd
// (same definitions as k_bn_bwd_reduce in bn.hip).  partial: [rows][3][C], one row per workgroup.
struct BnBwdFuse {
  const unsigned short* x;
  const float* scale;
  const float* shift;
  const float* alpha;      // nullptr: no PReLU
  const float* mean;
  const float* invstd;
  float* partial;
  int acc;                 // accumulator mode: partial is a zero-initialised double[MSML_ACC_ROWS][3][C]
};

// ---- BatchNorm BACKWARD applied to a backward-data conv's input in LDS (conv_halo.hip, accumulator mode) -----------
// The conv input dc = d(loss)/d(BatchNorm input) is computed from the BatchNorm's output gradient dy (the tensor the
// conv DMA-loads), its saved input x (loaded by the lane that transforms the chunk) and the three backward sums the
// producer accumulated (double[MSML_ACC_ROWS][3][C]: sum g, sum g * xhat, sum dy * min(z, 0)) -- the arithmetic of
// k_bn_fin_bwd_apply (bn.hip): g = dy * prelu'(z), dc = scale * (g - s0 / n - xhat * s1 / n); one workgroup adds the
// parameter gradients (dgamma, dbeta, dalpha) and the transformed image is written through to `store` for the weight
// gradient.  The separate backward-apply launch in front of the conv disappears.
struct BnBwdIn {
  const unsigned short* x = nullptr;     // nullptr: no backward input transform
  const float* scale = nullptr;
  const float* shift = nullptr;
  const float* alpha = nullptr;          // nullptr: no PReLU
  const float* mean = nullptr;
  const float* invstd = nullptr;
  const double* acc = nullptr;
  double count = 0.0;
  float* dgamma = nullptr;
  float* dbeta = nullptr;
  float* dalpha = nullptr;
  int accumulate = 0;
  unsigned short* store = nullptr;
};
// tab: LDS table [7][C]: scale, shift, alpha, mean, invstd, k1 = s0 / n, k2 = s1 / n
__device__ __forceinline__ void bnbin_fill_acc(const BnBwdIn& f, float* tab, int C, int t, int nt, bool writer) {
  for (int c = t; c < C; c += nt) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < MSML_ACC_ROWS; r++) {
      s0 += f.acc[(r * 3 + 0) * C + c];
      s1 += f.acc[(r * 3 + 1) * C + c];
      s2 += f.acc[(r * 3 + 2) * C + c];
    }
    tab[c] = f.scale[c];
    tab[C + c] = f.shift[c];
    tab[2 * C + c] = f.alpha ? f.alpha[c] : 1.f;
    tab[3 * C + c] = f.mean[c];
    tab[4 * C + c] = f.invstd[c];
    tab[5 * C + c] = (float)(s0 / f.count);
    tab[6 * C + c] = (float)(s1 / f.count);
    if (writer) {
      if (f.dbeta) f.dbeta[c] = (f.accumulate ? f.dbeta[c] : 0.f) + (float)s0;
      if (f.dgamma) f.dgamma[c] = (f.accumulate ? f.dgamma[c] : 0.f) + (float)s1;
      if (f.dalpha) f.dalpha[c] = (f.accumulate ? f.dalpha[c] : 0.f) + (float)s2;
    }
  }
}
// one 16-B chunk (8 channels from `ch`) of the LDS image holds dy; xraw = the same chunk of the saved BatchNorm input
__device__ __forceinline__ void bnbin_chunk(char* lds16, const u32x4& xraw, const float* tab, int C, int ch, bool has_alpha) {
#pragma unroll 1
  for (int hf = 0; hf < 2; hf++) {
    u32x2 raw = *reinterpret_cast<const u32x2*>(lds16 + hf * 8);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + ch + hf * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(tab + C + ch + hf * 4);
    const f32x4 al = *reinterpret_cast<const f32x4*>(tab + 2 * C + ch + hf * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(tab + 3 * C + ch + hf * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(tab + 4 * C + ch + hf * 4);
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(tab + 5 * C + ch + hf * 4);
    const f32x4 k2 = *reinterpret_cast<const f32x4*>(tab + 6 * C + ch + hf * 4);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned int gw = raw[j >> 1], xw = xraw[hf * 2 + (j >> 1)];
      float gg = (j & 1) ? __uint_as_float(gw & 0xffff0000u) : __uint_as_float(gw << 16);
      const float v = (j & 1) ? __uint_as_float(xw & 0xffff0000u) : __uint_as_float(xw << 16);
      if (has_alpha) {
        float z = v * sc[j] + sh[j];
        if (z <= 0.f) gg *= al[j];
      }
      float xh = (v - mu[j]) * is[j];
      o[j] = sc[j] * (gg - k1[j] - xh * k2[j]);
    }
    raw[0] = (unsigned int)f2bf(o[0]) | ((unsigned int)f2bf(o[1]) << 16);
    raw[1] = (unsigned int)f2bf(o[2]) | ((unsigned int)f2bf(o[3]) << 16);
    *reinterpret_cast<u32x2*>(lds16 + hf * 8) = raw;
  }
}

struct BnbCoef {
  float sc[8], sh[8], al[8], is[8], nm[8];     // nm = -mean * invstd: xhat = x * is + nm
};
__device__ __forceinline__ BnbCoef bnb_load_coef(const BnBwdFuse& f, int c0) {
  BnbCoef k;
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(f.scale + c0), s1 = *reinterpret_cast<const f32x4*>(f.scale + c0 + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(f.shift + c0), h1 = *reinterpret_cast<const f32x4*>(f.shift + c0 + 4);
  const f32x4 m0 = *reinterpret_cast<const f32x4*>(f.mean + c0), m1 = *reinterpret_cast<const f32x4*>(f.mean + c0 + 4);
  const f32x4 i0 = *reinterpret_cast<const f32x4*>(f.invstd + c0), i1 = *reinterpret_cast<const f32x4*>(f.invstd + c0 + 4);
  f32x4 a0 = {1.f, 1.f, 1.f, 1.f}, a1 = a0;
  if (f.alpha) {
    a0 = *reinterpret_cast<const f32x4*>(f.alpha + c0);
    a1 = *reinterpret_cast<const f32x4*>(f.alpha + c0 + 4);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    k.sc[j] = s0[j]; k.sc[4 + j] = s1[j];
    k.sh[j] = h0[j]; k.sh[4 + j] = h1[j];
    k.al[j] = a0[j]; k.al[4 + j] = a1[j];
    k.is[j] = i0[j]; k.is[4 + j] = i1[j];
    k.nm[j] = -m0[j] * i0[j]; k.nm[4 + j] = -m1[j] * i1[j];
  }
  return k;
}
// branch-free; has_alpha is wave-uniform
__device__ __forceinline__ void bnb_accum(const BnbCoef& k, bool has_alpha, const Vec8& dy, const Vec8& x,
                                          float (&q)[3][8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const float d = dy.v[j];
    const float z = x.v[j] * k.sc[j] + k.sh[j];
    const bool neg = has_alpha & (z <= 0.f);
    const float g = neg ? d * k.al[j] : d;
    const float xh = x.v[j] * k.is[j] + k.nm[j];
    q[0][j] += g;
    q[1][j] += g * xh;
    q[2][j] += neg ? d * z : 0.f;
  }
}
