// Forward of the OSB decoder's transposed convolutions (backbones/osb/unet.py:140-156, deconv2..5:
// ConvTranspose2d(36 -> 18, k 4, s 2, p 1) on cat(seg, gcm), both segments 18 channels stored as 32) for bf16 NHWC.
//
//     out[n, 2i + cy, 2j + cx, co] = sum over the 2 x 2 taps of parity class (cy, cx), both segments, ci of
//                                    in_seg[n, i + dy, j + dx, ci] * w[ci][co][r][s]
//     cy = 0: (r, dy) in {(1, 0), (3, -1)};  cy = 1: (r, dy) in {(0, +1), (2, 0)}   (same for cx / s / dx)
//
// HBM-bound (deconv5: 103 MB in, 205 MB out), but the im2col kernel walks the four parity classes as four launch
// slices that each gather the input once per tap from L2 (16 passes): 208 us at 56 x 56.  Here a persistent workgroup
// keeps all 16 taps of both segments in LDS (64 KB), loads a 14 x 16 tile of input positions with its one-pixel halo
// once (two 32-channel planes, LDS-DMA, double buffered) and produces the 28 x 32 output pixels of all four classes
// from it; D = W_frag x X_frag, so a lane pair stores the 32 channels of a pixel with v_permlane32_swap.
#include <stdlib.h>

#include <mutex>

#include "common.h"

typedef __attribute__((address_space(3))) void* lptr_t;

#define CD_OOB 0x78000000u

struct ConvD4Args {
  const unsigned short* in0; const unsigned short* in1; unsigned int in_bytes;     // [N][H][H][32] each
  const unsigned short* wp; unsigned int w_bytes;                                    // packed [32 rows][2 * 16 * 32]
  unsigned short* out;                                                               // [N][2H][2H][32]
  const float* bias;
  int N, H, ty, tx, ntiles;
};

__global__ void __launch_bounds__(512) k_deconv4_fwd(const ConvD4Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TH = 14, TW = 16, PW = TW + 2, XPIX = (TH + 2) * PW;       // 288 region pixels
  constexpr int XPL = 18 * 1024;                                             // one 32-channel plane of the region (288 x 64 B)
  constexpr int WPL = 16 * 32 * 64;                                          // one segment's weights: [tap][co][32 ch]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem;                                                           // [2 segments][16 taps][32 co][64 B]
  char* Xs = smem + 2 * WPL;                                                 // [2 stages][2 planes][288 px][64 B]
  MSML_LDS_REGION(Ws, 2 * WPL);
  MSML_LDS_REGION(Xs, 2 * 2 * XPL);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int H = p.H;

  __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in0, 0, (int)p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, (int)p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.w_bytes, 0x00020000);

  // 64-B rows: a 1-KB DMA block covers 16 rows; swizzle key counts groups of four rows (conv_line.hip)
  auto key = [&](int row) -> int { return (row >> 2) & 3; };
  const int lrow = lane >> 2, lslot = lane & 3;

  // weights: packed row co holds K = [segment][tap][32 ch]; LDS rows = (seg * 16 + tap) * 32 + co
  for (int blk = wave; blk < 2 * 16 * 32 / 16; blk += 8) {
    const int row = blk * 16 + lrow;
    const int st = row >> 5, co = row & 31;                                  // st = seg * 16 + tap
    const unsigned int off = (unsigned int)(co * 1024 + st * 32 + ((lslot ^ key(row)) * 8)) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(Ws + blk * 1024), 16, off, 0, 0, 0);
  }

  auto issue = [&](int tile, int buf) {
    const int per = p.ty * p.tx;
    const int n = tile / per, rem = tile - n * per, tyi = rem / p.tx;
    const int i0 = tyi * TH, j0 = (rem - tyi * p.tx) * TW;
    char* xb = Xs + buf * (2 * XPL);
    for (int blk = wave; blk < 2 * 18; blk += 8) {                          // 18 blocks of 16 region pixels per plane
      const int pl = blk / 18, j = blk - pl * 18;
      const int q = j * 16 + lrow;
      const int ri = q / PW, rj = q - ri * PW;
      const int iy = i0 + ri - 1, ix = j0 + rj - 1;
      const bool ok = (q < XPIX) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)H);
      const unsigned int off = ok ? (unsigned int)(((n * H + iy) * H + ix) * 32 + ((lslot ^ key(q)) * 8)) * 2u : CD_OOB;
      if (pl == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(xb + blk * 1024), 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(xb + blk * 1024), 16, off, 0, 0, 0);
    }
  };

  const int r32 = lane & 31, h = lane >> 5;
  const int cls = wave & 3, cy = cls >> 1, cx = cls & 1;                    // this wave's output parity class
  // the class's 2 x 2 taps: (r, dy) / (s, dx)
  const int r_a = cy ? 0 : 1, dy_a = cy ? 1 : 0, r_b = cy ? 2 : 3, dy_b = cy ? 0 : -1;
  const int s_a = cx ? 0 : 1, dx_a = cx ? 1 : 0, s_b = cx ? 2 : 3, dx_b = cx ? 0 : -1;
  const int tap_i[4] = {r_a * 4 + s_a, r_a * 4 + s_b, r_b * 4 + s_a, r_b * 4 + s_b};
  const int shift[4] = {dy_a * PW + dx_a, dy_a * PW + dx_b, dy_b * PW + dx_a, dy_b * PW + dx_b};
  float bv[4][4];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int j = 0; j < 4; j++) bv[g][j] = p.bias ? p.bias[8 * g + 4 * h + j] : 0.f;

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue(tile, 0);
  __syncthreads();
  int cur = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    if (nxt < p.ntiles) issue(nxt, cur ^ 1);
    const char* xb = Xs + cur * (2 * XPL);
    const int per = p.ty * p.tx;
    const int n = tile / per, rem = tile - n * per, tyi = rem / p.tx;
    const int i0 = tyi * TH, j0 = (rem - tyi * p.tx) * TW;
    // 7 blocks of 2 rows x 16 columns of input positions; waves 0-3 take blocks 0, 2, 4, 6, waves 4-7 blocks 1, 3, 5
    for (int blk = wave >> 2; blk < 7; blk += 2) {
      const int ri = 2 * blk + (r32 >> 4), rj = r32 & 15;                   // tile position of this lane's pixel
      const int q0 = (ri + 1) * PW + rj + 1;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
      for (int ti = 0; ti < 4; ti++) {
        const int q = q0 + shift[ti], kq = key(q);
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
          for (int kk = 0; kk < 2; kk++) {
            const u32x4 bfrag = *reinterpret_cast<const u32x4*>(xb + pl * XPL + q * 64 + (((kk * 2 + h) ^ kq) << 4));
            const int row = (pl * 16 + tap_i[ti]) * 32 + r32;
            const u32x4 afrag = *reinterpret_cast<const u32x4*>(Ws + row * 64 + (((kk * 2 + h) ^ key(row)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfrag),
                                                          acc, 0, 0, 0);
          }
      }
      const int iy = i0 + ri, ix = j0 + rj;
      const bool valid = (iy < H) & (ix < H);
      unsigned short* o = p.out + ((long)(n * 2 * H + 2 * iy + cy) * (2 * H) + 2 * ix + cx) * 32 + 8 * h;
      u32x2 pk[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = acc[g * 4 + j] + bv[g][j];
        pk[g][0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
        pk[g][1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
      }
      u32x4 lo, hi;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        auto r01 = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
        auto r23 = __builtin_amdgcn_permlane32_swap(pk[2][e], pk[3][e], false, false);
        lo[e] = r01[0]; lo[2 + e] = r01[1];
        hi[e] = r23[0]; hi[2 + e] = r23[1];
      }
      if (valid) {
        *reinterpret_cast<u32x4*>(o) = lo;
        *reinterpret_cast<u32x4*>(o + 16) = hi;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
#endif
}

bool msml_deconv4_applies(int c0p, int c1p, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                          int pad_h, int pad_w, int transposed) {
  static const bool off = getenv("MSML_NO_D4_CONV") != nullptr;
  if (off || !transposed || R != 4 || S != 4 || stride != 2 || pad_h != 1 || pad_w != 1) return false;
  if (c0p != 32 || c1p != 32 || coutp != 32 || H != W || P != 2 * H || Q != 2 * W) return false;
  if (!(H == 56 || H == 28 || H == 14)) return false;
  if ((long)N * H * W * 64 >= 0x70000000L) return false;
  return true;
}

bool msml_deconv4_dispatch(const void* in0, const void* in1, const void* wp, int kop, const float* bias, void* out, int N,
                           int H, hipStream_t st) {
  ConvD4Args a;
  a.in0 = (const unsigned short*)in0; a.in1 = (const unsigned short*)in1;
  a.in_bytes = (unsigned int)((long)N * H * H * 64);
  a.wp = (const unsigned short*)wp; a.w_bytes = (unsigned int)((long)kop * 1024 * 2);
  a.out = (unsigned short*)out; a.bias = bias;
  a.N = N; a.H = H; a.ty = cdiv(H, 14); a.tx = cdiv(H, 16);
  a.ntiles = N * a.ty * a.tx;
  const size_t lds = 2 * 16 * 32 * 64 + 2 * 2 * 18 * 1024;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deconv4_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  });
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int grid = cus < a.ntiles ? cus : a.ntiles;
  k_deconv4_fwd<<<dim3(grid), dim3(512), lds, st>>>(a);
  return true;
}


// ------------------------------------------------------------------------------------------------------------
// Backward-data of the same layers: dX_seg[n, i, j, ci] = sum_{r, s, co} dY[n, 2i - 1 + r, 2j - 1 + s, co] * w[seg, ci][co][r][s]
// for BOTH input segments from one pass over dY (the autograd graph used two strided-conv launches, each gathering
// the 205 MB dY of deconv5 once per tap: 2 x 100 us).  An 8 x 16 tile of output positions needs an 18 x 34 region of
// dY (one 32-channel plane, 38 KB) which goes to LDS once; wave = (32-pixel block, segment), 16 taps x 2 k-steps.
struct ConvD4BwdArgs {
  const unsigned short* dy; unsigned int dy_bytes;       // [N][2H][2H][32]
  const unsigned short* wp0; const unsigned short* wp1; unsigned int w_bytes;     // [32 ci rows][16 taps * 32 co] each
  unsigned short* dx0; unsigned short* dx1;              // [N][H][H][32] each
  int N, H, ty, tx, ntiles;
};

__global__ void __launch_bounds__(512) k_deconv4_bwd(const ConvD4BwdArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TH = 8, TW = 16, RW = 2 * TW + 2, RPIX = (2 * TH + 2) * RW;          // 18 x 34 = 612 region pixels
  constexpr int XBLK = (RPIX + 15) / 16, XST = XBLK * 1024;                          // 39 KB per stage
  constexpr int WPL = 16 * 32 * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem;                                       // [2 segments][16 taps][32 ci][64 B of co]
  char* Xs = smem + 2 * WPL;                             // [2 stages][RPIX][64 B]
  MSML_LDS_REGION(Ws, 2 * WPL);
  MSML_LDS_REGION(Xs, 2 * XST);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int H = p.H, H2 = 2 * p.H;
  __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp0, 0, (int)p.w_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp1, 0, (int)p.w_bytes, 0x00020000);
  auto key = [&](int row) -> int { return (row >> 2) & 3; };
  const int lrow = lane >> 2, lslot = lane & 3;
  for (int blk = wave; blk < 2 * 16 * 32 / 16; blk += 8) {
    const int row = blk * 16 + lrow;                     // (seg * 16 + tap) * 32 + ci
    const int seg = row >> 9, tap = (row >> 5) & 15, ci = row & 31;
    const unsigned int off = (unsigned int)(ci * 512 + tap * 32 + ((lslot ^ key(row)) * 8)) * 2u;
    if (seg == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw0, (lptr_t)(Ws + blk * 1024), 16, off, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw1, (lptr_t)(Ws + blk * 1024), 16, off, 0, 0, 0);
  }
  auto issue = [&](int tile, int buf) {
    const int per = p.ty * p.tx;
    const int n = tile / per, rem = tile - n * per, tyi = rem / p.tx;
    const int i0 = tyi * TH, j0 = (rem - tyi * p.tx) * TW;
    char* xb = Xs + buf * XST;
    for (int blk = wave; blk < XBLK; blk += 8) {
      const int q = blk * 16 + lrow;
      const int rr = q / RW, cc = q - rr * RW;
      const int y = 2 * i0 - 1 + rr, x = 2 * j0 - 1 + cc;
      const bool ok = (q < RPIX) & ((unsigned)y < (unsigned)H2) & ((unsigned)x < (unsigned)H2);
      const unsigned int off = ok ? (unsigned int)(((n * H2 + y) * H2 + x) * 32 + ((lslot ^ key(q)) * 8)) * 2u : CD_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (lptr_t)(xb + blk * 1024), 16, off, 0, 0, 0);
    }
  };
  const int r32 = lane & 31, h = lane >> 5;
  const int blk4 = wave & 3, seg = wave >> 2;            // this wave's 32-pixel block (2 rows x 16) and segment
  const int ri = 2 * blk4 + (r32 >> 4), rj = r32 & 15;
  const int q0 = (2 * ri) * RW + 2 * rj;
  int tile = blockIdx.x;
  if (tile < p.ntiles) issue(tile, 0);
  __syncthreads();
  int cur = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    if (nxt < p.ntiles) issue(nxt, cur ^ 1);
    const char* xb = Xs + cur * XST;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
    for (int tp = 0; tp < 16; tp++) {
      const int q = q0 + (tp >> 2) * RW + (tp & 3), kq = key(q);
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const u32x4 bfrag = *reinterpret_cast<const u32x4*>(xb + q * 64 + (((kk * 2 + h) ^ kq) << 4));
        const int row = (seg * 16 + tp) * 32 + r32;
        const u32x4 afrag = *reinterpret_cast<const u32x4*>(Ws + row * 64 + (((kk * 2 + h) ^ key(row)) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfrag), acc,
                                                      0, 0, 0);
      }
    }
    const int per = p.ty * p.tx;
    const int n = tile / per, rem = tile - n * per, tyi = rem / p.tx;
    const int i = tyi * TH + ri, j = (rem - tyi * p.tx) * TW + rj;
    const bool valid = (i < H) & (j < H);
    unsigned short* o = (seg ? p.dx1 : p.dx0) + ((long)(n * H + i) * H + j) * 32 + 8 * h;
    u32x2 pk[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      pk[g][0] = (unsigned int)f2bf(acc[g * 4 + 0]) | ((unsigned int)f2bf(acc[g * 4 + 1]) << 16);
      pk[g][1] = (unsigned int)f2bf(acc[g * 4 + 2]) | ((unsigned int)f2bf(acc[g * 4 + 3]) << 16);
    }
    u32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      auto r01 = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
      auto r23 = __builtin_amdgcn_permlane32_swap(pk[2][e], pk[3][e], false, false);
      lo[e] = r01[0]; lo[2 + e] = r01[1];
      hi[e] = r23[0]; hi[2 + e] = r23[1];
    }
    if (valid) {
      *reinterpret_cast<u32x4*>(o) = lo;
      *reinterpret_cast<u32x4*>(o + 16) = hi;
    }
    __syncthreads();
    cur ^= 1;
  }
#endif
}

extern "C" int msml_deconv4_bwd_data(const void* dy, const void* wp0, const void* wp1, void* dx0, void* dx1, int N, int H,
                                     void* stream) {
  static const bool off = getenv("MSML_NO_D4_CONV") != nullptr;
  MSML_CHECK(dy && wp0 && wp1 && dx0 && dx1 && N > 0, MSML_ERR_SHAPE, "deconv4_bwd_data: bad arguments");
  if (off || !(H == 56 || H == 28 || H == 14) || (long)N * 4 * H * H * 64 >= 0x70000000L) {
    msml_set_error("deconv4_bwd_data: shape not covered (H=%d)", H);
    return MSML_ERR_UNSUPPORTED;
  }
  ConvD4BwdArgs a;
  a.dy = (const unsigned short*)dy; a.dy_bytes = (unsigned int)((long)N * 4 * H * H * 64);
  a.wp0 = (const unsigned short*)wp0; a.wp1 = (const unsigned short*)wp1; a.w_bytes = 32u * 512u * 2u;
  a.dx0 = (unsigned short*)dx0; a.dx1 = (unsigned short*)dx1;
  a.N = N; a.H = H; a.ty = cdiv(H, 8); a.tx = cdiv(H, 16);
  a.ntiles = N * a.ty * a.tx;
  const size_t lds = 2 * 16 * 32 * 64 + 2 * (size_t)((18 * 34 + 15) / 16) * 1024;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deconv4_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  });
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = cus < a.ntiles ? cus : a.ntiles;
  k_deconv4_bwd<<<dim3(grid), dim3(512), lds, (hipStream_t)stream>>>(a);
  MSML_LAUNCH_OK("deconv4_bwd_data");
  return MSML_OK;
}
