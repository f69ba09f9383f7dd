// Weight gradient for NARROW operands (32 stored channels on both sides): the 4x4 / stride-2 transposed
// convolutions of the OSB decoder on its 18-channel maps (backbones/osb/unet.py:140-156, deconv2..5: u = X on
// the P x Q grid, v = dY on the 2P x 2Q grid) and the 3x3 / stride-1 convs of the 32-channel FM bottlenecks
// (backbones/fm/fmoperator.py:43-44).  Same math and slab format as wgrad_fast.hip / wgrad_halo.hip:
//     dW[a][tap][b] = sum_pixels u[p][a] * v[stride * p - 1 + tap][b].
//
// These layers are HBM-bound (deconv5: 256 MB of operands for 26 GFLOP), but the im2col kernel gathers v once
// PER TAP from L2 -- 16 passes over the 205 MB dY of deconv5: 356 us per launch, 23 TFLOP/s.  Here a
// workgroup walks over strips of 4 x 16 u-pixels; the u strip and the v region WITH its halo go to LDS once
// (LDS-DMA, double-buffered) and serve all 16 (9) taps.  For stride 2 the v region is stored as two column-
// parity planes (x odd / x even), so that the 16 consecutive u-pixels of a k-step read 16 CONSECUTIVE plane
// pixels for every tap (x = 2 px - 1 + s: plane s & 1, shift s >> 1) and the transposing fragment read of
// wgrad_fast.hip (ds_read_b64_tr_b16 on [px][64 B] rows) applies unchanged.  8 waves share the taps
// (wave w: taps w, w + 8); the MFMA work is ~5 % of the fill time.
#include <stdlib.h>

#include <mutex>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) void* lptr_t;

#define WN_OOB 0x78000000u

struct WgradN32Args {
  const unsigned short* u; unsigned int u_bytes;     // [N][P][Q][32]
  const unsigned short* v; unsigned int v_bytes;     // [N][H][W][32]
  int N, P, Q, H, W, spy, spx;
  int nstrips, chunk;
  float* ws;                                         // [split][32][taps][32]
};

template <int R, int STRIDE>
__global__ void __launch_bounds__(512) k_wgrad_n32(const WgradN32Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TAPS = R * R, SR = 4, VR = STRIDE * SR + R - STRIDE, NPL = STRIDE;
  constexpr int UB = SR * 1024, VB = VR * NPL * 2048, STAGE = UB + VB + 1024;    // + 1 KB that absorbs padding DMAs
  constexpr int NBLK = SR + VR * NPL * 2, NISS = (NBLK + 7) / 8;
  static_assert(TAPS <= 16, "two taps per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  MSML_LDS_REGION(smem, 3 * STAGE);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int split = blockIdx.x;
  const int s_begin = split * p.chunk;
  int s_end = s_begin + p.chunk;
  if (s_end > p.nstrips) s_end = p.nstrips;
  const int spi = p.spy * p.spx;

  __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, (int)p.u_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)p.v, 0, (int)p.v_bytes, 0x00020000);

  const int lp = lane >> 2, lc = lane & 3;           // DMA slot: pixel lane / 4, 16-B chunk lane % 4
  auto issue = [&](int strip, int buf) {
    const int n = strip / spi, rem = strip - n * spi, sy = rem / p.spx;
    const int y0 = sy * SR, x0 = (rem - sy * p.spx) * 16;
    char* ub = smem + buf * STAGE;
    char* vb = ub + UB;
#pragma unroll
    for (int i = 0; i < NISS; i++) {
      const int blk = wave + 8 * i;
      if (blk >= NBLK) {
        // every wave issues exactly NISS DMA instructions per stage, so that the counted vmcnt wait below
        // means the same thing in every wave: the surplus one writes zeros into the stage's spare KB
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lptr_t)(vb + VB), 16, WN_OOB, 0, 0, 0);
        continue;
      }
      if (blk < SR) {
        const int y = y0 + blk, x = x0 + lp;
        const bool ok = (y < p.P) & (x < p.Q);
        const unsigned int off = ok ? (unsigned int)((n * p.P + y) * p.Q + x) * 64u + (unsigned int)lc * 16u : WN_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lptr_t)(ub + blk * 1024), 16, off, 0, 0, 0);
      } else {
        const int bb = blk - SR, half = bb & 1, plane = (bb >> 1) % NPL, vrow = (bb >> 1) / NPL;
        const int jj = half * 16 + lp;
        int x, y;
        bool ok;
        if (STRIDE == 2) {
          x = 2 * x0 - 1 + 2 * jj + plane;           // plane 0: odd columns 2 x0 - 1 + 2 jj, plane 1: even 2 x0 + 2 jj
          y = 2 * y0 - 1 + vrow;
          ok = jj <= 16;
        } else {
          x = x0 - 1 + jj;
          y = y0 - 1 + vrow;
          ok = jj < 18;
        }
        ok = ok & ((unsigned)x < (unsigned)p.W) & ((unsigned)y < (unsigned)p.H);
        const unsigned int off = ok ? (unsigned int)((n * p.H + y) * p.W + x) * 64u + (unsigned int)lc * 16u : WN_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(vb + bb * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[k][e] = 0.f;

  // transposing fragment read (wgrad_fast.hip): the lane addresses pixel px = 8 (g4 >> 1) + q4 (the second
  // read 4 pixels on), 4 channels pp of a 16-channel half g4 & 1
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int px = 8 * (g4 >> 1) + q4;
  const int chan = (2 * (g4 & 1) + (pp >> 1)) * 16 + (pp & 1) * 8;
  const int aofs = px * 64 + chan;
  const int ntaps = (wave + 8 < TAPS) ? 2 : (wave < TAPS ? 1 : 0);
  int vofs[2], vrow0[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int tp = wave + 8 * k < TAPS ? wave + 8 * k : 0;
    const int r = tp / R, s = tp - r * R;
    const int plane = STRIDE == 2 ? (s & 1) : 0, shift = STRIDE == 2 ? (s >> 1) : s;
    vrow0[k] = r;
    vofs[k] = plane * 2048 + (px + shift) * 64 + chan;
  }
  typedef __attribute__((address_space(3))) s16x4* tr_ptr;
  auto tr2 = [&](const char* lo) -> s16x8 {
    s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(lo));
    s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(lo + 256));
    s16x8 o;
    o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = l[3];
    o[4] = h[0]; o[5] = h[1]; o[6] = h[2]; o[7] = h[3];
    return o;
  };

  // three-stage ring, two strips in flight: stage `strip` is awaited with a counted vmcnt (this wave's newest
  // NISS DMAs belong to stage strip + 1), the raw barrier then says (a) every wave's part of the stage has
  // landed and (b) every wave is done reading stage strip - 1, whose buffer the next issue refills
  if (s_begin < s_end) issue(s_begin, 0);
  if (s_begin + 1 < s_end) issue(s_begin + 1, 1);
  int cur = 0, nxt = 2;
  for (int strip = s_begin; strip < s_end; strip++) {
    if (strip + 1 < s_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NISS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (strip + 2 < s_end) issue(strip + 2, nxt);
    const char* ub = smem + cur * STAGE + aofs;
    const char* vb = smem + cur * STAGE + UB;
#pragma unroll
    for (int j = 0; j < SR; j++) {                   // k-step = u row j of the strip (16 pixels)
      const s16x8 fa = tr2(ub + j * 1024);
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (k < ntaps) {
          const s16x8 fb = tr2(vb + (STRIDE * j + vrow0[k]) * (NPL * 2048) + vofs[k]);
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                           acc[k], 0, 0, 0);
        }
    }
    cur = cur == 2 ? 0 : cur + 1;
    nxt = nxt == 2 ? 0 : nxt + 1;
  }

  // slab [split][a][tap][b] (k_wgrad_reduce sums the splits in a fixed order)
  const int h = lane >> 5, b = lane & 31;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    if (k >= ntaps) continue;
    const int tp = wave + 8 * k;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int a = (e & 3) + 8 * (e >> 2) + 4 * h;
      p.ws[(((long)split * 32 + a) * TAPS + tp) * 32 + b] = acc[k][e];
    }
  }
#endif
}

static int wn_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}

// splits for a shape (0 = shape not covered by this kernel)
int msml_wgrad_n32_splits(int up, int vp, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                          int pad_w) {
  static const bool off = getenv("MSML_NO_N32_WGRAD") != nullptr;
  if (off || up != 32 || vp != 32 || R != S || pad_h != 1 || pad_w != 1) return 0;
  const bool d4 = R == 4 && stride == 2 && H == 2 * P && W == 2 * Q;
  const bool c3 = R == 3 && stride == 1 && H == P && W == Q;
  if (!d4 && !c3) return 0;
  if (P < 14 || Q < 14) return 0;                    // tiny maps: the generic kernel is as good
  if ((long)N * H * W * 64 >= 0x70000000L) return 0;
  const long strips = (long)N * cdiv(P, 4) * cdiv(Q, 16);
  long splits = 2L * wn_cus();                       // two resident workgroups per CU where LDS allows (3x3)
  if (d4) splits = wn_cus();
  if (splits > strips) splits = strips;
  if (splits > 512) splits = 512;
  return (int)splits;
}

template <int R, int STRIDE>
static void wn_launch(const WgradN32Args& a, int splits, hipStream_t st) {
  constexpr int VR = STRIDE * 4 + R - STRIDE;
  const size_t lds = 3 * (4 * 1024 + VR * STRIDE * 2048 + 1024);
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_n32<R, STRIDE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  k_wgrad_n32<R, STRIDE><<<dim3(splits), dim3(512), lds, st>>>(a);
}

bool msml_wgrad_n32_launch(const void* u, const void* v, float* ws, int N, int H, int W, int P, int Q, int R,
                           int stride, int splits, hipStream_t st) {
  WgradN32Args a;
  a.u = (const unsigned short*)u; a.u_bytes = (unsigned int)((long)N * P * Q * 64);
  a.v = (const unsigned short*)v; a.v_bytes = (unsigned int)((long)N * H * W * 64);
  a.N = N; a.P = P; a.Q = Q; a.H = H; a.W = W;
  a.spy = cdiv(P, 4); a.spx = cdiv(Q, 16);
  a.nstrips = N * a.spy * a.spx;
  a.chunk = cdiv(a.nstrips, splits);
  a.ws = ws;
  if (R == 4 && stride == 2) wn_launch<4, 2>(a, splits, st);
  else if (R == 3 && stride == 1) wn_launch<3, 1>(a, splits, st);
  else return false;
  return true;
}


// ------------------------------------------------------------------------------------------------------------
// Weight gradient of the GCM line convs (backbones/osb/unet.py:16-38: 7x1 / 1x7, stride 1): u = dY (18 -> 32 stored
// channels), v = X (32 or 64 stored channels).  Same line-major tiles as conv_line.hip: a tile is a full line along the
// conv axis times B positions across it (224 pixels = 14 k-steps of 16), the v region is the tile plus three zero
// pixels at either end of every line, and tap t reads the region t * B pixels further on -- operands go to LDS once
// for all 7 taps (the im2col kernel gathered v once per tap: 104-113 us per launch at 56 x 56).  Wave w < 7 owns tap w.
struct WgradLineArgs {
  const unsigned short* u; unsigned int u_bytes;     // dY [N][L][L][32]
  const unsigned short* v; unsigned int v_bytes;     // X  [N][L][L][VP]
  int N, L, B, tiles_per_img, ntiles, chunk, vertical;
  float* ws;                                         // [split][32][7][VP]
};

template <int VP>
__global__ void __launch_bounds__(512) k_wgrad_line(const WgradLineArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NH = VP / 32, UBLK = 14, VBLK = 18;  // 224 u pixels, <= 288 region pixels per 32-channel half
  constexpr int UB = UBLK * 1024, VB = NH * VBLK * 1024, STAGE = UB + VB;
  constexpr int NBLK = UBLK + NH * VBLK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  MSML_LDS_REGION(smem, 2 * STAGE);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int split = blockIdx.x, L = p.L, B = p.B, nreg = (L + 6) * B;
  const int t_begin = split * p.chunk;
  int t_end = t_begin + p.chunk;
  if (t_end > p.ntiles) t_end = p.ntiles;

  __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, (int)p.u_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)p.v, 0, (int)p.v_bytes, 0x00020000);
  const int lp = lane >> 2, lc = lane & 3;
  auto issue = [&](int tile, int buf) {
    const int n = tile / p.tiles_per_img, b0 = (tile - n * p.tiles_per_img) * B;
    char* ub = smem + buf * STAGE;
    char* vb = ub + UB;
    for (int blk = wave; blk < NBLK; blk += 8) {
      if (blk < UBLK) {
        const int q = blk * 16 + lp, a = q / B, b = b0 + (q - a * B);
        const bool ok = b < L;
        const int y = p.vertical ? a : b, x = p.vertical ? b : a;
        const unsigned int off = ok ? (unsigned int)((n * L + y) * L + x) * 64u + (unsigned int)lc * 16u : WN_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lptr_t)(ub + blk * 1024), 16, off, 0, 0, 0);
      } else {
        const int bb = blk - UBLK, hf = bb / VBLK, j = bb - hf * VBLK;
        const int q = j * 16 + lp, a = q / B - 3, b = b0 + (q - (q / B) * B);
        const bool ok = (q < nreg) & ((unsigned)a < (unsigned)L) & (b < L);
        const int y = p.vertical ? a : b, x = p.vertical ? b : a;
        const unsigned int off = ok ? (unsigned int)((n * L + y) * L + x) * (unsigned int)(VP * 2) + (unsigned int)(hf * 64 + lc * 16)
                                    : WN_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(vb + bb * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  f32x16 acc[NH];
#pragma unroll
  for (int k = 0; k < NH; k++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[k][e] = 0.f;
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int px = 8 * (g4 >> 1) + q4;
  const int chan = (2 * (g4 & 1) + (pp >> 1)) * 16 + (pp & 1) * 8;
  const int aofs = px * 64 + chan;
  const int vofs = (wave < 7 ? wave : 0) * B * 64 + px * 64 + chan;      // tap = wave: region pixel p + tap * B
  typedef __attribute__((address_space(3))) s16x4* tr_ptr;
  auto tr2 = [&](const char* lo) -> s16x8 {
    s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(lo));
    s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(lo + 256));
    s16x8 o;
    o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = l[3];
    o[4] = h[0]; o[5] = h[1]; o[6] = h[2]; o[7] = h[3];
    return o;
  };

  if (t_begin < t_end) issue(t_begin, 0);
  __syncthreads();
  int cur = 0;
  for (int tile = t_begin; tile < t_end; tile++) {
    if (tile + 1 < t_end) issue(tile + 1, cur ^ 1);
    if (wave < 7) {
      const char* ub = smem + cur * STAGE + aofs;
      const char* vb = smem + cur * STAGE + UB + vofs;
#pragma unroll
      for (int j = 0; j < UBLK; j++) {               // k-step = 16 consecutive line-major pixels
        const s16x8 fa = tr2(ub + j * 1024);
#pragma unroll
        for (int k = 0; k < NH; k++) {
          const s16x8 fb = tr2(vb + k * (VBLK * 1024) + j * 1024);
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                           acc[k], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  if (wave < 7) {
    const int h = lane >> 5, b = lane & 31;
#pragma unroll
    for (int k = 0; k < NH; k++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int a = (e & 3) + 8 * (e >> 2) + 4 * h;
        p.ws[(((long)split * 32 + a) * 7 + wave) * VP + k * 32 + b] = acc[k][e];
      }
  }
#endif
}

int msml_wgrad_line_splits(int up, int vp, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                           int pad_w) {
  static const bool off = getenv("MSML_NO_N32_WGRAD") != nullptr;
  if (off || up != 32 || (vp != 32 && vp != 64) || stride != 1) return 0;
  if (!((R == 7 && S == 1 && pad_h == 3 && pad_w == 0) || (R == 1 && S == 7 && pad_h == 0 && pad_w == 3))) return 0;
  if (H != W || P != H || Q != W || !(H == 56 || H == 28)) return 0;
  if ((long)N * H * W * vp * 2 >= 0x70000000L) return 0;
  const long tiles = (long)N * cdiv(H, 224 / H);
  long splits = (vp == 32 ? 2L : 1L) * wn_cus();
  if (splits > tiles) splits = tiles;
  if (splits > 512) splits = 512;
  return (int)splits;
}

template <int VP>
static void wl_launch(const WgradLineArgs& a, int splits, hipStream_t st) {
  const size_t lds = 2 * (size_t)(14 + (VP / 32) * 18) * 1024;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_line<VP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  });
  k_wgrad_line<VP><<<dim3(splits), dim3(512), lds, st>>>(a);
}

bool msml_wgrad_line_launch(const void* u, const void* v, int vp, float* ws, int N, int H, int R, int splits,
                            hipStream_t st) {
  WgradLineArgs a;
  a.u = (const unsigned short*)u; a.u_bytes = (unsigned int)((long)N * H * H * 64);
  a.v = (const unsigned short*)v; a.v_bytes = (unsigned int)((long)N * H * H * vp * 2);
  a.N = N; a.L = H; a.B = 224 / H;
  a.tiles_per_img = cdiv(H, a.B);
  a.ntiles = N * a.tiles_per_img;
  a.chunk = cdiv(a.ntiles, splits);
  a.vertical = R == 7 ? 1 : 0;
  a.ws = ws;
  if (vp == 64) wl_launch<64>(a, splits, st);
  else wl_launch<32>(a, splits, st);
  return true;
}
