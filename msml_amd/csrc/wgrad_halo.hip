// Weight gradient of 3x3 / stride-1 / pad-1 convs for bf16 NHWC tensors with Cout % 128 == 0 and
// Cin % 64 == 0 (the 28x28 / 14x14 stages of IResNet and of the OSB encoder,
// backbones/frb/iresnet.py:40-67) -- same math and slab format as wgrad_fast.hip /
// conv_wgrad.hip:   dW[a][tap][b] = sum_pixels dY[p][a] * X[p + tap][b].
//
// wgrad_fast gathers X once per tap (im2col) and is bound by the L2 -> LDS fill rate like the
// im2col conv.  Here a workgroup owns a 128 (Cout) x 64 (Cin) block of dW for ALL 9 taps and walks
// over strips of 7 x 14 output pixels: the dY strip (7 rows x 16-pixel pitch, the 2 padding
// pixels of a row are zero) and the X strip WITH its halo (9 rows x 16) go to LDS once and serve
// the 9 taps -- 48 KB of fill per 16.5 MFLOP (the im2col kernel: 32 KB per 2.1 MFLOP).  A k-step is
// one strip row (16 pixels); tap (r, s) reads the X image r rows down and s pixels right, which
// is a per-lane offset in the blocked image [row][32-channel group][16 px][64 B] (the layout
// whose transposing ds_read_b64_tr_b16 blocks are conflict-free, see wgrad_fast.hip).
// 8 waves: wave = (pair of 32-row Cout tiles, 32-column Cin half, tap group {0-4} / {5-8}): 10 or 8
// accumulator tiles; per k-step 2 dY fragments + 5 (4) X fragments feed 10 (8) MFMAs -- 1.4 LDS
// fragment reads per MFMA (one Cout tile x 9 taps per wave needed 2.2 and was LDS-read bound).
// Waves w and w + 4 share a SIMD and carry one tap group each, so every SIMD runs 18 MFMAs per
// k-step.
#include <stdlib.h>

#include <type_traits>

#include <mutex>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) void* lptr_t;

#define WH_OOB 0x78000000u

#define WH_MAXGROUP 8
struct WgradHaloArgs {
  const unsigned short* u[WH_MAXGROUP]; int up; unsigned int u_bytes;     // dY [N][H][W][up], one per grouped layer
  const unsigned short* v[WH_MAXGROUP]; int vp; unsigned int v_bytes;     // X  [N][H][W][vp]
  int N, H, W, spy, spx;      // dY map; strips per image column / row (7 rows x 14 columns each)
  int XH, XW;                 // S2: the conv input map (2 H x 2 W); otherwise = H, W
  int nstrips, chunk;         // total strips, strips per split
  int zper;                   // splits per layer: grid.z = layers x zper, z = layer * zper + split
  int remap;                  // 1: XCD-aware workgroup order (all dW tiles of one z on one XCD's L2)
  int pair7;                  // 7 x 7 maps: a strip = TWO images side by side (columns 0-6 | gap | 8-14): the P7 instantiation
  float* ws;                  // [z][up][9][vp]
  BnIn xin;                   // xin.scale != nullptr: X is PReLU(v * scale + shift), applied in LDS (common.h)
};

// CO = Cout rows per workgroup: 128 (a wave carries a pair of 32-row tiles) or 64 (one tile)
// XF: the conv input X is a BatchNorm(+PReLU) of the stored tensor v; the strips are normalised in LDS
// P7: 7 x 7 maps, a strip = two images side by side (issue()); compile-time, the 128-row instantiation has no register
// to spare (a run-time switch spilled three dwords: 940 -> 595 TFLOP/s on the 14 x 14 layers)
// S2 (round 5): stride-2 convs (conv2 of the first block of a stage, backbones/frb/iresnet.py:56-67 with stride 2): the X
// strip is held as its four PARITY PLANES X(2i + py, 2j + px), each 8 rows x 16 pixels x 2 channel groups around the dY
// strip (halo origin -1); tap (r, s) reads plane ((r != 1), (s != 1)) at row offset (r != 0), column offset (s != 0) -- a
// per-tap constant like the stride-1 shifts, the MFMA loop is unchanged.  64 KB of X per strip: 64-row tiles only (LDS).
template <int CO, bool XF = false, bool P7 = false, bool S2 = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_wgrad_halo(const WgradHaloArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(!S2 || (CO == 64 && !XF && !P7), "the stride-2 strips come with 64-row tiles, no input transform");
  constexpr int NXB = S2 ? 4 * 8 * 2 : 20;             // X blocks of 1 KB per strip
  constexpr int GU = CO / 32, NI = CO / 64, UBLK = 7 * GU, NBLK = UBLK + NXB, NISS = (NBLK + 7) / 8;
  constexpr int UB = UBLK * 1024, VB = NXB * 1024, STAGE = UB + VB;      // dY strip, X strip + halo (+ 1 row)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  MSML_LDS_REGION(smem, 2 * STAGE + (XF ? 3 * 64 * 4 : 0));
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int mp = wave & 1, nh = (wave >> 1) & 1, tg = wave >> 2;   // Cout tile pair, Cin half, tap group
  // Several layers of one shape share a launch (grid.z = layers x zper): the split-K slab traffic per layer falls
  // with the number of workgroups a layer gets.  XCD-aware order: hardware deals consecutive workgroups round-robin
  // to the 8 XCDs; re-numbering them so that the gx * gy dW tiles of one z (same dY / X strips) sit on ONE XCD
  // makes their operand re-reads L2 hits.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.remap) {
    const int T = gridDim.x * gridDim.y, total = T * gridDim.z;
    const int L = bx + gridDim.x * (by + gridDim.y * bz);
    const int Lp = (L & 7) * (total >> 3) + (L >> 3);
    const int tile = Lp % T;
    bz = Lp / T;
    bx = tile % gridDim.x;
    by = tile / gridDim.x;
  }
  const int layer = __builtin_amdgcn_readfirstlane(bz / p.zper);
  const int a0 = bx * CO, b0 = by * 64, split = bz - layer * p.zper;
  const int s_begin = split * p.chunk;
  int s_end = s_begin + p.chunk;
  if (s_end > p.nstrips) s_end = p.nstrips;
  const int spi = p.spy * p.spx;

  __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)p.u[layer], 0, (int)p.u_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)p.v[layer], 0, (int)p.v_bytes, 0x00020000);

  // DMA slot of a lane inside a 1-KB block [16 px][64 B]: pixel lane / 4, 16-B chunk lane % 4
  const int lp = lane >> 2, lc = lane & 3;
  // 7 x 7 maps (pair7): a strip is the PAIR of images 2 strip, 2 strip + 1 laid side by side on the 16-pixel pitch:
  // dY columns 0-6 = image A, 7 = zero, 8-14 = image B, 15 = zero; X halo columns 0 = zero, 1-7 = A, 8 = zero (A's right
  // and B's left padding at once), 9-15 = B, and "column 16" of a row is column 0 of the next one = zero.  Every tap
  // offset of the k loop below then reads exactly the pixels a 7 x 7 conv with zero padding reads: 14 real k-values
  // of 16 per strip row, as on the 14-wide maps, with no change to the MFMA loop.
  auto issue = [&](int strip, int buf) {
    char* ub = smem + buf * STAGE;
    char* vb = ub + UB;
    if constexpr (P7) {
      const int n = 2 * strip + (lp >> 3), x = lp & 7;  // image of the pair, column inside it (7 = the zero gap)
      const bool img = n < p.N;
#pragma unroll
      for (int i = 0; i < NISS; i++) {
        const int blk = wave + 8 * i;
        if (blk >= NBLK) break;
        if (blk < UBLK) {
          const int j = blk / GU, g = blk % GU;
          const unsigned int off = (img & (x < 7)) ? (unsigned int)((n * 7 + j) * 7 + x) * (unsigned int)(p.up * 2) +
                                                         (unsigned int)(a0 + g * 32 + lc * 8) * 2u
                                                   : WH_OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lptr_t)(ub + blk * 1024), 16, off, 0, 0, 0);
        } else {
          const int bb = blk - UBLK, hr = bb >> 1, g = bb & 1;
          const int y = hr - 1;                          // halo columns 0 and 8 (x == 0) are the zero padding
          const bool ok = img & ((unsigned)y < 7u) & (x > 0);
          const unsigned int off = ok ? (unsigned int)((n * 7 + y) * 7 + x - 1) * (unsigned int)(p.vp * 2) +
                                            (unsigned int)(b0 + g * 32 + lc * 8) * 2u
                                      : WH_OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(vb + bb * 1024), 16, off, 0, 0, 0);
        }
      }
      return;
    }
    const int n = strip / spi, rem = strip - n * spi, sy = rem / p.spx;
    const int y0 = sy * 7, x0 = (rem - sy * p.spx) * 14;
#pragma unroll
    for (int i = 0; i < NISS; i++) {
      const int blk = wave + 8 * i;                    // 7 GU dY blocks, then 20 X blocks (S2: 4 planes x 8 rows x 2 groups)
      if (blk >= NBLK) break;
      if (S2 && blk >= UBLK) {
        const int bb = blk - UBLK, plane = bb >> 4, hr = (bb >> 1) & 7, g = bb & 1;
        const int y = 2 * (y0 + hr - 1) + (plane >> 1), x = 2 * (x0 + lp - 1) + (plane & 1);
        const bool ok = ((unsigned)y < (unsigned)p.XH) & ((unsigned)x < (unsigned)p.XW);
        const unsigned int off = ok ? (unsigned int)((n * p.XH + y) * p.XW + x) * (unsigned int)(p.vp * 2) +
                                          (unsigned int)(b0 + g * 32 + lc * 8) * 2u
                                    : WH_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(vb + bb * 1024), 16, off, 0, 0, 0);
        continue;
      }
      if (blk < UBLK) {
        const int j = blk / GU, g = blk % GU;
        const int y = y0 + j, x = x0 + lp;
        const bool ok = (lp < 14) & (x < p.W) & (y < p.H);
        const unsigned int off = ok ? (unsigned int)((n * p.H + y) * p.W + x) * (unsigned int)(p.up * 2) +
                                          (unsigned int)(a0 + g * 32 + lc * 8) * 2u
                                    : WH_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lptr_t)(ub + blk * 1024), 16, off, 0, 0, 0);
      } else {
        const int bb = blk - UBLK, hr = bb >> 1, g = bb & 1;
        const int y = y0 + hr - 1, x = x0 + lp - 1;
        const bool ok = (hr < 9) & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
        const unsigned int off = ok ? (unsigned int)((n * p.H + y) * p.W + x) * (unsigned int)(p.vp * 2) +
                                          (unsigned int)(b0 + g * 32 + lc * 8) * 2u
                                    : WH_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(vb + bb * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  float* xtab = reinterpret_cast<float*>(smem + 2 * STAGE);    // XF: [3][64] scale, shift, alpha of channels b0 ..
  // XF: every wave normalises the X chunks it DMA'd itself (after its own vmcnt wait); padding stays zero
  auto xform = [&](int strip, int buf) {
    const int n = strip / spi, rem = strip - n * spi, sy = rem / p.spx;
    const int y0 = sy * 7, x0 = (rem - sy * p.spx) * 14;
    char* vb = smem + buf * STAGE + UB;
    const bool has_alpha = p.xin.alpha != nullptr;
#pragma unroll
    for (int i = 0; i < NISS; i++) {
      const int blk = wave + 8 * i;
      if (blk >= NBLK) break;
      if (blk >= UBLK) {
        const int bb = blk - UBLK, hr = bb >> 1, g = bb & 1;
        const int y = y0 + hr - 1, x = x0 + lp - 1;
        if ((hr < 9) & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W))
          bn_in_chunk(vb + bb * 1024 + lane * 16, xtab, 64, g * 32 + lc * 8, has_alpha);
      }
    }
  };

  f32x16 acc[NI][5];                                   // [Cout tile of the wave][tap of the group]
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int k = 0; k < 5; k++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][k][e] = 0.f;

  // transposing fragment read (wgrad_fast.hip): the lane addresses pixel px = 8 (g4 >> 1) + q4 (the
  // second read 4 pixels on), 4 channels pp of a 16-channel half g4 & 1
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int px = 8 * (g4 >> 1) + q4;
  const int chan = (2 * (g4 & 1) + (pp >> 1)) * 16 + (pp & 1) * 8;
  const int aofs = mp * NI * 1024 + px * 64 + chan;    // + row * GU KB, + 1024 for a pair's second tile (+ 256: second read)
  // X: pixel px + s of halo row (row + r); past pixel 15 it continues in the next row's block
  int vlo[3], vhi[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    // S2: a plane has no spare row behind its eighth -- "pixel 16" of the last row would be read from memory no request
    // ever filled (it only meets the zero dY pixel 15, but 0 x NaN is NaN: one NaN in three runs of the variant's test,
    // round 6) -- so the shifted read stays on pixel 15 of its own row, which is loaded (or zero-filled) and finite
    const int p0 = (S2 && px + s > 15) ? 15 : px + s, p1 = (S2 && px + 4 + s > 15) ? 15 : px + 4 + s;
    vlo[s] = nh * 1024 + (p0 >> 4) * 2048 + (p0 & 15) * 64 + chan;
    vhi[s] = nh * 1024 + (p1 >> 4) * 2048 + (p1 & 15) * 64 + chan;
  }
  // offsets of this wave's taps (wave-uniform choice among the three column shifts, row shift folded in)
  const int ntaps = tg == 0 ? 5 : 4, tap0 = tg * 5;
  int vlok[5], vhik[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int tp = tap0 + k, r = tp / 3, sft = tp - r * 3;
    if (S2) {
      const int plane = (r != 1) * 2 + (sft != 1), lr = r != 0, ls = sft != 0;
      vlok[k] = plane * 16384 + lr * 2048 + (ls ? vlo[1] : vlo[0]);
      vhik[k] = plane * 16384 + lr * 2048 + (ls ? vhi[1] : vhi[0]);
    } else {
      vlok[k] = r * 2048 + (sft == 0 ? vlo[0] : (sft == 1 ? vlo[1] : vlo[2]));
      vhik[k] = r * 2048 + (sft == 0 ? vhi[0] : (sft == 1 ? vhi[1] : vhi[2]));
    }
  }
  typedef __attribute__((address_space(3))) s16x4* tr_ptr;
  auto tr2 = [&](const char* lo, const char* hi) -> s16x8 {
    s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(lo));
    s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(hi));
    s16x8 o;
    o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = l[3];
    o[4] = h[0]; o[5] = h[1]; o[6] = h[2]; o[7] = h[3];
    return o;
  };

  if (s_begin < s_end) issue(s_begin, 0);
  if (XF) bn_in_fill(p.xin, xtab, b0, 64, t, 512);
  __syncthreads();
  if (XF) {
    if (s_begin < s_end) xform(s_begin, 0);
    __syncthreads();
  }
  int cur = 0;
  s16x8 fa[2][NI], fb5[2][5];
  for (int strip = s_begin; strip < s_end; strip++) {
#ifndef WH_ABLATE_LOADS
    if (strip + 1 < s_end) issue(strip + 1, cur ^ 1);
#endif
    const char* ub = smem + cur * STAGE + aofs;
    const char* vb = smem + cur * STAGE + UB;
    // taps tap0 + k, k < ntaps, of this wave's group.  All fragments of k-step j + 1 are requested
    // before the MFMAs of k-step j (register double buffer); the fences keep hipcc from sinking the
    // reads next to their MFMAs.
    auto fetch = [&](int j, int fb) {
#pragma unroll
      for (int i = 0; i < NI; i++)
        fa[fb][i] = tr2(ub + j * (GU * 1024) + i * 1024, ub + j * (GU * 1024) + i * 1024 + 256);
#pragma unroll
      for (int k = 0; k < 5; k++)
        if (k < ntaps) fb5[fb][k] = tr2(vb + j * 2048 + vlok[k], vb + j * 2048 + vhik[k]);
    };
#ifdef WH_ABLATE_READS
    if (strip == s_begin)
#endif
    fetch(0, 0);
#pragma unroll
    for (int j = 0; j < 7; j++) {                      // k-step = strip row j (16 pixels)
      const int fb = j & 1;
#ifdef WH_ABLATE_READS
      if (j == 0 && strip == s_begin) fetch(1, 1);
#else
      if (j + 1 < 7) fetch(j + 1, fb ^ 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 5; k++)
        if (k < ntaps) {
#pragma unroll
          for (int i = 0; i < NI; i++)
            acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[fb][i]),
                                                                __builtin_bit_cast(bf16x8, fb5[fb][k]), acc[i][k], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      // next strip's X chunks (requested at the top of this strip).  The 128-row variant has no
      // registers to spare while fragments are live, so it waits for the last k-step (the VALU
      // work then runs in the shadow of the MFMAs just issued).
      if (XF && j == (CO == 128 ? 6 : 2) && strip + 1 < s_end) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xform(strip + 1, cur ^ 1);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

#ifdef WH_ABLATE_EPI
  if (p.N >= 0) return;
#endif
  // slab [z][a][tap][b] (k_wgrad_reduce sums the zper splits of a layer in a fixed order)
  const int h = lane >> 5, c32 = lane & 31;
  const int b = b0 + nh * 32 + c32;
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int k = 0; k < 5; k++) {
      if (k >= ntaps) continue;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int a = a0 + (mp * NI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        p.ws[(((long)bz * p.up + a) * 9 + tap0 + k) * p.vp + b] = acc[i][k][e];
      }
    }
#endif
}

static int wh_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}

// splits for a shape (0 = shape not covered by this kernel)
// 128-row dW tiles halve the operand traffic per FLOP; 64-row tiles halve the split-K slab bytes
// (measured at 256 -> 256 @ 14x14: 97 us with 128-row tiles, 104 us with 64-row tiles)
static bool wh_wide(int up) {
  static const int lim = getenv("MSML_WGRAD_HALO_WIDE_MIN") ? atoi(getenv("MSML_WGRAD_HALO_WIDE_MIN")) : 128;
  return up % 128 == 0 && up >= lim;
}

int msml_wgrad_halo_splits(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q, int R, int S,
                           int stride, int pad_h, int pad_w) {
  static const bool off = getenv("MSML_NO_HALO_WGRAD") != nullptr;
  if (off) return 0;
  if (R != 3 || S != 3 || stride != 1 || pad_h != 1 || pad_w != 1 || P != H || Q != W) return 0;
  if (up % 64 != 0 || vp % 64 != 0 || A != up || Breal != vp) return 0;
  const bool pair7 = H == 7 && W == 7 && getenv("MSML_WGRAD_HALO_NO_PAIR7") == nullptr;   // two 7 x 7 images per strip
  const long strips = pair7 ? (N + 1) / 2 : (long)N * cdiv(H, 7) * cdiv(W, 14);
  if ((long)N * H * W * 10 < strips * 112 * 7) return 0;           // < 70 % real k-values
  if ((long)N * H * W * up * 2 >= 0x70000000L || (long)N * H * W * vp * 2 >= 0x70000000L) return 0;
  // (image-pair strips: 64-row tiles only -- the 128-row instantiation of that variant spills)
  const int tiles = ((wh_wide(up) && !pair7) ? up / 128 : up / 64) * (vp / 64);
  long splits = wh_cus() / tiles;                      // one resident workgroup per CU
  if (splits < 1) splits = 1;
  if (splits > strips) splits = strips;
  if (splits > 512) splits = 512;
  return (int)splits;
}

template <int CO, bool XF, bool P7 = false, bool S2 = false>
static void wh_launch(const WgradHaloArgs& a, dim3 grid, hipStream_t st) {
  // two stages of (7 dY row blocks per 32 Cout + 10 X rows x 2 channel groups [S2: 4 planes x 8 rows x 2]) KB (+ coefficient table)
  const size_t lds = 2 * (7 * (CO / 32) + (S2 ? 64 : 10 * 2)) * 1024 + (XF ? 3 * 64 * sizeof(float) : 0);
  static std::once_flag attr_once;                     // (per template instantiation; launches come from
  std::call_once(attr_once, [&] {                      //  the forward thread AND the autograd thread)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_halo<CO, XF, P7, S2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  k_wgrad_halo<CO, XF, P7, S2><<<grid, dim3(512), lds, st>>>(a);
}

// Stride-2 3x3 / pad-1 weight gradient on the strip kernel (S2): splits (0 = shape not covered).  (H, W) = the conv input,
// (P, Q) = dY.
int msml_wgrad_halo_s2_splits(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q, int R, int S, int stride,
                              int pad_h, int pad_w) {
  // Measured (round 5, tools/bench_conv.py --only wgrad, batch 256): NOT faster than the im2col kernel -- 64 @ 112 152 vs 153 us,
  // 128 @ 56 136 vs 124, 256 @ 28 110 vs 121: the four X planes are 64 KB of LDS fill per strip against 34 KB at stride 1, and
  // only 64-row tiles fit.  Opt-in (read per call, so that a test can switch it on): MSML_HALO_WGRAD_S2=1.
#ifndef MSML_EXPERIMENTS
  return 0;                    // (the S2 instantiation exists in experiment builds only: tools/build_variant.py --all MSML_EXPERIMENTS)
#endif
  if (getenv("MSML_NO_HALO_WGRAD") != nullptr || getenv("MSML_HALO_WGRAD_S2") == nullptr) return 0;
  if (R != 3 || S != 3 || stride != 2 || pad_h != 1 || pad_w != 1 || (H & 1) || (W & 1) || P != H / 2 || Q != W / 2) return 0;
  if (up % 64 != 0 || vp % 64 != 0 || A != up || Breal != vp) return 0;
  const long strips = (long)N * cdiv(P, 7) * cdiv(Q, 14);
  if ((long)N * P * Q * 10 < strips * 112 * 7) return 0;           // < 70 % real k-values (7x7 / 4x4 maps: im2col kernel)
  if ((long)N * P * Q * up * 2 >= 0x70000000L || (long)N * H * W * vp * 2 >= 0x70000000L) return 0;
  const int tiles = (up / 64) * (vp / 64);
  long splits = wh_cus() / tiles;
  if (splits < 1) splits = 1;
  if (splits > strips) splits = strips;
  if (splits > 512) splits = 512;
  return (int)splits;
}

bool msml_wgrad_halo_s2_launch(const void* u, int up, const void* v, int vp, float* ws, int N, int H, int W, int P, int Q,
                               int splits, hipStream_t st) {
  WgradHaloArgs a;
  for (int i = 0; i < WH_MAXGROUP; i++) {
    a.u[i] = (const unsigned short*)u;
    a.v[i] = (const unsigned short*)v;
  }
  a.up = up; a.u_bytes = (unsigned int)((long)N * P * Q * up * 2);
  a.vp = vp; a.v_bytes = (unsigned int)((long)N * H * W * vp * 2);
  a.N = N; a.H = P; a.W = Q; a.XH = H; a.XW = W; a.spy = cdiv(P, 7); a.spx = cdiv(Q, 14);
  a.pair7 = 0;
  a.nstrips = N * a.spy * a.spx;
  a.chunk = cdiv(a.nstrips, splits);
  a.zper = splits;
  a.ws = ws;
  a.xin = BnIn{nullptr, nullptr, nullptr};
  const int tiles = (up / 64) * (vp / 64);
  a.remap = (getenv("MSML_WGRAD_HALO_NO_REMAP") == nullptr && tiles >= 4 && splits % 8 == 0) ? 1 : 0;
#ifdef MSML_EXPERIMENTS
  wh_launch<64, false, false, true>(a, dim3(up / 64, vp / 64, splits), st);
#else
  (void)st;
  return false;
#endif
  return true;
}

// splits per layer when `group` layers of this shape share one launch (0: not covered)
int msml_wgrad_halo_group_splits(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q, int R, int S,
                                 int stride, int pad_h, int pad_w, int group) {
  const int hs = msml_wgrad_halo_splits(up, vp, A, Breal, N, H, W, P, Q, R, S, stride, pad_h, pad_w);
  if (hs <= 0 || group < 1 || group > WH_MAXGROUP) return 0;
  const int per = hs / group;
  return per < 1 ? 0 : per;
}

// group: number of layers (u[i], v[i]), i < group, of ONE shape; splits = splits per layer; slabs land in
// ws[(layer * splits + split)][up][9][vp]
bool msml_wgrad_halo_launch_group(const void* const* u, int up, const void* const* v, int vp, float* ws, int N, int H,
                                  int W, int group, int splits, hipStream_t st, const BnIn* xin) {
  static const bool no_remap = getenv("MSML_WGRAD_HALO_NO_REMAP") != nullptr;
  WgradHaloArgs a;
  for (int i = 0; i < WH_MAXGROUP; i++) {
    a.u[i] = (const unsigned short*)u[i < group ? i : 0];
    a.v[i] = (const unsigned short*)v[i < group ? i : 0];
  }
  a.up = up; a.u_bytes = (unsigned int)((long)N * H * W * up * 2);
  a.vp = vp; a.v_bytes = (unsigned int)((long)N * H * W * vp * 2);
  a.N = N; a.H = H; a.W = W; a.XH = H; a.XW = W; a.spy = cdiv(H, 7); a.spx = cdiv(W, 14);
  a.pair7 = (H == 7 && W == 7) ? 1 : 0;
  a.nstrips = a.pair7 ? (N + 1) / 2 : N * a.spy * a.spx;
  a.chunk = cdiv(a.nstrips, splits);
  a.zper = splits;
  a.ws = ws;
  a.xin = BnIn{nullptr, nullptr, nullptr};
  if (xin) a.xin = *xin;
  const int gz = group * splits;
  const int tiles = ((wh_wide(up) && !a.pair7) ? up / 128 : up / 64) * (vp / 64);
  a.remap = (!no_remap && tiles >= 4 && gz % 8 == 0) ? 1 : 0;
  if (a.pair7) {
    if (xin) return false;                             // (no in-LDS BatchNorm variant of the image-pair strips)
    wh_launch<64, false, true>(a, dim3(up / 64, vp / 64, gz), st);
    return true;
  }
  if (wh_wide(up)) {
    if (xin) wh_launch<128, true>(a, dim3(up / 128, vp / 64, gz), st);
    else wh_launch<128, false>(a, dim3(up / 128, vp / 64, gz), st);
  } else {
    if (xin) wh_launch<64, true>(a, dim3(up / 64, vp / 64, gz), st);
    else wh_launch<64, false>(a, dim3(up / 64, vp / 64, gz), st);
  }
  return true;
}

bool msml_wgrad_halo_launch(const void* u, int up, const void* v, int vp, float* ws, int N, int H, int W,
                            int splits, hipStream_t st, const BnIn* xin) {
  return msml_wgrad_halo_launch_group(&u, up, &v, vp, ws, N, H, W, 1, splits, st, xin);
}
