// 3x3 / stride-1 / pad-1 convolution (forward, and backward-data with the tap walk flipped) for
// bf16 NHWC maps with Cin % 64 == 0 and Cout % 128 == 0 -- the 28x28x128 and 14x14x256 stages that
// hold 36 of the 48 3x3 convs of IResNet-50 (backbones/frb/iresnet.py:40-67) and the matching
// levels of the OSB encoder (backbones/osb/unet.py:80-91).  The image is cut into 14 x 14 pixel
// tiles; same contract as msml_conv2d / msml_conv2d_fused.
//
// k_conv_fast (im2col gather) re-fetches every input pixel 9 times and is bound by the
// L2 -> LDS fill rate (~80 GB/s per CU).  Here a workgroup owns a strip of TH image rows and all
// BN output channels:
//   * the input strip WITH its 1-pixel halo goes to LDS once per 64-channel slab; the image is
//     stored with a power-of-two row pitch (PITCH >= W + 1), so the right halo column of row y is
//     the left halo column of row y + 1 (both zero padding) and GEMM row m <-> LDS pixel m, and
//     the A operand of tap (r, s) is the SAME image read at the constant offset r * PITCH + s:
//     fill traffic per MFMA drops ~1.7x, addressing is one immediate per tap,
//   * GEMM rows whose x >= W are padding (196 real of 224 rows for 14x14) and are dropped in
//     the epilogue; in exchange one image = one workgroup = one even wave of 256 workgroups
//     (the 128-row tiling left the second round of workgroups half empty),
//   * weights stream through a double-buffered [BN][64] stage per (slab, tap) by LDS-DMA,
//   * every wave owns 32 output channels x all rows (7 accumulator tiles): one B fragment feeds
//     7 MFMAs; per-channel BatchNorm partial sums need no cross-wave reduction.
// LDS-DMA zero-fills out-of-range offsets (conv padding) -- see conv_fast.hip.
#include <stdlib.h>

#include <mutex>

#include "common.h"

struct ConvHaloArgs {
  const unsigned short* in;
  unsigned int in_bytes;
  int C;              // input channels (multiple of 64)
  int N, H, W, tpy, tpx;   // 14 x 14 pixel tiles per image column / row
  int flip;           // backward-data: tap (r, s) reads the image at (2 - r, 2 - s)
  const unsigned short* wp;
  unsigned int w_bytes;
  int Ktot;
  unsigned short* out;
  int coutp;
  const float* bias;
  const float* scale;
  const float* alpha;
  const unsigned short* residual;
  int res_first;
  int bias9;                      // X3: `bias` is float[9][coutp] by border class (common.h)
  float* stats;
  int stats_rows;
  int stats_acc;      // accumulator mode (common.h): stats is double[MSML_ACC_ROWS][2][coutp]
  BnBwdFuse bnb;      // bnb.partial != nullptr: fused BatchNorm backward-reduce (common.h)
  BnIn xin;           // xin.scale != nullptr: BatchNorm(+PReLU) applied to the input image in LDS (common.h)
  BnBwdIn bin;        // bin.x != nullptr: BatchNorm BACKWARD applied to the input image in LDS (common.h; XB kernels)
};

#define HALO_OOB 0x78000000u

typedef __attribute__((address_space(3))) void* lptr_t;

#ifdef HALO_TRACE
// (-DHALO_TRACE, tools/halo_trace.py: wave `HALO_TRACE_WAVE` of workgroup 0 stamps s_memrealtime at the points of every stage)
__device__ unsigned long long g_halo_trace[8192];
extern "C" int msml_halo_trace_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_halo_trace), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#define HALO_STAMP(k)                                                                           \
  do {                                                                                          \
    if (blockIdx.x == 0 && wave == HALO_TRACE_WAVE && lane == 0 && tix < 8192 - 8)              \
      g_halo_trace[tix++] = ((unsigned long long)(k) << 56) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffffffffull); \
  } while (0)
#ifndef HALO_TRACE_WAVE
#define HALO_TRACE_WAVE 0
#endif
#else
#define HALO_STAMP(k)
#endif

// FUSE: backward-data launch with the fused BatchNorm backward-reduce (no bias / scale / PReLU /
// residual / statistics in that case) -- a compile-time split keeps both epilogues in registers.
// BN output channels per workgroup, NWM wave groups along the pixel rows: (256, 1) = 8 waves x
// 7 accumulator tiles, (128, 2) = 4 channel groups x {4, 3} tiles -- waves w and w + 4 share a SIMD,
// so every SIMD still carries 7 tiles.
// XF: forward launch whose input is PReLU(in * xin.scale + xin.shift), applied per slab in LDS.
// X3: split-bf16 inference (x3.hip): `in` / `residual` / `out` hold 3 x their logical channels as planes
// [hi | lo | hi]; C = 3 x the logical input channels (the K loop is unaware), coutp = logical output channels.
// M16: the MFMAs are v_mfma_f32_16x16x32_bf16 (28 accumulator tiles of 16 channels x 16 pixels per wave instead of 7 of
// 32 x 32): same fragments, LDS reads and FLOP per stage, but the chip holds a higher clock under a 16x16x32 stream
// (MI355X_MICROARCH.md, DVFS give-back item 7).  Plain forward and FUSE launches only (no XF / X3).
// XB: backward-data launch whose input is the BatchNorm backward of (in = dy, bin.x = the BatchNorm's saved input), applied
// per slab in LDS from the producer's accumulated sums, written through to bin.store (M16 + FUSE instantiations only).
// R15 (round 5, M16 kernels): the weight ring with a prefetch distance of 1.5 stages on the same 8 KB per wave.  A stage's
// weights are two k-window halves [w][32 rows][64 B] (chunk c at c ^ ((row >> 2) & 3), the 64-B-row swizzle of
// conv_line.hip); half w = 0 of stage s + 2 is requested in the MIDDLE of stage s (its slot is free once the w = 0 fragments
// are in registers), half w = 1 of stage s + 1 at the start of stage s: two requests at a time, the mid-stage ones in the
// shadow of the MFMAs, instead of four in front of every stage with all eight waves queueing on the CU's one path into LDS;
// every wait is counted (s_waitcnt vmcnt(N) returns when all but the N youngest requests are done: N = 4, + 4 while a slab
// image requested after the wanted half is still among them), and the image is requested after the mid-stage weights.
// BREG (round 6, M16 kernels, experiment builds, opt-in MSML_HALO_BREG=1): the wave's weights never touch LDS.  The B fragment of the 16x16x32
// MFMA is 16 B per lane (output channel 16 g + l16, k chunk 4 w + q16) -- exactly one buffer_load_dwordx4 per (g, w) from the
// packed weights, four per stage and wave, double-buffered in registers (the stage's fragments are copied out of the landing
// registers behind the stage's wait, then the next stage's loads are issued into them).  What it takes off the CU's LDS:
// 32 KB of LDS-DMA writes and 32 KB of fragment reads per stage (of 288 KB), and the DMA requests' issue cost.
template <int BN, int NWM, bool FUSE, bool XF = false, bool X3 = false, bool M16 = false, bool XB = false, bool R15 = false,
          bool BREG = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv_halo(const ConvHaloArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(!R15 || M16, "the half-stage ring rides on the 16x16x32 tiling");
  static_assert(!BREG || (M16 && !R15), "weights in registers: 16x16x32 tiling, two-slot schedule");
  constexpr int PL2 = 4, PITCH = 16, MT = 7, KG = BN / 32, NW = KG * NWM, NT = NW * 64, BM = MT * 32;
  constexpr int TW = 14, TH = 14, HR = TH + 2, HPX = HR << PL2;
  constexpr int MTW = NWM == 1 ? MT : 4;               // accumulator tiles of one wave (at most)
  constexpr int ABYTES = HPX * 128;
  constexpr int NAJ = HPX / 8;                         // DMA wave-instructions per slab image
  constexpr int NAI = (NAJ + NW - 1) / NW;
  static_assert(NW == 8 && HPX % 8 == 0, "tile config");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                                     // [2][HPX][128 B]
  char* Bs = smem + 2 * ABYTES;                        // [NW][2][32][128 B]
  float* xtab = reinterpret_cast<float*>(smem + 2 * ABYTES + NW * 8192);   // XF: [3][C] scale, shift, alpha
  MSML_LDS_REGION(As, 2 * ABYTES);
  MSML_LDS_REGION(Bs, NW * 8192);
  if (XF) MSML_LDS_REGION(xtab, 3 * p.C * 4);
  if (XB) MSML_LDS_REGION(xtab, 7 * p.C * 4);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);     // scalar: LDS-DMA bases go to M0
  // XOR key of the eight 16-B chunks of LDS row p.  ds_read_b128 is served in four NON-contiguous 16-lane groups
  // ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md, LDS): under the 32x32x16 fragment map (row = lane & 31) the key
  // (p >> 1) & 7 is conflict-free for every tap shift, under the 16x16x32 map (row = lane & 15, chunk = lane >> 4)
  // it is 2-way for the shifted taps and p & 7 is the conflict-free one (tools/probe/lds_probe.hip, lds_groups.py)
  auto skey = [](int p_) { return M16 ? (p_ & 7) : ((p_ >> 1) & 7); };
  const int kg = wave % KG, mg = wave / KG;            // channel group, pixel-row group
  const int i0 = mg * 4, nmt = NWM == 1 ? MT : (mg == 0 ? 4 : 3);   // this wave's tiles [i0, i0 + nmt)
  const int tile = blockIdx.x, tpi = p.tpy * p.tpx;
  const int n = tile / tpi, trem = tile - n * tpi, ty = trem / p.tpx;
  const int y0 = ty * TH, x0 = (trem - ty * p.tpx) * TW;
  const int n0 = blockIdx.y * BN;
  auto pix_ok = [&](int m) { return ((m & 15) < TW) & (x0 + (m & 15) < p.W) & (y0 + (m >> 4) < p.H); };
  auto pix_off = [&](int m) { return ((long)(n * p.H + y0 + (m >> 4)) * p.W + x0 + (m & 15)) * p.coutp; };

  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.w_bytes, 0x00020000);

  // halo image: LDS pixel hp = hy * PITCH + hx holds input (y0 + hy - 1, x0 + hx - 1) (zeros
  // outside the image); its eight 16-B chunks are XOR-swizzled by (hp >> 1) & 7
  unsigned int aoff[NAI];
#pragma unroll
  for (int i = 0; i < NAI; i++) {
    const int j = wave + i * NW;
    const int hp = j * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ skey(hp);
    const int iy = y0 + (hp >> PL2) - 1, ix = x0 + (hp & (PITCH - 1)) - 1;
    const bool v = (j < NAJ) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
    aoff[i] = v ? (unsigned int)((n * p.H + iy) * p.W + ix) * (unsigned int)(p.C * 2) + logical * 16u : HALO_OOB;
  }
  // weights: every wave streams ITS 32 output channels through a private two-stage ring
  // Bs[wave][2][32 rows][128 B].  Only the issuing wave reads them, so its own vmcnt orders the
  // reads (MI355X_MICROARCH.md item 7) and the K loop needs a workgroup barrier only when the
  // shared image slab changes (every 9 stages) -- waves drift apart and one wave's fragment
  // reads overlap its SIMD partner's MFMAs.
  unsigned int boffg[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = i * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ skey(row);
    boffg[i] = (unsigned int)((n0 + kg * 32 + row) * p.Ktot) * 2u + logical * 16u;
  }
  u32x4 xr2[XB ? NAI : 1];                             // XB: this lane's chunks of the BatchNorm's saved input
  auto issue_a = [&](int cs, int buf) {
    char* a = As + buf * ABYTES;
#pragma unroll
    for (int i = 0; i < NAI; i++) {
      const int j = wave + i * NW;
      if (j < NAJ)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(a + j * 1024), 16, aoff[i] + cs * 128u, 0, 0, 0);
      if constexpr (XB) {
        xr2[i] = (j < NAJ && aoff[i] != HALO_OOB)
                     ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.bin.x) + aoff[i] + cs * 128u)
                     : u32x4{0, 0, 0, 0};
      }
    }
  };
  // XF: every wave normalises the chunks it DMA'd itself (ordered by its own vmcnt wait); the
  // zero padding stays zero
  auto xform = [&](int cs, int buf) {
    char* a = As + buf * ABYTES;
    const bool has_alpha = p.xin.alpha != nullptr;
    // 16x16x32 tiling (key p & 7): hp & 7 == (lane >> 3) & 7 for every chunk of this lane, i.e. ONE channel chunk per
    // lane and slab -- its coefficients are read from the table once
    f32x4 rsc[2], rsh[2], ral[2];
    if constexpr (M16 && !XB) {
      const float* tb = xtab + cs * 64 + (((lane & 7) ^ ((lane >> 3) & 7)) << 3);
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        rsc[hf] = *reinterpret_cast<const f32x4*>(tb + hf * 4);
        rsh[hf] = *reinterpret_cast<const f32x4*>(tb + p.C + hf * 4);
        ral[hf] = *reinterpret_cast<const f32x4*>(tb + 2 * p.C + hf * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < NAI; i++) {
      const int j = wave + i * NW;
      const int hp = j * 8 + (lane >> 3);
      const int logical = (lane & 7) ^ skey(hp);
      if (j < NAJ && aoff[i] != HALO_OOB) {
        if constexpr (XB) bnbin_chunk(a + j * 1024 + lane * 16, xr2[i], xtab, p.C, cs * 64 + logical * 8, p.bin.alpha != nullptr);
        else if constexpr (M16) bn_in_chunk_r(a + j * 1024 + lane * 16, rsc, rsh, ral, has_alpha);
        else bn_in_chunk(a + j * 1024 + lane * 16, xtab, p.C, cs * 64 + logical * 8, has_alpha);
        // write-through of the normalised image (accumulator mode): the pixels this tile OWNS (not its halo), once
        // per pixel tile (the first channel block of the grid)
        const int hy = hp >> PL2, hx = hp & (PITCH - 1);
        unsigned short* through = XB ? p.bin.store : p.xin.store;
        if (through && blockIdx.y == 0 && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW)
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(through) + aoff[i] + cs * 128u) =
              *reinterpret_cast<const u32x4*>(a + j * 1024 + lane * 16);
      }
    }
  };
  auto issue_b = [&](int cs, int tap, int buf) {
    char* b = Bs + wave * 8192 + buf * 4096;
    const unsigned int col = (unsigned int)(tap * p.C + cs * 64) * 2u;
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(b + i * 1024), 16, boffg[i] + col, 0, 0, 0);
  };

  // BREG: this lane's 16 B of (channel 16 g + l16, k chunk 4 w + q16) of stage (cs, tap) straight into registers
  u32x4 bnx[2][2];                                     // [w][g]: landing registers of the NEXT stage
  unsigned int boffr[2];
#pragma unroll
  for (int g = 0; g < 2; g++)
    boffr[g] = (unsigned int)((n0 + kg * 32 + 16 * g + (lane & 15)) * p.Ktot) * 2u + (unsigned int)((lane >> 4) * 16);
  auto load_b = [&](int cs_, int tap) {
    const unsigned int col = (unsigned int)(tap * p.C + cs_ * 64) * 2u;
#pragma unroll
    for (int w = 0; w < 2; w++)
#pragma unroll
      for (int g = 0; g < 2; g++)
        bnx[w][g] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, boffr[g] + w * 64u, col, 0);
  };

  // R15: half (tap, slab cs, window w) -> slot `buf`: two requests of 16 rows x 64 B
  auto key2 = [](int row) { return (row >> 2) & 3; };
  unsigned int boffh[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = i * 16 + (lane >> 2);
    boffh[i] = (unsigned int)((n0 + kg * 32 + row) * p.Ktot) * 2u + (unsigned int)(((lane & 3) ^ key2(row)) * 16);
  }
  auto issue_h = [&](int cs_, int tap, int w, int buf) {
    char* b = Bs + wave * 8192 + buf * 4096 + w * 2048;
    const unsigned int col = (unsigned int)(tap * p.C + cs_ * 64 + w * 32) * 2u;
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(b + i * 1024), 16, boffh[i] + col, 0, 0, 0);
  };

  // D = W_frag x X_frag: accumulator rows = output channels, columns (lanes) = pixels, so a lane
  // ends up with 4 consecutive channels of one pixel per register quad (8-B LDS stores below)
  static_assert(!M16 || !X3, "the 16x16x32 variant serves the plain forward, FUSE and XF launches");
  static_assert(!XB || (M16 && FUSE && !XF), "the backward input transform rides on the 16x16x32 FUSE launch");
  constexpr int NG = 2 * MTW, NGH = NG / 2;            // M16: 16-pixel groups of one wave (at most), per pipeline phase
  f32x16 acc[M16 ? 1 : MTW];
  f32x4 acc4[M16 ? NG : 1][2];                         // M16: [pixel group][channel half]: channels 16 g + 4 q + j
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : MTW); i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < (M16 ? NG : 1); i++)
#pragma unroll
    for (int g = 0; g < 2; g++) acc4[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r32 = lane & 31, h = lane >> 5;
  const int l16 = lane & 15, q16 = lane >> 4;          // M16: pixel / channel row inside a group, 8-deep k block
  // fragment offsets: pixel row (32 i + r32) of tap (r, s) is LDS pixel 32 i + r32 + r PITCH + s;
  // tile i and the vertical tap r are immediates ((32 i + r PITCH) / 2 = 0 mod 8 keeps the
  // swizzle), the horizontal tap s moves the row and its swizzle
  int bfr[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bfr[kk] = wave * 8192 + r32 * 128 + (((kk * 2 + h) ^ ((r32 >> 1) & 7)) << 4);

  // M16: weights row 16 g + l16 of the wave's ring, 16-B chunk 4 w + q16 of the stage's two 32-deep windows w
  int bfr16[2][2];
#pragma unroll
  for (int g = 0; g < 2; g++)
#pragma unroll
    for (int w = 0; w < 2; w++) {
      const int row = 16 * g + l16;
      bfr16[g][w] = R15 ? wave * 8192 + w * 2048 + row * 64 + ((q16 ^ key2(row)) << 4)
                        : wave * 8192 + row * 128 + (((4 * w + q16) ^ skey(row)) << 4);
    }

  const int nslab = p.C >> 6, nstage = nslab * 9;
  issue_a(0, 0);
  if constexpr (R15) {                                 // stage 0 whole, window 0 of stage 1 (stage q = slab q / 9, tap q % 9)
    issue_h(0, 0, 0, 0);
    issue_h(0, 0, 1, 0);
    issue_h(0, 1, 0, 1);
  } else if constexpr (BREG) {
    load_b(0, 0);
  } else {
    issue_b(0, 0, 0);
  }
  if (XF) {
    if (p.xin.acc) bn_in_fill_acc(p.xin, xtab, p.C, t, NT, blockIdx.x == 0 && blockIdx.y == 0);
    else bn_in_fill(p.xin, xtab, 0, p.C, t, NT);
  }
  if (XB) bnbin_fill_acc(p.bin, xtab, p.C, t, NT, blockIdx.x == 0 && blockIdx.y == 0);
  __syncthreads();                                     // (drains vmcnt first)
  if (XF || XB) {
    xform(0, 0);
    __syncthreads();
  }
  int cs = 0, tr = 0, ts = 0;                          // slab, tap row / column of stage q
#ifdef HALO_TRACE
  int tix = 0;
  if (wave == 0 && lane == 0) g_halo_trace[4096 + 2 * (blockIdx.x & 1023)] = __builtin_amdgcn_s_memrealtime();
#endif
#if defined(HALO_PRIO)
  // (experiment: one static priority for half of the waves -- the two waves of a SIMD otherwise run their MFMA phases in
  // lockstep, sharing the pipe, and then wait / request together with the pipe idle)
  if (HALO_PRIO == 1 ? (wave >= 4) : (wave < 4)) __builtin_amdgcn_s_setprio(1);
#endif
  u32x4 a[2][MTW], b[2];
  u32x4 bcur[2][2];                                    // BREG: [w][g] fragments of the current stage
  int img_m1 = 0, img_m2 = 0;                          // R15: a slab image was requested in the middle of stage q - 1 / q - 2
#ifdef HALO_ABLATE_READS
  u32x4 a16x[2][MTW], b16x[2][2];                      // (ablation build: fragments read once, reused by every stage)
#endif
  for (int q = 0; q < nstage; q++) {
    int ncs = cs, ntr = tr, nts = ts + 1;
    if (nts == 3) { nts = 0; ntr++; }
    if (ntr == 3) { ntr = 0; ncs++; }
    // this wave's weights of stage q (issued one stage ago) have landed; queue stage q + 1 and,
    // at the first tap of a slab, this wave's share of the next slab's image
#ifdef HALO_SKEW
    // (experiment: the second wave of every SIMD half a stage behind the first, re-established after every slab-switch
    // barrier -- while one waits for its weights and requests the next ones the other is in its MFMA phase)
    if (wave >= 4 && (tr | ts) == 0) __builtin_amdgcn_s_sleep(HALO_SKEW);
#endif
    HALO_STAMP(1);
    if constexpr (R15) {
      // window 0 of this stage (requested in the middle of stage q - 2); younger: window 1 of this stage, window 0 of the
      // next one, and a slab image if one was requested in the middle of one of the last two stages
      if (img_m1 | img_m2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      HALO_STAMP(2);
      // window 1 of stage q + 1 (its slot: read last in stage q - 1); past the last stage: a harmless re-read that keeps
      // the request pattern, and with it the counts, the same to the end
      {
        const int qn = q + 1 < nstage ? q + 1 : q;
        issue_h(qn / 9, qn % 9, 1, (q + 1) & 1);
      }
    } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HALO_STAMP(2);
    if constexpr (BREG) {
#pragma unroll
      for (int w = 0; w < 2; w++)
#pragma unroll
        for (int g = 0; g < 2; g++) bcur[w][g] = bnx[w][g];
    }
#ifndef HALO_ABLATE_LOADS
    if (q + 1 < nstage) {
      if constexpr (BREG) load_b(ncs, ntr * 3 + nts);
      else issue_b(ncs, ntr * 3 + nts, (q + 1) & 1);
      if ((tr | ts) == 0 && cs + 1 < nslab) issue_a(cs + 1, (cs + 1) & 1);
    }
#endif
    }
    // (the image chunks requested one stage ago have landed for this wave: the wait above)
    // (round 5: waves 4-7, the SIMD partners of 0-3, transform one tap later -- one wave's VALU beside the other's MFMAs;
    // bit-identical, 128 @ 28x28 bn + conv 103.6 -> 101.8 us, the step 29.41 / 29.50 -> 29.36 / 29.39 ms on one box.
    // -DHALO_XF_NO_STAGGER: all eight waves at tap 1)
#ifdef HALO_XF_NO_STAGGER
    if ((XF || XB) && tr == 0 && ts == 1 && cs + 1 < nslab) xform(cs + 1, (cs + 1) & 1);
#else
    if constexpr (R15) {
      // (the image was requested in the middle of tap 0: by tap 3 two waits with four requests to spare lie behind it)
      if ((XF || XB) && tr == 1 && ts == (wave < 4 ? 0 : 1) && cs + 1 < nslab) xform(cs + 1, (cs + 1) & 1);
    } else {
      if ((XF || XB) && tr == 0 && ts == (wave < 4 ? 1 : 2) && cs + 1 < nslab) xform(cs + 1, (cs + 1) & 1);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    HALO_STAMP(3);
#ifndef HALO_ABLATE_COMPUTE
    const int r = p.flip ? 2 - tr : tr, s = p.flip ? 2 - ts : ts;
    if constexpr (M16) {
      // pixel row (16 j + l16) of tap (r, s) is LDS pixel 16 j + l16 + r PITCH + s; the swizzle key follows l16 + s
      // ((16 j + r PITCH + 32 i0) / 2 = 0 mod 8).  A stage = two 32-deep windows x two halves of the pixel groups:
      // four phases of <= NGH fragments + 2 NGH MFMAs, the next phase's fragments requested before the current MFMAs.
      const int ng = 2 * nmt;
      const int arow = l16 + s, asw = skey(arow);
      const char* Arow = As + (cs & 1) * ABYTES + (((r << PL2) + i0 * 32) * 128) + arow * 128;
      const char* B = Bs + (q & 1) * 4096;
#ifdef HALO_ABLATE_READS
      if (q == 0) {
#pragma unroll
        for (int j = 0; j < NGH; j++) a16x[0][j] = a16x[1][j] = *reinterpret_cast<const u32x4*>(Arow + ((q16 ^ asw) << 4) + j * 2048);
#pragma unroll
        for (int g = 0; g < 2; g++) b16x[0][g] = b16x[1][g] = *reinterpret_cast<const u32x4*>(B + bfr16[g][0]);
      }
      u32x4 (&a16)[2][NGH] = a16x;
      u32x4 (&b16)[2][2] = b16x;
#else
      u32x4 a16[2][NGH], b16[2][2];
#pragma unroll
      for (int j = 0; j < NGH; j++)
        if (j < ng) a16[0][j] = *reinterpret_cast<const u32x4*>(Arow + ((q16 ^ asw) << 4) + j * 2048);
      if constexpr (!BREG) {
#pragma unroll
        for (int g = 0; g < 2; g++) b16[0][g] = *reinterpret_cast<const u32x4*>(B + bfr16[g][0]);
      }
#endif
#pragma unroll
      for (int ph = 0; ph < 4; ph++) {
        const int cb = ph & 1, nb = cb ^ 1, w = ph >> 1, hf = ph & 1;
        if constexpr (R15) {
          if (ph == 1) {
            // window 1 of this stage (requested at the start of stage q - 1); younger: window 0 of stage q + 1, the
            // image of the middle of stage q - 1 (if any), window 1 of stage q + 1
            if (img_m1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          }
        }
#ifndef HALO_ABLATE_READS
        if (ph + 1 < 4) {
          const int nw = (ph + 1) >> 1, nhf = (ph + 1) & 1;
          const int ao = ((4 * nw + q16) ^ asw) << 4;
#pragma unroll
          for (int j = 0; j < NGH; j++)
            if (nhf * NGH + j < ng) a16[nb][j] = *reinterpret_cast<const u32x4*>(Arow + ao + (nhf * NGH + j) * 2048);
          if (nhf == 0 && !BREG) {
#pragma unroll
            for (int g = 0; g < 2; g++) b16[nw & 1][g] = *reinterpret_cast<const u32x4*>(B + bfr16[g][nw]);
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NGH; j++)
          if (hf * NGH + j < ng) {
#pragma unroll
            for (int g = 0; g < 2; g++)
              acc4[hf * NGH + j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, BREG ? bcur[w][g] : b16[w & 1][g]), __builtin_bit_cast(bf16x8, a16[cb][j]),
                  acc4[hf * NGH + j][g], 0, 0, 0);
          }
        if constexpr (R15) {
          if (ph == (wave < 4 ? 1 : 2)) {              // (the two waves of a SIMD one phase apart: one requests while the other computes)
            // middle of the stage, behind the MFMAs just issued: window 0 of stage q + 2 into the slot whose fragments are
            // in registers, then (first tap of a slab) the next slab's image
            const int qn = q + 2 < nstage ? q + 2 : q;
            issue_h(qn / 9, qn % 9, 0, q & 1);
            img_m2 = img_m1;
            img_m1 = 0;
            if ((tr | ts) == 0 && cs + 1 < nslab) {
              issue_a(cs + 1, (cs + 1) & 1);
              img_m1 = 1;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    const int arow = r32 + s, asw = (arow >> 1) & 7;
    const char* Arow = As + (cs & 1) * ABYTES + (((r << PL2) + i0 * 32) * 128) + arow * 128;
    const char* B = Bs + (q & 1) * 4096;
    // register double buffer of the fragments of one 16-deep k step; the fences keep hipcc from
    // sinking the reads next to their MFMAs (which exposes the LDS latency at every MFMA)
#ifdef HALO_ABLATE_READS
    if (q == 0) {
#pragma unroll
      for (int i = 0; i < MTW; i++) {
        a[0][i] = *reinterpret_cast<const u32x4*>(Arow + ((h ^ asw) << 4) + i * 4096);
        a[1][i] = a[0][i];
      }
      b[0] = *reinterpret_cast<const u32x4*>(B + bfr[0]);
      b[1] = b[0];
    }
#else
#pragma unroll
    for (int i = 0; i < MTW; i++)
      if (i < nmt) a[0][i] = *reinterpret_cast<const u32x4*>(Arow + ((h ^ asw) << 4) + i * 4096);
    b[0] = *reinterpret_cast<const u32x4*>(B + bfr[0]);
#endif
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int cb = kk & 1, nb = cb ^ 1;
#ifndef HALO_ABLATE_READS
      if (kk + 1 < 4) {
        const int ao = (((kk + 1) * 2 + h) ^ asw) << 4;
#pragma unroll
        for (int i = 0; i < MTW; i++)
          if (i < nmt) a[nb][i] = *reinterpret_cast<const u32x4*>(Arow + ao + i * 4096);
        b[nb] = *reinterpret_cast<const u32x4*>(B + bfr[kk + 1]);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MTW; i++)
        if (i < nmt)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[cb]),
                                                           __builtin_bit_cast(bf16x8, a[cb][i]), acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
#endif
    HALO_STAMP(4);
    if (ncs != cs && ncs < nslab) {
      if constexpr (R15) {
        // (no drain of the request queue: this wave's share of the next image was requested nine stages ago and the
        // counted waits since have long covered it; what is in flight are the next stages' weights.  LDS traffic of the
        // in-LDS BatchNorm is waited for, the barrier orders the rest)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      } else {
        __syncthreads();                               // slab switch: next image landed everywhere
      }
      HALO_STAMP(5);
    }
    cs = ncs; tr = ntr; ts = nts;
  }
  __syncthreads();
  HALO_STAMP(5);

  // ---------------- epilogue: affine / PReLU, BatchNorm partials, LDS transpose, 16-B stores -----
#ifdef HALO_ABLATE_EPILOGUE
  if (p.N >= 0) return;
#endif
  constexpr int OP = BN + 8;                           // 528-B rows
  unsigned short* otile = reinterpret_cast<unsigned short*>(smem);
  MSML_LDS_REGION(otile, BM * OP * 2);
  // copy-out: thread t stores 16-B chunk c8 of rows (t + k NT) / C8.  With the fused BatchNorm
  // backward-reduce the saved BatchNorm input of those chunks is requested now, so the loads fly
  // during the accumulator -> LDS transpose.
  constexpr int C8 = BN / 8, ITERS = BM * C8 / NT;
  static_assert(NT % C8 == 0 && (BM * C8) % NT == 0, "a thread keeps one 8-channel chunk in the copy-out loop");
  const int c8 = t % C8;
  constexpr bool fuse = FUSE;
  // FUSE on the 16x16x32 tiling (round 5): the rounded dX leaves straight from registers (after v_permlane16_swap a lane
  // holds channels cdir .. cdir + 7 of its pixel) and feeds the BatchNorm sums in that layout, with the saved BatchNorm
  // input fetched per (pixel group, lane) -- no LDS transpose, no barrier between the main loop and the stores
  // Measured NOT faster on these maps (round 5, tools/bench_bnbwd.py + step A/B on one box: 128 @ 28x28 102.1 -> 99.5 us, 256 @
  // 14x14 65.8 -> 65.8 us, the step 29.24 / 29.21 -> 29.35 / 29.35 ms: the per-lane loads of the saved input touch 64 B per
  // pixel and wave where the copy-out loop reads whole 512-B pixel rows) -- kept as a build switch, -DHALO_FDIR.
#ifdef HALO_FDIR
  constexpr bool FDIR = FUSE && M16;
#else
  constexpr bool FDIR = false;
#endif
  const int cdir = kg * 32 + (q16 & 1) * 16 + (q16 >> 1) * 8;
  u32x4 xr[FDIR ? NG : ITERS];
  BnbCoef bk;
  float bq[3][8];
  if constexpr (FDIR) {
#pragma unroll
    for (int k = 0; k < NG; k++) {
      const int m = i0 * 32 + k * 16 + l16;
      xr[k] = (k < 2 * nmt && pix_ok(m)) ? *reinterpret_cast<const u32x4*>(p.bnb.x + pix_off(m) + n0 + cdir) : u32x4{0, 0, 0, 0};
    }
    bk = bnb_load_coef(p.bnb, n0 + cdir);
  } else if (fuse) {
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
      const int m = (t + k * NT) / C8;
      xr[k] = pix_ok(m) ? *reinterpret_cast<const u32x4*>(p.bnb.x + pix_off(m) + n0 + c8 * 8) : u32x4{0, 0, 0, 0};
    }
    bk = bnb_load_coef(p.bnb, n0 + c8 * 8);
  }
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) bq[q][j] = 0.f;
  // this lane's channels: kb + 8 g + j, g < 4 (M16: kb + 16 g + j, g < 2)
  const int kb = M16 ? kg * 32 + 4 * q16 : kg * 32 + 4 * h;
  const bool act_here = !FUSE && p.alpha && !(p.residual && p.res_first);
  f32x4 bv[4], sv[4], av[4], s1[4], s2[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int col = n0 + kb + (M16 ? 16 : 8) * (M16 ? (g & 1) : g);
    bv[g] = (!FUSE && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    sv[g] = (!FUSE && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    av[g] = act_here ? *reinterpret_cast<const f32x4*>(p.alpha + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    s1[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Forward launches without a residual store straight from registers: v_permlane32_swap gives
  // the lane pair of a pixel 16 contiguous channels each (32 B), so a wave writes 64 B runs per
  // pixel with two 16-B stores per lane and tile -- no LDS transpose, no barrier.
  const bool direct = X3 || FDIR || (!FUSE && p.residual == nullptr);
  if constexpr (M16) {
#pragma unroll
    for (int jg = 0; jg < NG; jg++) {
      if (jg >= 2 * nmt) break;
      const int m = i0 * 32 + jg * 16 + l16;
      const bool valid = pix_ok(m);
      u32x2 pk[2];
#pragma unroll
      for (int g = 0; g < 2; g++) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float z = acc4[jg][g][j];
          if (!FUSE) {
            z = z * sv[g][j] + bv[g][j];
            if (act_here) z = z > 0.f ? z : z * av[g][j];
          }
          v[j] = z;
          if (!FUSE && valid) {
            s1[g][j] += z;
            s2[g][j] += z * z;
          }
        }
        pk[g][0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
        pk[g][1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
        if (!direct) *reinterpret_cast<u32x2*>(otile + m * OP + kb + 16 * g) = pk[g];
      }
      if (direct) {
        // the four lanes (q16 = 0..3) of a pixel hold channels 16 g + 4 q16 + (0..3); v_permlane16_swap trades the
        // odd 16-lane rows of the g = 0 registers for the even rows of the g = 1 registers, which leaves every lane
        // with 8 CONTIGUOUS channels: rows 0 / 1 / 2 / 3 -> channels 0-7 / 16-23 / 8-15 / 24-31 of the wave's 32
        u32x4 o16;
#pragma unroll
        for (int e = 0; e < 2; e++) {
          auto sw = __builtin_amdgcn_permlane16_swap(pk[0][e], pk[1][e], false, false);
          o16[e] = sw[0]; o16[2 + e] = sw[1];
        }
        if (valid) {
#ifdef HALO_NT_STORE
          __builtin_nontemporal_store(o16, reinterpret_cast<u32x4*>(p.out + pix_off(m) + n0 + kg * 32 + (q16 & 1) * 16 + (q16 >> 1) * 8));
#else
          *reinterpret_cast<u32x4*>(p.out + pix_off(m) + n0 + kg * 32 + (q16 & 1) * 16 + (q16 >> 1) * 8) = o16;
#endif
          if constexpr (FDIR)
            bnb_accum(bk, p.bnb.alpha != nullptr, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&o16)),
                      load8<unsigned short>(reinterpret_cast<const unsigned short*>(&xr[jg])), bq);
        }
      }
    }
  } else {
#pragma unroll
  for (int i = 0; i < MTW; i++) {
    if (i >= nmt) break;
    const int m = (i0 + i) * 32 + r32;
    const bool valid = pix_ok(m);
    if constexpr (X3) {
      // split-bf16 output: residual (hi + lo planes) joins in f32, the result leaves as three planes
      const long po = pix_off(m) * 3;
      u32x2 pkh[4], pkl[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.residual && valid) {
          const unsigned short* rp = p.residual + po + n0 + kb + 8 * g;
          const u32x2 rh = *reinterpret_cast<const u32x2*>(rp), rl = *reinterpret_cast<const u32x2*>(rp + p.coutp);
          r4[0] = __uint_as_float(rh[0] << 16) + __uint_as_float(rl[0] << 16);
          r4[1] = __uint_as_float(rh[0] & 0xffff0000u) + __uint_as_float(rl[0] & 0xffff0000u);
          r4[2] = __uint_as_float(rh[1] << 16) + __uint_as_float(rl[1] << 16);
          r4[3] = __uint_as_float(rh[1] & 0xffff0000u) + __uint_as_float(rl[1] & 0xffff0000u);
        }
        unsigned short hh[4], ll[4];
        f32x4 bg = bv[g];
        if (p.bias9 && valid)
          bg = *reinterpret_cast<const f32x4*>(p.bias + border_class(y0 + (m >> 4), x0 + (m & 15), p.H, p.W) * p.coutp + n0 + kb +
                                               8 * g);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float z = acc[i][g * 4 + j] * sv[g][j] + bg[j];
          if (act_here) z = z > 0.f ? z : z * av[g][j];
          if (p.residual) {
            z += r4[j];
            if (p.res_first && p.alpha) z = z > 0.f ? z : z * p.alpha[n0 + kb + 8 * g + j];
          }
          hh[j] = f2bf(z);
          ll[j] = f2bf(z - bf2f(hh[j]));
        }
        pkh[g][0] = (unsigned int)hh[0] | ((unsigned int)hh[1] << 16);
        pkh[g][1] = (unsigned int)hh[2] | ((unsigned int)hh[3] << 16);
        pkl[g][0] = (unsigned int)ll[0] | ((unsigned int)ll[1] << 16);
        pkl[g][1] = (unsigned int)ll[2] | ((unsigned int)ll[3] << 16);
      }
      u32x4 loh, hih, lol, hil;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        auto a01 = __builtin_amdgcn_permlane32_swap(pkh[0][e], pkh[1][e], false, false);
        auto a23 = __builtin_amdgcn_permlane32_swap(pkh[2][e], pkh[3][e], false, false);
        loh[e] = a01[0]; loh[2 + e] = a01[1];
        hih[e] = a23[0]; hih[2 + e] = a23[1];
        auto b01 = __builtin_amdgcn_permlane32_swap(pkl[0][e], pkl[1][e], false, false);
        auto b23 = __builtin_amdgcn_permlane32_swap(pkl[2][e], pkl[3][e], false, false);
        lol[e] = b01[0]; lol[2 + e] = b01[1];
        hil[e] = b23[0]; hil[2 + e] = b23[1];
      }
      if (valid) {
        unsigned short* o = p.out + po + n0 + kg * 32 + 8 * h;
        *reinterpret_cast<u32x4*>(o) = loh;
        *reinterpret_cast<u32x4*>(o + 16) = hih;
        *reinterpret_cast<u32x4*>(o + p.coutp) = lol;
        *reinterpret_cast<u32x4*>(o + p.coutp + 16) = hil;
        *reinterpret_cast<u32x4*>(o + 2 * p.coutp) = loh;
        *reinterpret_cast<u32x4*>(o + 2 * p.coutp + 16) = hih;
      }
      continue;
    }
    u32x2 pk[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float z = acc[i][g * 4 + j];
        if (!FUSE) {
          z = z * sv[g][j] + bv[g][j];
          if (act_here) z = z > 0.f ? z : z * av[g][j];
        }
        v[j] = z;
        if (!FUSE && valid) {
          s1[g][j] += z;
          s2[g][j] += z * z;
        }
      }
      pk[g][0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
      pk[g][1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
      if (!direct) *reinterpret_cast<u32x2*>(otile + m * OP + kb + 8 * g) = pk[g];
    }
    if (direct) {
      // lanes (pixel, h = 0 / 1) hold channels 8 g + 4 h + (0..3); after the swaps h = 0 holds
      // channels 0-7 and 16-23, h = 1 holds 8-15 and 24-31 of the wave's 32
      u32x4 lo, hi;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        auto r01 = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
        auto r23 = __builtin_amdgcn_permlane32_swap(pk[2][e], pk[3][e], false, false);
        lo[e] = r01[0]; lo[2 + e] = r01[1];
        hi[e] = r23[0]; hi[2 + e] = r23[1];
      }
      if (valid) {
        unsigned short* o = p.out + pix_off(m) + n0 + kg * 32 + 8 * h;
        *reinterpret_cast<u32x4*>(o) = lo;
        *reinterpret_cast<u32x4*>(o + 16) = hi;
      }
    }
  }
  }
  if (!direct) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITERS; k++) {
    const int m = (t + k * NT) / C8;
    if (pix_ok(m)) {
      u32x4 v = *reinterpret_cast<const u32x4*>(otile + m * OP + c8 * 8);
      const int c0 = n0 + c8 * 8;
      const long o = pix_off(m) + c0;
      if (!FUSE && p.residual) {
        Vec8 a8 = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v));
        Vec8 r8 = load8<unsigned short>(p.residual + o);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          float z = a8.v[j] + r8.v[j];
          if (p.res_first && p.alpha) z = z > 0.f ? z : z * p.alpha[c0 + j];
          a8.v[j] = z;
        }
        store8<unsigned short>(reinterpret_cast<unsigned short*>(&v), a8);
      }
      *reinterpret_cast<u32x4*>(p.out + o) = v;
      if (fuse)
        bnb_accum(bk, p.bnb.alpha != nullptr, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v)),
                  load8<unsigned short>(reinterpret_cast<const unsigned short*>(&xr[k])), bq);
    }
  }
  }
  if (fuse) {
    constexpr int G = FDIR ? 16 * NWM : NT / C8;       // threads that share an 8-channel chunk
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    MSML_LDS_REGION(red, G * 3 * BN * 4);
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if constexpr (FDIR) red[((mg * 16 + l16) * 3 + q) * BN + cdir + j] = bq[q][j];
        else red[((t / C8) * 3 + q) * BN + (t % C8) * 8 + j] = bq[q][j];
      }
    __syncthreads();
    for (int i = t; i < 3 * BN; i += NT) {
      const int q = i / BN, c = i % BN;
      float sum = 0.f;
      for (int g = 0; g < G; g++) sum += red[(g * 3 + q) * BN + c];
      bnb_emit(p.bnb.partial, p.bnb.acc, blockIdx.x, q, p.coutp, n0 + c, sum);
    }
  }
  if (!FUSE && p.stats) {
    // per-channel (sum, sumsq) over this workgroup's pixels: lanes hold pixels, so the 32 partials
    // of every lane go through LDS and each lane adds up one (statistic, channel) in a fixed order.
    // One row pair per workgroup; the rows the 128-pixel tiling would have had beyond that are
    // zeroed so msml_bn_finalize can sum the whole [rows][2][C] block.
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + wave * 64 * 33;
    MSML_LDS_REGION(smem, NW * 64 * 33 * 4);
#pragma unroll
    for (int g = 0; g < (M16 ? 2 : 4); g++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        red[lane * 33 + g * 4 + j] = s1[g][j];
        red[lane * 33 + 16 + g * 4 + j] = s2[g][j];
      }
    __syncthreads();
    if (mg == 0) {                                     // the waves of pixel-row group 0 add both groups
      const int which = lane >> 5, kl = lane & 31;
      float sum = 0.f;
      if constexpr (M16) {                             // channel kl = 16 g + 4 q + j lives in the 16 lanes 16 q + rr
        const int k = which * 16 + (kl >> 4) * 4 + (kl & 3), qq = (kl >> 2) & 3;
#pragma unroll
        for (int gm = 0; gm < NWM; gm++)
#pragma unroll 8
          for (int rr = 0; rr < 16; rr++) sum += red[gm * KG * 64 * 33 + (qq * 16 + rr) * 33 + k];
      } else {                                         // channel kl = 8 g + 4 hh + j
        const int k = which * 16 + (kl >> 3) * 4 + (kl & 3), hh = (kl >> 2) & 1;
#pragma unroll
        for (int gm = 0; gm < NWM; gm++)
#pragma unroll 8
          for (int rr = 0; rr < 32; rr++) sum += red[gm * KG * 64 * 33 + (hh * 32 + rr) * 33 + k];
      }
      stats_emit(p.stats, p.stats_acc, blockIdx.x, which, p.coutp, n0 + kg * 32 + kl, sum);
    }
    for (int row = gridDim.x + blockIdx.x; !p.stats_acc && row < p.stats_rows; row += gridDim.x)
      for (int c = t; c < 2 * BN; c += NT)
        p.stats[((long)row * 2 + c / BN) * p.coutp + n0 + c % BN] = 0.f;
  }
  HALO_STAMP(6);
#ifdef HALO_TRACE
  if (wave == 0 && lane == 0) g_halo_trace[4096 + 2 * (blockIdx.x & 1023) + 1] = __builtin_amdgcn_s_memrealtime();
#endif
#endif
}

// ---- persistent variant of the 128-channel tile (round 5) ------------------------------------------------------
// A 128 -> 128 @ 28x28 launch is FOUR rounds of 14 x 14 tiles: every tile pays its exposed 64 KB image load, its epilogue
// burst, its statistics atomics and a workgroup launch -- ~12 us that the one-round 256-channel launch pays once (525
// against 930 TFLOP/s for the same FLOP).  Here a workgroup walks its tiles (blockIdx.x, + gridDim.x, ...) through ONE
// software pipeline: the (tile, slab) sequence keeps alternating the two image buffers and the weight rings across the
// tile boundary -- tile t + 1's first image is requested during tile t's last slab, its first weight stage before tile
// t's epilogue -- the epilogue stores straight from registers beside the next tile's first stages, and the BatchNorm
// statistics / fused backward sums stay in registers until the workgroup's last tile (one set of atomics per workgroup).
// Same tiling, fragment maps and per-tile arithmetic as k_conv_halo<128, 2, FUSE, ..., M16>: outputs are bit-identical,
// the sums differ in the order the tiles are added.  Plain forward (+ accumulator-mode statistics) and backward-data with
// the fused BatchNorm sums (accumulator mode; register layout of -DHALO_FDIR); coutp == 128.
// XF (round 6): the forward launch whose input is PReLU(BatchNorm(in)) with the coefficients derived from the producer's
// f64 sums in the prologue (msml_conv2d_bnin_acc): every wave normalises the image chunks it requested itself, one or two
// taps after the request (its own vmcnt wait orders them; the slab-switch barrier publishes them), and writes the tile's
// own pixels through for the weight gradient -- k_conv_halo's XF arithmetic, here ALSO for the next tile's first image,
// which is requested during this tile's last slab.  The one-round kernel paid this transform with its 14-us tile overhead
// on top (128 @ 28x28: bn 21 + conv 75-85 us as two launches, 96 us fused there); VERDICT r5 item 1.
// (Tried and dropped: the write-through as always-issued buffer stores with a counted `vmcnt(4)` wait in the stage behind
// the transform, so that the stores need not be acknowledged before the next MFMA phase -- three interleaved pairs on one
// box: conv family 14.09 / 14.01 / 13.94 ms plain, 14.13 / 14.09 / 14.04 counted; the four extra out-of-range stores per
// wave and slab cost more than the wait they save.)
template <bool FUSE, bool XF = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv_halo_p(const ConvHaloArgs p, const int ntiles, const unsigned int out_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PL2 = 4, PITCH = 16, BN = 128, KG = 4, NW = 8, NT = 512;
  constexpr int TW = 14, TH = 14, HPX = 16 << PL2, ABYTES = HPX * 128, NAJ = HPX / 8, NAI = NAJ / NW;
  constexpr int NG = 8, NGH = 4;                       // 16-pixel groups of one wave (at most), per pipeline phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                                     // [2][HPX][128 B]
  char* Bs = smem + 2 * ABYTES;                        // [NW][2][32][128 B]
  static_assert(!(FUSE && XF), "the input transform rides on the forward launch");
  float* xtab = reinterpret_cast<float*>(smem + 2 * ABYTES + NW * 8192);   // XF: [3][C] scale, shift, alpha
  MSML_LDS_REGION(As, 2 * ABYTES);
  MSML_LDS_REGION(Bs, NW * 8192);
  if (XF) MSML_LDS_REGION(xtab, 3 * p.C * 4);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  auto skey = [](int p_) { return p_ & 7; };
  const int kg = wave % KG, mg = wave / KG;
  const int i0 = mg * 4, nmt = mg == 0 ? 4 : 3, ng = 2 * nmt;
  const int tpi = p.tpy * p.tpx;
  const int l16 = lane & 15, q16 = lane >> 4;

  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.w_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)out_bytes, 0x00020000);

  auto calc_aoff = [&](int tile, unsigned int (&ao)[NAI]) {
    const int n = tile / tpi, trem = tile - n * tpi, ty = trem / p.tpx;
    const int y0 = ty * TH, x0 = (trem - ty * p.tpx) * TW;
#pragma unroll
    for (int i = 0; i < NAI; i++) {
      const int j = wave + i * NW;
      const int hp = j * 8 + (lane >> 3);
      const int logical = (lane & 7) ^ skey(hp);
      const int iy = y0 + (hp >> PL2) - 1, ix = x0 + (hp & (PITCH - 1)) - 1;
      const bool v = (tile < ntiles) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      ao[i] = v ? (unsigned int)((n * p.H + iy) * p.W + ix) * (unsigned int)(p.C * 2) + logical * 16u : HALO_OOB;
    }
  };
  auto issue_a = [&](const unsigned int (&ao)[NAI], int cs, int buf) {
    char* a = As + buf * ABYTES;
#pragma unroll
    for (int i = 0; i < NAI; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(a + (wave + i * NW) * 1024), 16,
                                               ao[i] == HALO_OOB ? HALO_OOB : ao[i] + cs * 128u, 0, 0, 0);
  };
  // XF: slab cs of the image whose request offsets are `ao`, in buffer `buf`: this wave's chunks (hp & 7 == (lane >> 3) & 7
  // for every chunk of a lane: ONE channel chunk per lane and slab, coefficients read from the table once); padding stays
  // zero; the pixels the tile owns (not its halo) are written through
  auto xform = [&](const unsigned int (&ao)[NAI], int cs, int buf) {
    char* a = As + buf * ABYTES;
    const bool has_alpha = p.xin.alpha != nullptr;
    f32x4 rsc[2], rsh[2], ral[2];
    const float* tb = xtab + cs * 64 + (((lane & 7) ^ ((lane >> 3) & 7)) << 3);
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      rsc[hf] = *reinterpret_cast<const f32x4*>(tb + hf * 4);
      rsh[hf] = *reinterpret_cast<const f32x4*>(tb + p.C + hf * 4);
      ral[hf] = *reinterpret_cast<const f32x4*>(tb + 2 * p.C + hf * 4);
    }
#pragma unroll
    for (int i = 0; i < NAI; i++) {
      const int j = wave + i * NW;
      const int hp = j * 8 + (lane >> 3);
      if (ao[i] != HALO_OOB) {
        bn_in_chunk_r(a + j * 1024 + lane * 16, rsc, rsh, ral, has_alpha);
        const int hy = hp >> PL2, hx = hp & (PITCH - 1);
        if (p.xin.store && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW)
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.xin.store) + ao[i] + cs * 128u) =
              *reinterpret_cast<const u32x4*>(a + j * 1024 + lane * 16);
      }
    }
  };
  unsigned int boffg[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = i * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ skey(row);
    boffg[i] = (unsigned int)((kg * 32 + row) * p.Ktot) * 2u + logical * 16u;
  }
  auto issue_b = [&](int cs, int tap, int buf) {
    char* b = Bs + wave * 8192 + buf * 4096;
    const unsigned int col = (unsigned int)(tap * p.C + cs * 64) * 2u;
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(b + i * 1024), 16, boffg[i] + col, 0, 0, 0);
  };
  int bfr16[2][2];
#pragma unroll
  for (int g = 0; g < 2; g++)
#pragma unroll
    for (int w = 0; w < 2; w++) {
      const int row = 16 * g + l16;
      bfr16[g][w] = wave * 8192 + row * 128 + (((4 * w + q16) ^ skey(row)) << 4);
    }

  // per-workgroup sums, in registers over all its tiles
  const int cdir = kg * 32 + (q16 & 1) * 16 + (q16 >> 1) * 8;   // the lane's 8 channels after the pair swap
  const int kb = kg * 32 + 4 * q16;                    // ... before it: kb + 16 g + j
  f32x4 s1[2], s2[2];
#pragma unroll
  for (int g = 0; g < 2; g++) s1[g] = s2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bq[3][8];
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) bq[q][j] = 0.f;
  BnbCoef bk;
  if (FUSE) bk = bnb_load_coef(p.bnb, cdir);

  const int nslab = p.C >> 6, nstage = nslab * 9;
  int tile = blockIdx.x;
  unsigned int aoff[NAI], aoffn[NAI];
  calc_aoff(tile, aoff);
  issue_a(aoff, 0, 0);
  issue_b(0, 0, 0);
  if (XF) bn_in_fill_acc(p.xin, xtab, p.C, t, NT, blockIdx.x == 0);
  __syncthreads();                                     // (drains vmcnt first)
  if (XF) {
    xform(aoff, 0, 0);
    __syncthreads();
  }
#if defined(HALO_PRIO)
  if (HALO_PRIO == 1 ? (wave >= 4) : (wave < 4)) __builtin_amdgcn_s_setprio(1);
#endif
  int gs = 0, gq = 0;                                  // running slab / stage counters: image buffer gs & 1, weight slot gq & 1
  bool first = true;
#ifdef HALO_TRACE
  int tix = 0;
#endif
  HALO_STAMP(9);
#ifdef HALO_TRACE
  if (wave == 0 && lane == 0) g_halo_trace[4096 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();   // every workgroup: start ...
#endif
  for (; tile < ntiles; tile += gridDim.x) {
    const int n = tile / tpi, trem = tile - n * tpi, ty = trem / p.tpx;
    const int y0 = ty * TH, x0 = (trem - ty * p.tpx) * TW;
    auto pix_ok = [&](int m) { return ((m & 15) < TW) & (x0 + (m & 15) < p.W) & (y0 + (m >> 4) < p.H); };
    auto pix_off = [&](int m) { return (unsigned int)((n * p.H + y0 + (m >> 4)) * p.W + x0 + (m & 15)) * (unsigned int)BN; };
    calc_aoff(tile + gridDim.x, aoffn);                // (a tile past the end: every request out of range)
    f32x4 acc4[NG][2];
#pragma unroll
    for (int i = 0; i < NG; i++)
#pragma unroll
      for (int g = 0; g < 2; g++) acc4[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cs = 0, tr = 0, ts = 0;
    for (int q = 0; q < nstage; q++, gq++) {
      int ncs = cs, ntr = tr, nts = ts + 1;
      if (nts == 3) { nts = 0; ntr++; }
      if (ntr == 3) { ntr = 0; ncs++; }
      HALO_STAMP(1);
      // this wave's weights of stage q have landed.  Stage 0 of a later tile: they were requested BEFORE the previous
      // tile's epilogue, whose NG stores (always issued, out of range when masked) are the only younger operations.
      // (Also requesting stage 1's weights before the epilogue, so that stage 1 does not wait for the stores to be
      // acknowledged either -- counted waits of 12 -- measured no faster alone and slower in the step; not kept.)
#ifdef HALO_ABLATE_EPILOGUE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      if (q == 0 && !first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      HALO_STAMP(2);
      if (q + 1 < nstage) issue_b(ncs, ntr * 3 + nts, (gq + 1) & 1);
      else issue_b(0, 0, (gq + 1) & 1);                // the next tile's first stage (past the end: a harmless re-read)
      if ((tr | ts) == 0) {                            // first tap of a slab: the next slab's image, this tile's or the next one's
        if (cs + 1 < nslab) issue_a(aoff, cs + 1, (gs + 1) & 1);
        else issue_a(aoffn, 0, (gs + 1) & 1);
      }
      // XF: the image requested one (waves 0-3) / two (their SIMD partners 4-7) stages ago has landed for this wave (the
      // wait above): normalise this wave's chunks of it, one wave's VALU beside its partner's MFMAs
      if (XF && tr == 0 && ts == (wave < 4 ? 1 : 2)) {
        if (cs + 1 < nslab) xform(aoff, cs + 1, (gs + 1) & 1);
        else xform(aoffn, 0, (gs + 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      HALO_STAMP(3);
      const int r = p.flip ? 2 - tr : tr, s = p.flip ? 2 - ts : ts;
      const int arow = l16 + s, asw = skey(arow);
      const char* Arow = As + (gs & 1) * ABYTES + (((r << PL2) + i0 * 32) * 128) + arow * 128;
      const char* B = Bs + (gq & 1) * 4096;
      u32x4 a16[2][NGH], b16[2][2];
#pragma unroll
      for (int j = 0; j < NGH; j++)
        if (j < ng) a16[0][j] = *reinterpret_cast<const u32x4*>(Arow + ((q16 ^ asw) << 4) + j * 2048);
#pragma unroll
      for (int g = 0; g < 2; g++) b16[0][g] = *reinterpret_cast<const u32x4*>(B + bfr16[g][0]);
#pragma unroll
      for (int ph = 0; ph < 4; ph++) {
        const int cb = ph & 1, nb = cb ^ 1, w = ph >> 1, hf = ph & 1;
        if (ph + 1 < 4) {
          const int nw = (ph + 1) >> 1, nhf = (ph + 1) & 1;
          const int ao = ((4 * nw + q16) ^ asw) << 4;
#pragma unroll
          for (int j = 0; j < NGH; j++)
            if (nhf * NGH + j < ng) a16[nb][j] = *reinterpret_cast<const u32x4*>(Arow + ao + (nhf * NGH + j) * 2048);
          if (nhf == 0) {
#pragma unroll
            for (int g = 0; g < 2; g++) b16[nw & 1][g] = *reinterpret_cast<const u32x4*>(B + bfr16[g][nw]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NGH; j++)
          if (hf * NGH + j < ng) {
#pragma unroll
            for (int g = 0; g < 2; g++)
              acc4[hf * NGH + j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, b16[w & 1][g]), __builtin_bit_cast(bf16x8, a16[cb][j]),
                  acc4[hf * NGH + j][g], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      HALO_STAMP(4);
      if (ncs != cs) {                                 // slab switch (also the tile's end): the next image has landed
        __syncthreads();                               // everywhere, and nobody reads the buffer the one after overwrites
        gs++;
        HALO_STAMP(5);
      }
      cs = ncs; tr = ntr; ts = nts;
    }
    first = false;
#ifdef HALO_ABLATE_EPILOGUE
    if (p.N >= 0) {
#pragma unroll
      for (int i = 0; i < NAI; i++) aoff[i] = aoffn[i];
      continue;
    }
#endif
    // ---- epilogue from registers: pair swap -> 8 contiguous channels per lane, one 16-B store per pixel group ----
    u32x4 xr[NG];
    if (FUSE) {
#pragma unroll
      for (int k = 0; k < NG; k++) {
        const int m = i0 * 32 + k * 16 + l16;
        xr[k] = (k < ng && pix_ok(m)) ? *reinterpret_cast<const u32x4*>(p.bnb.x + pix_off(m) + cdir) : u32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int jg = 0; jg < NG; jg++) {
      const int m = i0 * 32 + jg * 16 + l16;
      const bool valid = (jg < ng) & pix_ok(m);
      u32x2 pk[2];
#pragma unroll
      for (int g = 0; g < 2; g++) {
        const f32x4 z = acc4[jg][g];
        if (!FUSE && valid) {
          s1[g] += z;
          s2[g] += z * z;
        }
        pk[g][0] = (unsigned int)f2bf(z[0]) | ((unsigned int)f2bf(z[1]) << 16);
        pk[g][1] = (unsigned int)f2bf(z[2]) | ((unsigned int)f2bf(z[3]) << 16);
      }
      u32x4 o16;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        auto sw = __builtin_amdgcn_permlane16_swap(pk[0][e], pk[1][e], false, false);
        o16[e] = sw[0]; o16[2 + e] = sw[1];
      }
      // (buffer store: a masked lane stores out of range = dropped; the count of stores per tile is fixed, see the wait)
      __builtin_amdgcn_raw_buffer_store_b128(o16, rs_out, valid ? (pix_off(m) + cdir) * 2u : HALO_OOB, 0, 0);
      if (FUSE && valid)
        bnb_accum(bk, p.bnb.alpha != nullptr, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&o16)),
                  load8<unsigned short>(reinterpret_cast<const unsigned short*>(&xr[jg])), bq);
    }
#pragma unroll
    for (int i = 0; i < NAI; i++) aoff[i] = aoffn[i];
    HALO_STAMP(6);
  }
  // ---- the workgroup's sums: the lanes that share a channel meet in LDS (every image / ring request has landed) --
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef HALO_TRACE
  if (wave == 0 && lane == 0) g_halo_trace[4096 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();  // ... and end of its tiles
#endif
  if (FUSE) {
    constexpr int G = 16 * 2;                          // lanes (mg, l16) share an 8-channel chunk
    float* red = reinterpret_cast<float*>(smem);
    MSML_LDS_REGION(red, G * 3 * BN * 4);
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int j = 0; j < 8; j++) red[((mg * 16 + l16) * 3 + q) * BN + cdir + j] = bq[q][j];
    __syncthreads();
    for (int i = t; i < 3 * BN; i += NT) {
      const int q = i / BN, c = i % BN;
      float sum = 0.f;
      for (int g = 0; g < G; g++) sum += red[(g * 3 + q) * BN + c];
      bnb_emit(p.bnb.partial, 1, blockIdx.x, q, BN, c, sum);
    }
  } else if (p.stats) {
    float* red = reinterpret_cast<float*>(smem) + wave * 64 * 33;
    MSML_LDS_REGION(smem, NW * 64 * 33 * 4);
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        red[lane * 33 + g * 4 + j] = s1[g][j];
        red[lane * 33 + 16 + g * 4 + j] = s2[g][j];
      }
    __syncthreads();
    if (mg == 0) {                                     // channel kl = 16 g + 4 q + j lives in the 16 lanes 16 q + rr of both row groups
      const int which = lane >> 5, kl = lane & 31;
      const int k = which * 16 + (kl >> 4) * 4 + (kl & 3), qq = (kl >> 2) & 3;
      float sum = 0.f;
#pragma unroll
      for (int gm = 0; gm < 2; gm++)
#pragma unroll 8
        for (int rr = 0; rr < 16; rr++) sum += red[gm * KG * 64 * 33 + (qq * 16 + rr) * 33 + k];
      stats_emit(p.stats, 1, blockIdx.x, which, BN, kg * 32 + kl, sum);
    }
  }
#endif
}

template <int BN, int NWM, bool FUSE, bool XF = false, bool X3 = false, bool M16 = false, bool XB = false, bool R15 = false,
          bool BREG = false>
static void launch_halo(ConvHaloArgs& a, hipStream_t st) {
  size_t lds = 2 * (size_t)256 * 128 + 8 * 8192;      // two halo images + eight private weight rings
  size_t olds = (size_t)224 * (BN + 8) * 2;
  if (olds > lds) lds = olds;
  if (XF) lds += 3 * 1024 * sizeof(float);             // coefficient table, C <= 1024
  if (XB) lds += 7 * 512 * sizeof(float);              // backward coefficient table, C <= 512
  static std::once_flag attr_once;                     // (per template instantiation; launches come from
  std::call_once(attr_once, [&] {                      //  the forward thread AND the autograd thread)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo<BN, NWM, FUSE, XF, X3, M16, XB, R15, BREG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  dim3 grid(a.N * a.tpy * a.tpx, a.coutp / BN);
  k_conv_halo<BN, NWM, FUSE, XF, X3, M16, XB, R15, BREG><<<grid, dim3(512), lds, st>>>(a);
}

static int halo_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}

// Shapes of the persistent 128-channel tile (k_conv_halo_p): 128 output channels, 64 k input channels (ONE 64-channel slab
// included: 64 -> 128 @ 56x56, conv1 of the first block of layer2, backbones/frb/iresnet.py:166-170 -- the nine-stage K loop
// whose prologue and epilogue ruled the one-tile kernel out runs back to back there), at least two rounds of tiles.
// MSML_HALO_PERSIST=0 / MSML_HALO_NO_ONE_SLAB=1 (read per call: the tests compare): off / 128 input channels and more only.
int msml_conv_halo_persist_shape(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                                 int pad_h, int pad_w) {
  const char* pe = getenv("MSML_HALO_PERSIST");
  if (pe != nullptr && atoi(pe) == 0) return 0;
  if (getenv("MSML_NO_HALO_CONV") != nullptr || getenv("MSML_HALO_WIDE_ONLY") != nullptr) return 0;   // (the halo conv's own
  // off-switches cover this 128-channel instantiation too)
  if (R != 3 || S != 3 || stride != 1 || pad_h != 1 || pad_w != 1 || P != H || Q != W) return 0;
  if (coutp != 128 || kop < coutp || c0p % 64 != 0 || c0p < 64) return 0;
  if (c0p == 64 && getenv("MSML_HALO_NO_ONE_SLAB") != nullptr) return 0;
  const long tiles = (long)N * cdiv(H, 14) * cdiv(W, 14);
  if (tiles < 2L * halo_num_cus() || (long)N * H * W * 10 < tiles * 224 * 7) return 0;
  if ((long)N * H * W * c0p * 2 >= 0x70000000L || (long)N * H * W * 128 * 2 >= 0x70000000L || (long)kop * 9 * c0p * 2 >= 0x70000000L)
    return 0;
  return 1;
}

// Shape test shared by the dispatch and by msml_conv2d_kernel (the name query).
bool msml_conv_halo_applies(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                            int stride, int pad_h, int pad_w, bool want_stats) {
  static const bool off = getenv("MSML_NO_HALO_CONV") != nullptr;
  if (off) return false;
  static const bool wide_only = getenv("MSML_HALO_WIDE_ONLY") != nullptr;    // A/B switch
  if (wide_only && coutp % 256 != 0) return false;
  if (R != 3 || S != 3 || stride != 1 || pad_h != 1 || pad_w != 1 || P != H || Q != W) return false;
  if (c0p % 64 != 0 || c0p < 128 || coutp % 128 != 0 || kop < coutp) return false;   // (one 64-channel
  // slab = a 9-stage K loop: prologue and epilogue of the single resident workgroup dominate)
  const long tiles = (long)N * cdiv(H, 14) * cdiv(W, 14);
  if ((long)N * H * W * 10 < tiles * 224 * 7) return false;       // < 70 % real GEMM rows: im2col kernel wins
  const int srows = cdiv((long)N * P * Q, msml_conv_tile_m(coutp));
  if (want_stats && tiles > srows) return false;        // (no batch-size threshold: the kernel choice, and
  // with it the summation order, must not depend on N -- results are batch-composition independent)
  const long in_bytes = (long)N * H * W * c0p * 2, w_bytes = (long)kop * 9 * c0p * 2;
  return in_bytes < 0x70000000L && w_bytes < 0x70000000L;
}

// Tried first by msml_conv_fast_dispatch; false = shape not covered here.
bool msml_conv_halo_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out,
                             int coutp, float* stats, int N, int H, int W, int P, int Q, int R, int S,
                             int stride, int pad_h, int pad_w, int transposed, hipStream_t st,
                             const float* scale, const float* alpha, const void* residual, int res_first,
                             const BnBwdFuse* bnb, int* bnb_rows, const BnIn* xin, int x3, const BnBwdIn* bin) {
  static const int m16_ = getenv("MSML_HALO_M16") ? atoi(getenv("MSML_HALO_M16")) : 2;
  static const bool xfp_ = !(getenv("MSML_BNIN_ACC_PERSIST") && atoi(getenv("MSML_BNIN_ACC_PERSIST")) == 0);
  const bool pshape = m16_ >= 2 && !x3 && !bin && !bias && !scale && !alpha && !residual &&
                      (!xin || (xfp_ && xin->acc && !transposed && !bnb && stats && c0p <= 1024)) &&
                      (!stats || msml_tl_stats_acc) && (!bnb || bnb->acc) &&
                      msml_conv_halo_persist_shape(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w) != 0;
  if (!pshape && !msml_conv_halo_applies(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, stats != nullptr))
    return false;
  if (bin && (!bnb || !transposed || x3 || xin || c0p > 512)) return false;
  if (x3 && (bnb || xin || stats || transposed)) return false;
  if (bnb && (bias || scale || alpha || residual || stats)) return false;
  if (xin && (bnb || transposed || c0p > 1024)) return false;
  ConvHaloArgs a;
  a.tpy = cdiv(H, 14); a.tpx = cdiv(W, 14);
  const long tiles = (long)N * a.tpy * a.tpx;
  const long in_bytes = (long)N * H * W * c0p * 2, w_bytes = (long)kop * 9 * c0p * 2;
  a.in = (const unsigned short*)in0; a.in_bytes = (unsigned int)in_bytes; a.C = c0p;
  a.N = N; a.H = H; a.W = W; a.flip = transposed;
  a.wp = (const unsigned short*)wp; a.w_bytes = (unsigned int)w_bytes; a.Ktot = 9 * c0p;
  a.out = (unsigned short*)out; a.coutp = coutp;
  a.bias = bias; a.scale = scale; a.alpha = alpha; a.residual = (const unsigned short*)residual;
  a.res_first = res_first; a.stats = stats;
  a.bias9 = (x3 && bias) ? msml_tl_bias9 : 0;
  if (msml_tl_bias9 && !x3) return false;
  a.stats_rows = cdiv((long)N * P * Q, msml_conv_tile_m(coutp));
  a.stats_acc = stats ? msml_tl_stats_acc : 0;
  a.bnb = BnBwdFuse{};
  if (bnb) a.bnb = *bnb;
  a.xin = BnIn{nullptr, nullptr, nullptr};
  if (xin) a.xin = *xin;
  a.bin = BnBwdIn{};
  if (bin) a.bin = *bin;
  if (bnb_rows) *bnb_rows = (int)tiles;
  const bool wide = coutp % 256 == 0;
  // the 16x16x32 MFMA variant serves the plain forward / FUSE launches (round 4: +4...8 % on every shape, interleaved
  // A/B on one box, LDS conflicts 0.7 %).  MSML_HALO_M16=0 restores the 32x32x16 kernels, 1 limits
  // the variant to the 256-channel tile.
  static const int m16 = getenv("MSML_HALO_M16") ? atoi(getenv("MSML_HALO_M16")) : 2;
  if (bin) {                             // BatchNorm backward in the prologue: 16x16x32 FUSE instantiations only
#ifdef MSML_EXPERIMENTS
    if (wide) launch_halo<256, 1, true, false, false, true, true>(a, st);
    else launch_halo<128, 2, true, false, false, true, true>(a, st);
    return true;
#else
    return false;                        // (measured slower: `XB` is instantiated in experiment builds only, tools/build_variant.py)
#endif
  }
  // 128-channel tile, several rounds of tiles per launch (128 -> 128 @ 28x28, 64 -> 128 @ 56x56): the persistent kernel
  if (pshape) {
    const size_t lds = 2 * (size_t)256 * 128 + 8 * 8192;
    const int grid = halo_num_cus();
    const unsigned int out_bytes = (unsigned int)((long)N * H * W * 128 * 2);
    if (xin) {                           // BatchNorm + PReLU of the input formed in the prologue (accumulator mode)
      const size_t ldsx = lds + 3 * 1024 * sizeof(float);
      static std::once_flag once;
      std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo_p<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx);
      });
      k_conv_halo_p<false, true><<<dim3(grid), dim3(512), ldsx, st>>>(a, (int)tiles, out_bytes);
    } else if (bnb) {
      static std::once_flag once;
      std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo_p<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      });
      k_conv_halo_p<true><<<dim3(grid), dim3(512), lds, st>>>(a, (int)tiles, out_bytes);
    } else {
      static std::once_flag once;
      std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo_p<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      });
      k_conv_halo_p<false><<<dim3(grid), dim3(512), lds, st>>>(a, (int)tiles, out_bytes);
    }
    return true;
  }
  // the half-stage weight ring (template parameter R15) on the 256-channel tile; MSML_HALO_R15=0 (read per call: the tests
  // compare the two): the two-slot ring with full waits
  // (experiment builds only, -DMSML_EXPERIMENTS: measured slower)
#ifdef MSML_EXPERIMENTS
  const char* r15e = getenv("MSML_HALO_R15");
  const bool r15 = r15e != nullptr && atoi(r15e) != 0;
  if (m16 && !x3 && wide && r15) {
    if (xin) launch_halo<256, 1, false, true, false, true, false, true>(a, st);
    else if (bnb) launch_halo<256, 1, true, false, false, true, false, true>(a, st);
    else launch_halo<256, 1, false, false, false, true, false, true>(a, st);
    return true;
  }
#endif
  // weights in registers (template parameter BREG) on the 256-channel tile; MSML_HALO_BREG read per call (A/B in one process).
  // Measured neutral (round 6, tools/bench_breg.py: bit-identical, 256 @ 14x14 forward 59.5 -> 57.4 us, with the BatchNorm
  // prologue 60.7 -> 60.8, backward-data + sums 56.6 -> 56.2; the step 29.53 -> 29.53 ms, three interleaved pairs): the
  // weights' trip through LDS is not what the tile waits for.  Experiment builds only.
#ifdef MSML_EXPERIMENTS
  const char* brege = getenv("MSML_HALO_BREG");
  if (m16 && !x3 && wide && brege != nullptr && atoi(brege) != 0) {
    if (xin) launch_halo<256, 1, false, true, false, true, false, false, true>(a, st);
    else if (bnb) launch_halo<256, 1, true, false, false, true, false, false, true>(a, st);
    else launch_halo<256, 1, false, false, false, true, false, false, true>(a, st);
    return true;
  }
#endif
  if (m16 && !x3 && (wide || m16 >= 2)) {
    if (xin) {                           // (same tiling as the plain launch: the two stay bit-identical)
      if (wide) launch_halo<256, 1, false, true, false, true>(a, st);
      else launch_halo<128, 2, false, true, false, true>(a, st);
    } else if (bnb) {
      if (wide) launch_halo<256, 1, true, false, false, true>(a, st);
      else launch_halo<128, 2, true, false, false, true>(a, st);
    } else {
      if (wide) launch_halo<256, 1, false, false, false, true>(a, st);
      else launch_halo<128, 2, false, false, false, true>(a, st);
    }
    return true;
  }
  if (x3) {
    if (wide) launch_halo<256, 1, false, false, true>(a, st);
    else launch_halo<128, 2, false, false, true>(a, st);
  } else if (xin) {
    if (wide) launch_halo<256, 1, false, true>(a, st);
    else launch_halo<128, 2, false, true>(a, st);
  } else if (bnb) {
    if (wide) launch_halo<256, 1, true>(a, st);
    else launch_halo<128, 2, true>(a, st);
  } else {
    if (wide) launch_halo<256, 1, false>(a, st);
    else launch_halo<128, 2, false>(a, st);
  }
  return true;
}
