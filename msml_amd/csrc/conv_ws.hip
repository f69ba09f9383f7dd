// Weights-stationary persistent 3x3 / stride-1 / pad-1 convolution for bf16 NHWC maps with
// Cin = Cout = 64: the 112x112 and 56x56 levels of the FRB and the OSB encoder
// (backbones/frb/iresnet.py:40-67, backbones/osb/unet.py:80-91) -- forward, and backward-data with
// the tap walk flipped.  Same contract as msml_conv2d / msml_conv2d_fused / msml_conv2d_bnbwd.
//
// With 64 x 64 channels the whole packed weight (9 taps x 64 x 64 bf16 = 72 KB) fits in LDS, so
// one workgroup per CU loads it ONCE and then walks over 14 x 14 pixel tiles (halo image in LDS,
// as in conv_halo.hip): per tile only 32 KB of input enter LDS for 16.5 MFLOP (the im2col kernel
// fills 40 KB per 2.1 MFLOP at this width and is bound by that fill).  The next tile's image is
// fetched during the current tile's MFMAs (two image buffers); the bf16 output tile is transposed
// through the image buffer that was just consumed.  No barrier and no load wait inside a tile's
// 36 k-steps.  BatchNorm partial statistics (forward) and the fused BatchNorm backward-reduce
// (backward-data) accumulate in registers over all tiles of the workgroup and leave one partial
// row per workgroup.
#include <stdlib.h>

#include <mutex>

#include "common.h"

struct ConvWsArgs {
  const unsigned short* in;
  unsigned int in_bytes;
  int N, H, W, tpy, tpx, ntiles;
  int flip;
  const unsigned short* wp;      // [64][576] bf16, K order [tap][c]
  unsigned short* out;
  const float* bias;
  const float* scale;
  const float* alpha;
  const unsigned short* residual;
  int res_first;
  float* stats;
  int stats_rows;
  int stats_acc;      // accumulator mode (common.h)
  BnBwdFuse bnb;
  BnIn xin;           // xin.scale != nullptr: BatchNorm(+PReLU) applied to the input image in LDS (common.h)
};

#define WS_OOB 0x78000000u

typedef __attribute__((address_space(3))) void* lptr_t;

// XF: forward launch whose input is PReLU(in * xin.scale + xin.shift), applied to each image in LDS.
// M16: v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (see conv_halo.hip): 8 accumulator tiles of 16 channels x 16
// pixels per wave, the 18 32-deep windows of a tile as one software pipeline; chunk key p & 7.
template <bool FUSE, bool XF = false, bool M16 = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_conv_ws(const ConvWsArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = 64, PL2 = 4, PITCH = 16, TW = 14, TH = 14, BM = 224, NT = 512;
  constexpr int WBYTES = 9 * 64 * 128, ABYTES = 256 * 128;
  constexpr int OP = C + 8;                            // transposed output tile pitch (elements)
  constexpr int C8 = C / 8, ITERS = (BM * C8 + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem;                                     // [9 taps][64 rows][128 B]
  char* As = smem + WBYTES;                            // [2][256 px][128 B]
  float* xtab = reinterpret_cast<float*>(smem + WBYTES + 2 * ABYTES + 512);   // XF: [3][64]
  MSML_LDS_REGION(Ws, WBYTES);
  MSML_LDS_REGION(As, 2 * ABYTES + 2 * 128);           // (+ the two pixels the padding rows read past an image)
  if (XF) MSML_LDS_REGION(xtab, 3 * C * 4);
  MSML_LDS_REGION(As, BM * OP * 2);                    // the transposed output tile goes through an image buffer
  MSML_LDS_REGION(As + ABYTES, BM * OP * 2);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kg = wave & 1, mg = wave >> 1;             // channel group (32), pixel-row group
  const int i0 = 2 * mg, nmt = mg < 3 ? 2 : 1;         // this wave's 32-pixel tiles [i0, i0 + nmt)
  const int r32 = lane & 31, h = lane >> 5;
  const int l16 = lane & 15, q16 = lane >> 4;          // M16: row inside a 16-group, 8-deep k block
  
  auto skey = [](int p_) { return M16 ? (p_ & 7) : ((p_ >> 1) & 7); };      // chunk swizzle of LDS row p (conv_halo.hip)
  const int tpi = p.tpy * p.tpx;

  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 64 * 576 * 2, 0x00020000);

  // ---- weights: 72 pieces of 8 rows x 128 B; piece j = tap * 8 + row block
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int j = wave + i * 8, tap = j >> 3, row = (j & 7) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ skey(row);
    const unsigned int off = (unsigned int)(row * 576 + tap * 64) * 2u + logical * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(Ws + j * 1024), 16, off, 0, 0, 0);
  }
  // ---- halo image of a tile: LDS pixel hp = hy * 16 + hx <- input (y0 + hy - 1, x0 + hx - 1)
  auto issue_a = [&](int tile, int buf) {
    const int n = tile / tpi, trem = tile - n * tpi, ty = trem / p.tpx;
    const int y0 = ty * TH, x0 = (trem - ty * p.tpx) * TW;
    char* a = As + buf * ABYTES;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int j = wave + i * 8;
      const int hp = j * 8 + (lane >> 3);
      const int logical = (lane & 7) ^ skey(hp);
      const int iy = y0 + (hp >> PL2) - 1, ix = x0 + (hp & (PITCH - 1)) - 1;
      const bool v = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const unsigned int off = v ? (unsigned int)((n * p.H + iy) * p.W + ix) * (unsigned int)(C * 2) + logical * 16u : WS_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(a + j * 1024), 16, off, 0, 0, 0);
    }
  };

  // XF: every wave normalises the chunks it DMA'd itself (after its own vmcnt wait); padding stays zero.  A lane's
  // chunks all hold the SAME 8 channels (chunk key (hp >> 1) & 7 = (4 (wave & 1) + (lane >> 4)) & 7, M16: hp & 7 =
  // (lane >> 3) & 7 -- neither depends on the DMA pass i), so its coefficients live in registers for the whole kernel
  // (xcoef, read from the LDS table once).  Accumulator mode (xin.acc): the transformed pixels this tile OWNS (not its
  // halo) are also written to xin.store -- the weight gradient reads that tensor (conv_halo.hip, XF).
  f32x4 xsc[2], xsh[2], xal[2];
  auto xform = [&](int tile, int buf) {
    const int n = tile / tpi, trem = tile - n * tpi, ty = trem / p.tpx;
    const int y0 = ty * TH, x0 = (trem - ty * p.tpx) * TW;
    char* a = As + buf * ABYTES;
    const bool has_alpha = p.xin.alpha != nullptr;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int j = wave + i * 8;
      const int hp = j * 8 + (lane >> 3);
      const int logical = (lane & 7) ^ skey(hp);
      const int hy = hp >> PL2, hx = hp & (PITCH - 1);
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      if (((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W)) {
        bn_in_chunk_r(a + j * 1024 + lane * 16, xsc, xsh, xal, has_alpha);
        if (p.xin.store && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW)
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.xin.store) +
                                    (size_t)((n * p.H + iy) * p.W + ix) * (size_t)(C * 2) + logical * 16u) =
              *reinterpret_cast<const u32x4*>(a + j * 1024 + lane * 16);
      }
    }
  };

  // fragment offsets (see conv_halo.hip): weights row kg * 32 + r32, pixels (i0 + i) * 32 + r32 + tap offset
  int bfr[4];
  {
    const int row = kg * 32 + r32;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) bfr[kk] = row * 128 + (((kk * 2 + h) ^ ((row >> 1) & 7)) << 4);
  }
  int bfr16[2][2];                                     // M16: weights row kg * 32 + 16 g + l16, chunk 4 w + q16
#pragma unroll
  for (int g = 0; g < 2; g++)
#pragma unroll
    for (int w = 0; w < 2; w++) {
      const int row = kg * 32 + 16 * g + l16;
      bfr16[g][w] = row * 128 + (((4 * w + q16) ^ skey(row)) << 4);
    }
  // this lane's output channels: kb + 8 g + j, g < 4 (M16: kb + 16 g + j, g < 2)
  const int kb = M16 ? kg * 32 + 4 * q16 : kg * 32 + 4 * h;
  const int c8 = t % C8;

  // epilogue coefficients and the accumulators that live across all tiles of this workgroup
  const bool act_here = !FUSE && p.alpha && !(p.residual && p.res_first);
  f32x4 bv[4], sv[4], av[4], s1[4], s2[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int col = kb + (M16 ? 16 * (g & 1) : 8 * g);
    bv[g] = (!FUSE && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    sv[g] = (!FUSE && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    av[g] = act_here ? *reinterpret_cast<const f32x4*>(p.alpha + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    s1[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  BnbCoef bk;
  float bq[3][8];
  // (FUSE on the 16x16x32 tiling: the sums are taken in the register layout of the direct store -- after the permlane swap
  // a lane holds channels cdir .. cdir + 7 of its pixel -- so its coefficients are those of that chunk)
  const int cdir = kg * 32 + (q16 & 1) * 16 + (q16 >> 1) * 8;
#ifdef WS_NO_FDIR
  if (FUSE) bk = bnb_load_coef(p.bnb, c8 * 8);
#else
  if (FUSE) bk = bnb_load_coef(p.bnb, M16 ? cdir : c8 * 8);
#endif
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) bq[q][j] = 0.f;

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue_a(tile, 0);
  if (XF) {
    if (p.xin.acc) bn_in_fill_acc(p.xin, xtab, C, t, NT, blockIdx.x == 0);
    else bn_in_fill(p.xin, xtab, 0, C, t, NT);
  }
  __syncthreads();                                     // weights + first image landed (drains vmcnt)
  if (XF) {
    const int ch = ((lane & 7) ^ skey(wave * 8 + (lane >> 3))) << 3;
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      xsc[hf] = *reinterpret_cast<const f32x4*>(xtab + ch + hf * 4);
      xsh[hf] = *reinterpret_cast<const f32x4*>(xtab + C + ch + hf * 4);
      xal[hf] = *reinterpret_cast<const f32x4*>(xtab + 2 * C + ch + hf * 4);
    }
    if (tile < p.ntiles) xform(tile, 0);
    __syncthreads();
  }

  // ---- direct epilogue of the 16x16x32 tiling (round 5): forward launches without a residual and backward-data launches
  // with the fused BatchNorm sums store straight from registers -- v_permlane16_swap leaves every lane with 8 contiguous
  // channels of its pixel (conv_halo.hip): no LDS transpose, one barrier per tile.  -DWS_STAGGER (tried, not faster): waves
  // 4-7 (the SIMD partners of 0-3) keep their accumulators across the barrier and store tile t at the top of iteration
  // t + 1, beside their partners' MFMAs (MI355X_MICROARCH.md, two waves per SIMD, item 9); bit-identical either way.
  f32x16 acc[2];
  f32x4 acc4[4][2];                                    // M16: [16-pixel group][channel half]
  u32x4 xr[ITERS];
#ifdef WS_NO_FDIR
  constexpr bool FDIR = false;                         // (A/B build)
#else
  constexpr bool FDIR = M16 && FUSE;
#endif
  auto epi_dir = [&](int tl) {
    const int n = tl / tpi, trem = tl - n * tpi, ty = trem / p.tpx;
    const int y0 = ty * TH, x0 = (trem - ty * p.tpx) * TW;
    auto pix_ok = [&](int m) { return ((m & 15) < TW) & (x0 + (m & 15) < p.W) & (y0 + (m >> 4) < p.H); };
    auto pix_off = [&](int m) { return ((long)(n * p.H + y0 + (m >> 4)) * p.W + x0 + (m & 15)) * C; };
#pragma unroll
    for (int jg = 0; jg < 4; jg++) {
      const int m = i0 * 32 + jg * 16 + l16;
      const bool valid = (jg < 2 * nmt) & pix_ok(m);
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
            if (valid) {
              s1[g][j] += z;
              s2[g][j] += z * z;
            }
          }
          v[j] = z;
        }
        pk[g][0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
        pk[g][1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
      }
      u32x4 o16;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        auto sw = __builtin_amdgcn_permlane16_swap(pk[0][e], pk[1][e], false, false);
        o16[e] = sw[0]; o16[2 + e] = sw[1];
      }
      if (valid) {
#ifndef WS_ABLATE_STORE
        *reinterpret_cast<u32x4*>(p.out + pix_off(m) + cdir) = o16;
#endif
        if constexpr (FDIR)
          bnb_accum(bk, p.bnb.alpha != nullptr, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&o16)),
                    load8<unsigned short>(reinterpret_cast<const unsigned short*>(&xr[jg])), bq);
      }
    }
  };
  const bool dir = M16 && (FDIR || (!FUSE && p.residual == nullptr));
#ifdef WS_STAGGER
  const bool late = dir && wave >= 4;
#else
  const bool late = false;     // measured (round 5, one box): staggered 398 / 89 us forward, 95.4 us fused backward-data against
                               // 390 / 87 / 89.5 us unstaggered -- the late waves' image requests start later; build switch only
#endif
  int ptile = -1;

  for (int it = 0; tile < p.ntiles; it++, tile += gridDim.x) {
    const int cur = it & 1;
    if (late && ptile >= 0) epi_dir(ptile);            // (before this tile's loads of xr and the zeroing of the accumulators)
    const int n = tile / tpi, trem = tile - n * tpi, ty = trem / p.tpx;
    const int y0 = ty * TH, x0 = (trem - ty * p.tpx) * TW;
    auto pix_ok = [&](int m) { return ((m & 15) < TW) & (x0 + (m & 15) < p.W) & (y0 + (m >> 4) < p.H); };
    auto pix_off = [&](int m) { return ((long)(n * p.H + y0 + (m >> 4)) * p.W + x0 + (m & 15)) * C; };
    // next tile's image (its buffer was the transpose tile of the previous iteration: the barrier
    // that closes an iteration orders those reads before this write)
#ifndef WS_ABLATE_LOADS
    if (tile + (int)gridDim.x < p.ntiles) issue_a(tile + gridDim.x, cur ^ 1);
#endif
    if (FDIR) {                                        // saved BatchNorm input in the direct-store layout
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int m = i0 * 32 + k * 16 + l16;
        xr[k] = (k < 2 * nmt && pix_ok(m)) ? *reinterpret_cast<const u32x4*>(p.bnb.x + pix_off(m) + cdir) : u32x4{0, 0, 0, 0};
      }
    } else if (FUSE) {                                 // saved BatchNorm input of this thread's chunks
#pragma unroll
      for (int k = 0; k < ITERS; k++) {
        const int idx = t + k * NT, m = idx / C8;
        xr[k] = (idx < BM * C8 && pix_ok(m)) ? *reinterpret_cast<const u32x4*>(p.bnb.x + pix_off(m) + c8 * 8)
                                              : u32x4{0, 0, 0, 0};
      }
    }
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int g = 0; g < 2; g++) acc4[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (M16) {
      // 9 taps x two 32-deep windows = 18 pipeline steps; the fragments of step k + 1 (4 pixel groups + 2 weight
      // halves) are requested before the 8 MFMAs of step k -- weights and image are resident, nothing to wait for
      const int ng = 2 * nmt;
      const char* Abase = As + cur * ABYTES + i0 * 32 * 128;
      u32x4 a16[2][4], b16[2][2];
      auto frags = [&](int step, u32x4 (&a)[4], u32x4 (&b)[2]) {
        const int tap = step >> 1, w = step & 1;
        const int tr = tap / 3, ts = tap - tr * 3;
        const int r = p.flip ? 2 - tr : tr, sft = p.flip ? 2 - ts : ts;
        const int arow = l16 + sft;
        const char* Arow = Abase + ((r << PL2) + arow) * 128 + (((4 * w + q16) ^ skey(arow)) << 4);
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (j < ng) a[j] = *reinterpret_cast<const u32x4*>(Arow + j * 2048);
#pragma unroll
        for (int g = 0; g < 2; g++) b[g] = *reinterpret_cast<const u32x4*>(Ws + tap * 8192 + bfr16[g][w]);
      };
      frags(0, a16[0], b16[0]);
#pragma unroll
      for (int step = 0; step < 18; step++) {
        const int cb = step & 1, nb = cb ^ 1;
        if (XF && step == (wave < 4 ? 6 : 12) && tile + (int)gridDim.x < p.ntiles) {   // next image: requested at the top of this tile;
          // waves 4-7 (the SIMD partners of 0-3) transform six steps later: one wave's VALU beside the other's MFMAs
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          xform(tile + gridDim.x, cur ^ 1);
        }
#ifdef WS_ABLATE_READS
        if (step + 1 < 18) { a16[nb][0] = a16[cb][0]; a16[nb][1] = a16[cb][1]; a16[nb][2] = a16[cb][2]; a16[nb][3] = a16[cb][3];
                             b16[nb][0] = b16[cb][0]; b16[nb][1] = b16[cb][1]; }   // (timing build: fragments read once per tile)
#else
        if (step + 1 < 18) frags(step + 1, a16[nb], b16[nb]);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (j < ng) {
#pragma unroll
            for (int g = 0; g < 2; g++)
              acc4[j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b16[cb][g]),
                                                                   __builtin_bit_cast(bf16x8, a16[cb][j]), acc4[j][g],
                                                                   0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    const char* Abase = As + cur * ABYTES + i0 * 32 * 128;
    u32x4 a[2][2], b[2];
    for (int tap = 0; tap < 9; tap++) {
      const int tr = tap / 3, ts = tap - tr * 3;
      const int r = p.flip ? 2 - tr : tr, s = p.flip ? 2 - ts : ts;
      const int arow = r32 + s, asw = (arow >> 1) & 7;
      const char* Arow = Abase + ((r << PL2) + arow) * 128;
      const char* B = Ws + tap * 8192;
      if (XF && tap == (wave < 4 ? 3 : 6) && tile + (int)gridDim.x < p.ntiles) {   // next image: requested >= 3 taps ago; waves
        // 4-7 (the SIMD partners of 0-3) transform three taps later: one wave's VALU beside the other's MFMAs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xform(tile + gridDim.x, cur ^ 1);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
        if (i < nmt) a[0][i] = *reinterpret_cast<const u32x4*>(Arow + ((h ^ asw) << 4) + i * 4096);
      b[0] = *reinterpret_cast<const u32x4*>(B + bfr[0]);
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int cb = kk & 1, nb = cb ^ 1;
        if (kk + 1 < 4) {
          const int ao = (((kk + 1) * 2 + h) ^ asw) << 4;
#pragma unroll
          for (int i = 0; i < 2; i++)
            if (i < nmt) a[nb][i] = *reinterpret_cast<const u32x4*>(Arow + ao + i * 4096);
          b[nb] = *reinterpret_cast<const u32x4*>(B + bfr[kk + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; i++)
          if (i < nmt)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[cb]),
                                                             __builtin_bit_cast(bf16x8, a[cb][i]), acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
    __syncthreads();     // every wave is done with image `cur`; the next image has landed (vmcnt drained)

    // ---- epilogue: affine / PReLU / statistics in registers, transpose through the consumed image
    unsigned short* otile = reinterpret_cast<unsigned short*>(As + cur * ABYTES);
    // forward launches without a residual and fused-BatchNorm backward-data launches of the 16x16x32 tiling store
    // straight from registers (epi_dir above); waves 4-7 do so at the top of the NEXT iteration (stagger)
    if (dir) {
      if (!late) epi_dir(tile);
      ptile = tile;
      continue;
    }
    if constexpr (M16) {
#pragma unroll
      for (int jg = 0; jg < 4; jg++) {
        if (jg >= 2 * nmt) break;
        const int m = i0 * 32 + jg * 16 + l16;
        const bool valid = pix_ok(m);
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
          u32x2 pk;
          pk[0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
          pk[1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
          *reinterpret_cast<u32x2*>(otile + m * OP + kb + 16 * g) = pk;
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (i >= nmt) break;
      const int m = (i0 + i) * 32 + r32;
      const bool valid = pix_ok(m);
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
        u32x2 pk;
        pk[0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
        pk[1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
        *reinterpret_cast<u32x2*>(otile + m * OP + kb + 8 * g) = pk;
      }
    }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
      const int idx = t + k * NT, m = idx / C8;
      if (idx < BM * C8 && pix_ok(m)) {
        u32x4 v = *reinterpret_cast<const u32x4*>(otile + m * OP + c8 * 8);
        const long o = pix_off(m) + c8 * 8;
        if (!FUSE && p.residual) {
          Vec8 a8 = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v));
          Vec8 r8 = load8<unsigned short>(p.residual + o);
#pragma unroll
          for (int j = 0; j < 8; j++) {
            float z = a8.v[j] + r8.v[j];
            if (p.res_first && p.alpha) z = z > 0.f ? z : z * p.alpha[c8 * 8 + j];
            a8.v[j] = z;
          }
          store8<unsigned short>(reinterpret_cast<unsigned short*>(&v), a8);
        }
        *reinterpret_cast<u32x4*>(p.out + o) = v;
        if (FUSE)
          bnb_accum(bk, p.bnb.alpha != nullptr, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v)),
                    load8<unsigned short>(reinterpret_cast<const unsigned short*>(&xr[k])), bq);
      }
    }
    __syncthreads();     // transpose tile read out: its buffer may take the image after next
  }

  if (late && ptile >= 0) epi_dir(ptile);
  if (dir) __syncthreads();                            // (the end-of-kernel reductions below reuse LDS the last tile's MFMAs read)

  // ---- one partial row per workgroup (the weight region of LDS is free now)
  if (FUSE) {
    constexpr int G = NT / C8;                         // 64 threads share a channel chunk
    float* red = reinterpret_cast<float*>(smem);
    MSML_LDS_REGION(red, G * 3 * C * 4);
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
#ifndef WS_NO_FDIR
        if constexpr (M16) red[((((wave >> 1) * 16 + l16)) * 3 + q) * C + cdir + j] = bq[q][j];   // 64 lanes share a chunk
        else
#endif
          red[((t / C8) * 3 + q) * C + c8 * 8 + j] = bq[q][j];
      }
    __syncthreads();
    for (int i = t; i < 3 * C; i += NT) {
      const int q = i / C, c = i % C;
      float sum = 0.f;
      for (int g = 0; g < G; g++) sum += red[(g * 3 + q) * C + c];
      bnb_emit(p.bnb.partial, p.bnb.acc, blockIdx.x, q, C, c, sum);
    }
  }
  if (!FUSE && p.stats) {
    float* red = reinterpret_cast<float*>(smem) + wave * 64 * 33;
    MSML_LDS_REGION(smem, 8 * 64 * 33 * 4);
#pragma unroll
    for (int g = 0; g < (M16 ? 2 : 4); g++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        red[lane * 33 + g * 4 + j] = s1[g][j];
        red[lane * 33 + 16 + g * 4 + j] = s2[g][j];
      }
    __syncthreads();
    if (mg == 0) {                                     // waves 0 / 1 add the four pixel-row groups
      const int which = lane >> 5, kl = lane & 31;
      float sum = 0.f;
      if constexpr (M16) {                             // channel kl = 16 g + 4 q + j lives in the 16 lanes 16 q + rr
        const int k = which * 16 + (kl >> 4) * 4 + (kl & 3), qq = (kl >> 2) & 3;
#pragma unroll
        for (int gm = 0; gm < 4; gm++)
#pragma unroll 8
          for (int rr = 0; rr < 16; rr++) sum += red[gm * 2 * 64 * 33 + (qq * 16 + rr) * 33 + k];
      } else {                                         // channel kl = 8 g + 4 hh + j
        const int k = which * 16 + (kl >> 3) * 4 + (kl & 3), hh = (kl >> 2) & 1;
#pragma unroll
        for (int gm = 0; gm < 4; gm++)
#pragma unroll 8
          for (int rr = 0; rr < 32; rr++) sum += red[gm * 2 * 64 * 33 + (hh * 32 + rr) * 33 + k];
      }
      stats_emit(p.stats, p.stats_acc, blockIdx.x, which, C, kg * 32 + kl, sum);
    }
    for (int row = gridDim.x + blockIdx.x; !p.stats_acc && row < p.stats_rows; row += gridDim.x)
      for (int c = t; c < 2 * C; c += NT) p.stats[(long)row * 2 * C + c] = 0.f;
  }
#endif
}

static int ws_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}

bool msml_conv_ws_applies(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S,
                          int stride, int pad_h, int pad_w, bool want_stats) {
  static const bool off = getenv("MSML_NO_WS_CONV") != nullptr;
  if (off) return false;
  if (R != 3 || S != 3 || stride != 1 || pad_h != 1 || pad_w != 1 || P != H || Q != W) return false;
  if (c0p != 64 || coutp != 64 || kop < 64) return false;
  const long tiles = (long)N * cdiv(H, 14) * cdiv(W, 14);
  if ((long)N * H * W * 10 < tiles * 224 * 7) return false;       // < 70 % real GEMM rows: im2col kernel wins
  const int wgs = tiles < ws_num_cus() ? (int)tiles : ws_num_cus();
  if (want_stats && wgs > cdiv((long)N * P * Q, msml_conv_tile_m(coutp))) return false;
  return (long)N * H * W * c0p * 2 < 0x70000000L && tiles < (1L << 30);
}

template <bool FUSE, bool XF = false, bool M16 = false>
static void launch_ws(ConvWsArgs& a, hipStream_t st) {
  // (+ the 2 pixels the padding rows read past an image, + the input-transform coefficient table)
  const size_t lds = 9 * 64 * 128 + 2 * 256 * 128 + 512 + (XF ? 3 * 64 * sizeof(float) : 0);
  static std::once_flag attr_once;                     // (per template instantiation; launches come from
  std::call_once(attr_once, [&] {                      //  the forward thread AND the autograd thread)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_ws<FUSE, XF, M16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int wgs = a.ntiles < ws_num_cus() ? a.ntiles : ws_num_cus();
  k_conv_ws<FUSE, XF, M16><<<dim3(wgs), dim3(512), lds, st>>>(a);
}

// Tried by msml_conv_fast_dispatch before the im2col kernel; false = shape not covered here.
bool msml_conv_ws_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out,
                           int coutp, float* stats, int N, int H, int W, int P, int Q, int R, int S,
                           int stride, int pad_h, int pad_w, int transposed, hipStream_t st,
                           const float* scale, const float* alpha, const void* residual, int res_first,
                           const BnBwdFuse* bnb, int* bnb_rows, const BnIn* xin) {
  if (!msml_conv_ws_applies(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, stats != nullptr))
    return false;
  if (bnb && (bias || scale || alpha || residual || stats)) return false;
  if (xin && (bnb || transposed)) return false;
#ifndef MSML_EXPERIMENTS
  if (xin) return false;       // (BatchNorm in this kernel's prologue measured slower, DESIGN section 8: experiment builds only)
#endif
  ConvWsArgs a;
  a.tpy = cdiv(H, 14); a.tpx = cdiv(W, 14);
  a.ntiles = N * a.tpy * a.tpx;
  a.in = (const unsigned short*)in0; a.in_bytes = (unsigned int)((long)N * H * W * c0p * 2);
  a.N = N; a.H = H; a.W = W; a.flip = transposed;
  a.wp = (const unsigned short*)wp;
  a.out = (unsigned short*)out;
  a.bias = bias; a.scale = scale; a.alpha = alpha; a.residual = (const unsigned short*)residual;
  a.res_first = res_first; a.stats = stats;
  a.stats_rows = cdiv((long)N * P * Q, msml_conv_tile_m(coutp));
  a.stats_acc = stats ? msml_tl_stats_acc : 0;
  a.bnb = BnBwdFuse{};
  if (bnb) a.bnb = *bnb;
  a.xin = BnIn{nullptr, nullptr, nullptr};
  if (xin) a.xin = *xin;
  if (bnb_rows) *bnb_rows = a.ntiles < ws_num_cus() ? a.ntiles : ws_num_cus();
  // 16x16x32 variant: measured neutral here (64 -> 64 @ 112x112 forward 426 -> 408 us, backward-data and the 56x56 maps
  // +-1 %: this kernel waits on its image loads and transposes, not on the MFMA clock) -- opt-in, MSML_WS_M16=1
  // Round 5: with the direct stores (no LDS transpose, one barrier per tile) the 16x16x32 variant is the faster one -- 64 -> 64 @
  // 112x112 forward 431 -> 385 us, @ 56x56 100 -> 93 us, the step 29.38 -> 29.17 / 29.26 ms (one box, twice): default;
  // MSML_WS_M16=0 restores the 32x32x16 kernels.
  static const bool m16 = !(getenv("MSML_WS_M16") && atoi(getenv("MSML_WS_M16")) == 0);
#ifdef MSML_EXPERIMENTS
  if (xin) { if (m16) launch_ws<false, true, true>(a, st); else launch_ws<false, true>(a, st); }   // (same tiling as the plain launch)
  else
#endif
  if (bnb) { if (m16) launch_ws<true, false, true>(a, st); else launch_ws<true>(a, st); }
  else { if (m16) launch_ws<false, false, true>(a, st); else launch_ws<false>(a, st); }
  return true;
}
