// Halo-tile 3x3 convolution, second family (round 5): the shapes conv_halo.hip leaves on the im2col kernel --
//   GEOM 1  stride-2 forward (conv2 of the first IBasicBlock of every stage, backbones/frb/iresnet.py:56-67 with
//           stride 2, :166-176; the OSB encoder's copies, backbones/osb/unet.py:80-91): the input is walked as its four
//           PARITY PLANES X(2i + py, 2j + px); over a plane the conv is a stride-1 conv with 1 / 2 / 2 / 4 taps at
//           offsets {-1, 0}, all four planes accumulate into the same output tile.  One LDS image per (64-channel
//           slab, plane), gathered with a pixel stride of two.
//   GEOM 2  stride-2 backward-data: one launch slice (blockIdx.z) per OUTPUT parity class (2i + cy, 2j + cx) of dX;
//           over dY a class is a stride-1 conv with 1 / 2 / 2 / 4 taps at offsets {0, +1}; strided store; the fused
//           BatchNorm backward sums (common.h) read the saved input at the strided pixels.
//   GEOM 0  stride-1 forward / backward-data (the contract of conv_halo.hip), kept here for the MOSAIC tiling only:
//   MOS 7   7x7 maps (the 512-channel stage, the deep OSB levels): a workgroup's tile is a 2 x 2 mosaic of FOUR images,
//           side by side on the 16-pixel LDS pitch with one zero pixel between them (1 + 7 + 1 + 7 columns = the
//           pitch; the zero column / row is at once the right halo of one image and the left halo of the next): 196
//           real pixels in 240 GEMM rows (the 14 x 14 tile: 49 of 224).
//   MOS 4   4x4 maps (the OSB's deepest level, backbones/osb/unet.py:205-209): 3 x 2 images per tile (1 + 4 + 1 + 4 + 1 +
//           4 + 1 columns, 9 GEMM row groups): 96 real pixels in 144 rows, 43 tiles x 4 channel blocks at batch 256
//           (the im2col kernel: 128 workgroups of 56-stage K loops at 250 TFLOP/s).
// Same machinery as k_conv_halo's 16x16x32 build: image + halo in LDS once per slab for all its taps (XOR chunk key
// p & 7), wave-private two-stage weight rings filled by LDS-DMA, D = W_frag x X_frag on v_mfma_f32_16x16x32_bf16,
// zero padding from out-of-range DMA offsets.  Statistics / fused BatchNorm sums in accumulator mode only.
#include <stdlib.h>

#include <mutex>

#include "common.h"

struct ConvHalo2Args {
  const unsigned short* in;
  unsigned int in_bytes;
  int C;              // input channels (multiple of 64)
  int N;
  int IH, IW;         // input map of the launch (GEOM 1: full resolution; GEOM 2: dY; GEOM 0: = GH x GW)
  int GH, GW;         // GEMM grid: the map the 14 x 14 tiles (or 7 x 7 mosaics) walk (GEOM 1: the output; GEOM 2: dY)
  int OH, OW;         // output map (GEOM 2: 2 GH x 2 GW)
  int tpy, tpx;       // tiles per image column / row (MOS: unused)
  int flip;           // GEOM 0 backward-data: tap (r, s) reads the image at (2 - r, 2 - s)
  const unsigned short* wp;
  unsigned int w_bytes;
  int Ktot;
  unsigned short* out;
  int coutp;
  const float* bias;                // (!FUSE) out = acc * scale + bias (+ residual); nullptr: 0 / 1
  const float* scale;
  const unsigned short* residual;   // (!FUSE) added to the result before the store (the FM `conv_tee` join)
  float* stats;       // accumulator-mode statistics double[MSML_ACC_ROWS][2][coutp] or nullptr
  BnBwdFuse bnb;
  // X3 (split-bf16 inference, x3.hip): C = 3 x the logical input channels ([hi | lo | hi] planes walked as plain channels),
  // `out` / `residual` hold 3 x coutp channels per pixel; out = [prelu](acc * scale + bias [+ residual]) [+ residual]
  const float* alpha;
  int res_first;
  int bias9;          // `bias` is float[9][coutp] by border class of the output pixel (common.h; stride-1 launches)
  int zrev;           // GEOM 2: launch slice z serves output class 3 - z -- the four-tap class (odd, odd) first, the one-tap
                      // class last, so that the light workgroups fill the launch's tail instead of the heavy ones making it
};

#define H2_OOB 0x78000000u

typedef __attribute__((address_space(3))) void* lptr_t;

template <int BN, int NWM, int GEOM, int MOS, bool FUSE, bool X3 = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv_halo2(const ConvHalo2Args p) {
  static_assert(!X3 || (!FUSE && GEOM != 2), "split-bf16: forward launches only");
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PL2 = 4, PITCH = 16, KG = BN / 32, NW = KG * NWM, NT = NW * 64;
  // mosaic geometry: images of IMG x IMG pixels every PER = IMG + 1 pixels, MC x MR of them per tile
  constexpr int IMG = MOS, PER = MOS + 1, MC = MOS == 4 ? 3 : 2, MR = 2, MPT = MC * MR;
  constexpr int NGRP = MOS ? MR * PER - 1 : 14, BM = NGRP * 16;  // 16-pixel GEMM row groups of a tile
  // LDS image: 16 rows of 16 pixels; MOS: + the zero row under the lower images + the 8 pixels behind it (the tap
  // (+1, +1) of the mosaic's last pixel, GEMM row 14 * 16 + 14, reads LDS pixel 272: zero-filled like the row)
  constexpr int HPX = MOS ? (NGRP + 2) * 16 + 8 : 16 * 16;
  constexpr int ABYTES = HPX * 128;
  constexpr int NAJ = HPX / 8, NAI = (NAJ + NW - 1) / NW;
  constexpr int GS = MOS == 4 ? 5 : 8;                 // NWM == 2: groups of the first pixel-row half
  constexpr int NGW = NWM == 1 ? NGRP : GS;            // groups of one wave (at most)
  constexpr int NGH = (NGW + 1) / 2;                   // ... per pipeline phase
  static_assert(NW == 8 && HPX % 8 == 0, "tile config");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                                     // [2][HPX][128 B]
  char* Bs = smem + 2 * ABYTES;                        // [NW][2][32][128 B]
  MSML_LDS_REGION(As, 2 * ABYTES);
  MSML_LDS_REGION(Bs, NW * 8192);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  auto skey = [](int p_) { return p_ & 7; };           // chunk key of LDS row p (conv_halo.hip, 16x16x32 lane map)
  const int kg = wave % KG, mg = wave / KG;
  const int g0 = mg * GS, ng = NWM == 1 ? NGRP : (mg == 0 ? GS : NGRP - GS);   // this wave's groups [g0, g0 + ng)
  const int tile = blockIdx.x;
  const int n0 = blockIdx.y * BN;
  const int zc = GEOM == 2 ? (p.zrev ? 3 - (int)blockIdx.z : (int)blockIdx.z) : 0;
  const int cy = zc >> 1, cx = zc & 1;
  int tn = 0, y0 = 0, x0 = 0;                          // plain tiling: image and tile origin on the GEMM grid
  if constexpr (!MOS) {
    const int tpi = p.tpy * p.tpx;
    tn = tile / tpi;
    const int trem = tile - tn * tpi, ty = trem / p.tpx;
    y0 = ty * 14; x0 = (trem - ty * p.tpx) * 14;
  }
  // GEMM row m = 16 my + mx  ->  is it a real output pixel, and where does it go
  auto pix_ok = [&](int m) {
    const int my = m >> 4, mx = m & 15;
    if constexpr (MOS) return (my % PER != IMG) & (mx % PER != IMG) & (mx < MC * PER - 1) &
                              (MPT * tile + MC * (my / PER) + mx / PER < p.N);
    else return (mx < 14) & (x0 + mx < p.GW) & (y0 + my < p.GH);
  };
  auto pix_off = [&](int m) {
    const int my = m >> 4, mx = m & 15;
    int n, gy, gx;
    if constexpr (MOS) { n = MPT * tile + MC * (my / PER) + mx / PER; gy = my % PER; gx = mx % PER; }
    else { n = tn; gy = y0 + my; gx = x0 + mx; }
    if constexpr (GEOM == 2) { gy = 2 * gy + cy; gx = 2 * gx + cx; }
    return ((long)(n * p.OH + gy) * p.OW + gx) * p.coutp;
  };

  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.w_bytes, 0x00020000);

  // LDS pixel hp = hy * PITCH + hx holds GEMM-grid pixel (y0 + hy - 1, x0 + hx - 1) of the plane / map being walked
  // (MOS: image (iy, ix) of the mosaic at hy = 1 + 8 iy + y, hx = 1 + 8 ix + x; everything else zero)
  unsigned int aoff[NAI];
#pragma unroll
  for (int i = 0; i < NAI; i++) {
    const int j = wave + i * NW;
    const int hp = j * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ skey(hp);
    const int hy = hp >> PL2, hx = hp & (PITCH - 1);
    int n, by, bx;
    bool v = j < NAJ;
    if constexpr (MOS) {
      const int ty_ = hy - 1, tx_ = hx - 1;
      v = v & (ty_ >= 0) & (tx_ >= 0) & (ty_ < NGRP) & (tx_ < MC * PER - 1);
      const int uy = ty_ < 0 ? 0 : ty_, ux = tx_ < 0 ? 0 : tx_;
      v = v & (uy % PER != IMG) & (ux % PER != IMG);
      n = MPT * tile + MC * (uy / PER) + ux / PER;
      by = uy % PER; bx = ux % PER;
      v = v & (n < p.N) & (by < p.GH) & (bx < p.GW);
    } else {
      n = tn; by = y0 + hy - 1; bx = x0 + hx - 1;
      v = v & ((unsigned)by < (unsigned)p.GH) & ((unsigned)bx < (unsigned)p.GW);
    }
    if constexpr (GEOM == 1) { by *= 2; bx *= 2; }     // plane (0, 0); plane (py, px) adds (py IW + px) pixels
    aoff[i] = v ? (unsigned int)((n * p.IH + by) * p.IW + bx) * (unsigned int)(p.C * 2) + logical * 16u : H2_OOB;
  }
  unsigned int boffg[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = i * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ skey(row);
    boffg[i] = (unsigned int)((n0 + kg * 32 + row) * p.Ktot) * 2u + logical * 16u;
  }

  // ---- the stage program: images (slab x plane) and, per image, a tap list rows x columns.
  // A list entry is (weight tap index r | LDS offset lr << 2); lr = 1 + the tap's pixel offset on the walked map.
  const int nslab = p.C >> 6;
  const int nimg = GEOM == 1 ? nslab * 4 : nslab;
  // GEOM 1, plane parity 1: taps r = 0 (offset -1) and r = 2 (offset 0); parity 0: r = 1 (offset 0)
  // GEOM 2, class parity 1: taps r = 0 (offset +1) and r = 2 (offset 0); parity 0: r = 1 (offset 0)
  auto axis_list = [&](int par, int& n_, int& pk_) {
    if constexpr (GEOM == 0) { n_ = 3; pk_ = p.flip ? 0x258 : 0xA50; }
    else if constexpr (GEOM == 1) { n_ = par ? 2 : 1; pk_ = par ? 0x60 : 0x5; }
    else { n_ = par ? 2 : 1; pk_ = par ? 0x68 : 0x5; }
  };
  auto img_kind = [&](int ii, int& nr_, int& rpk_, int& ns_, int& spk_, int& cs_, unsigned int& poff_) {
    if constexpr (GEOM == 1) {
      const int pid = 3 - (ii & 3), py = pid >> 1, px = pid & 1;     // the 4-tap plane first
      axis_list(py, nr_, rpk_);
      axis_list(px, ns_, spk_);
      cs_ = ii >> 2;
      poff_ = (unsigned int)(py * p.IW + px) * (unsigned int)(p.C * 2);
    } else {
      axis_list(cy, nr_, rpk_);
      axis_list(cx, ns_, spk_);
      cs_ = ii;
      poff_ = 0;
    }
  };
  auto issue_a = [&](int ii) {
    int nr_, rpk_, ns_, spk_, cs_;
    unsigned int poff_;
    img_kind(ii, nr_, rpk_, ns_, spk_, cs_, poff_);
    char* a = As + (ii & 1) * ABYTES;
#pragma unroll
    for (int i = 0; i < NAI; i++) {
      const int j = wave + i * NW;
      if (j < NAJ) {
        const unsigned int off = aoff[i] == H2_OOB ? H2_OOB : aoff[i] + poff_ + cs_ * 128u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(a + j * 1024), 16, off, 0, 0, 0);
      }
    }
  };
  auto issue_b = [&](int cs_, int wtap, int buf) {
    char* b = Bs + wave * 8192 + buf * 4096;
    const unsigned int col = (unsigned int)(wtap * p.C + cs_ * 64) * 2u;
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(b + i * 1024), 16, boffg[i] + col, 0, 0, 0);
  };

  f32x4 acc4[NGW][2];                                  // [pixel group][channel half]: channels 16 g + 4 q + j
#pragma unroll
  for (int i = 0; i < NGW; i++)
#pragma unroll
    for (int g = 0; g < 2; g++) acc4[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l16 = lane & 15, q16 = lane >> 4;
  int bfr16[2][2];                                     // weights row 16 g + l16 of the wave's ring, chunk 4 w + q16
#pragma unroll
  for (int g = 0; g < 2; g++)
#pragma unroll
    for (int w = 0; w < 2; w++) {
      const int row = 16 * g + l16;
      bfr16[g][w] = wave * 8192 + row * 128 + (((4 * w + q16) ^ skey(row)) << 4);
    }

  // current image / tap state (all wave-uniform)
  int ii = 0, jr = 0, js = 0;
  int nr, rpk, ns, spk, cs;
  unsigned int poff;
  img_kind(0, nr, rpk, ns, spk, cs, poff);
  issue_a(0);
  issue_b(cs, (rpk & 3) * 3 + (spk & 3), 0);
  __syncthreads();                                     // (drains vmcnt first)
  for (int q = 0;; q++) {
    // the stage after this one
    int nii = ii, njr = jr, njs = js + 1;
    if (njs == ns) { njs = 0; njr++; }
    if (njr == nr) { njr = 0; nii++; }
    int nnr = nr, nrpk = rpk, nns = ns, nspk = spk, ncs = cs;
    unsigned int npoff = poff;
    const bool last = nii == nimg;
    if (nii != ii && !last) img_kind(nii, nnr, nrpk, nns, nspk, ncs, npoff);
    // this wave's weights of stage q (issued one stage ago) have landed; queue stage q + 1 and, at the first tap of an
    // image, this wave's share of the next image (its buffer was read last in the image before this one)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!last) {
      issue_b(ncs, ((nrpk >> (4 * njr)) & 3) * 3 + ((nspk >> (4 * njs)) & 3), (q + 1) & 1);
      if ((jr | js) == 0 && ii + 1 < nimg) issue_a(ii + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      // pixel row (16 j + l16) of this tap is LDS pixel 16 j + l16 + lr PITCH + ls; the chunk key follows l16 + ls.
      // A stage = two 32-deep windows x two halves of the pixel groups: four phases of <= NGH image fragments +
      // 2 weight fragments feeding <= 2 NGH MFMAs, the next phase's fragments requested before the current MFMAs
      const int lr = (rpk >> (4 * jr + 2)) & 3, ls = (spk >> (4 * js + 2)) & 3;
      const int arow = l16 + ls, asw = skey(arow);
      const char* Arow = As + (ii & 1) * ABYTES + (((lr << PL2) + g0 * 16) * 128) + arow * 128;
      const char* B = Bs + (q & 1) * 4096;
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
          if (hf * NGH + j < ng && hf * NGH + j < NGW) {
#pragma unroll
            for (int g = 0; g < 2; g++)
              acc4[hf * NGH + j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, b16[w & 1][g]), __builtin_bit_cast(bf16x8, a16[cb][j]),
                  acc4[hf * NGH + j][g], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (last) break;
    if (nii != ii) __syncthreads();                    // image switch: the next image has landed everywhere
    ii = nii; jr = njr; js = njs;
    nr = nnr; rpk = nrpk; ns = nns; spk = nspk; cs = ncs; poff = npoff;
  }
  __syncthreads();

  // ---------------- epilogue ----------------------------------------------------------------------------------
  constexpr int OP = BN + 8;
  unsigned short* otile = reinterpret_cast<unsigned short*>(smem);
  MSML_LDS_REGION(otile, BM * OP * 2);
  constexpr int C8 = BN / 8, ITERS = (BM * C8 + NT - 1) / NT;
  static_assert(NT % C8 == 0, "a thread keeps one 8-channel chunk in the copy-out loop");
  const int c8 = t % C8;
  // FDIR (mosaic tiles only): the rounded dX leaves straight from registers (after v_permlane16_swap a lane holds channels
  // cdir .. cdir + 7 of its pixel) and feeds the BatchNorm sums in that layout, the saved BatchNorm input fetched per (pixel
  // group, lane): no LDS transpose, no barrier between the main loop and the stores.  Measured (tools/bench_bnbwd.py): 512 ->
  // 512 @ 7x7 69.2 -> 63.4 us; on the large maps the 64-B granularity of those loads costs more than the transpose saves (128 @
  // 56x56 stride 2: 193 -> 205 us, 256 @ 28x28: 94 -> 100 us), so the 14 x 14 tilings keep the LDS-transposed copy-out.
  constexpr bool FDIR = FUSE && MOS != 0;
  const int cdir = kg * 32 + ((lane >> 4) & 1) * 16 + (lane >> 5) * 8;
  u32x4 xr[FDIR ? NGW : ITERS];
  BnbCoef bk;
  float bq[3][8];
  if constexpr (FDIR) {
#pragma unroll
    for (int k = 0; k < NGW; k++) {
      const int m = (g0 + k) * 16 + (lane & 15);
      xr[k] = (k < ng && pix_ok(m)) ? *reinterpret_cast<const u32x4*>(p.bnb.x + pix_off(m) + n0 + cdir) : u32x4{0, 0, 0, 0};
    }
    bk = bnb_load_coef(p.bnb, n0 + cdir);
  } else if constexpr (FUSE) {
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
      const int idx = t + k * NT, m = idx / C8;
      xr[k] = (idx < BM * C8 && pix_ok(m)) ? *reinterpret_cast<const u32x4*>(p.bnb.x + pix_off(m) + n0 + c8 * 8)
                                            : u32x4{0, 0, 0, 0};
    }
    bk = bnb_load_coef(p.bnb, n0 + c8 * 8);
  }
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) bq[q][j] = 0.f;
  const int kb = kg * 32 + 4 * q16;                    // this lane's channels: kb + 16 g + j, g < 2
  f32x4 s1[2], s2[2], bv[2], sv[2];
#pragma unroll
  for (int g = 0; g < 2; g++) {
    s1[g] = s2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    bv[g] = (!FUSE && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + n0 + kb + 16 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    sv[g] = (!FUSE && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + n0 + kb + 16 * g) : f32x4{1.f, 1.f, 1.f, 1.f};
  }
  if constexpr (X3) {
    // split-bf16 output: the pair swap runs on the f32 accumulators (the lane then holds channels cdir .. cdir + 7 of its
    // pixel), affine + PReLU + residual (hi + lo planes) in f32, the result leaves as three planes (k_conv_fast's X3 epilogue)
    const bool act_here = p.alpha && !(p.residual && p.res_first), act_after = p.alpha && p.residual && p.res_first;
    const int cc = n0 + cdir;
    float sc[8], bi[8], al[8];
    {
      const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
      const f32x4 s0 = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + cc) : one;
      const f32x4 s1_ = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + cc + 4) : one;
      const f32x4 c0 = (p.bias && !p.bias9) ? *reinterpret_cast<const f32x4*>(p.bias + cc) : zero;
      const f32x4 c1 = (p.bias && !p.bias9) ? *reinterpret_cast<const f32x4*>(p.bias + cc + 4) : zero;
      const f32x4 a0 = p.alpha ? *reinterpret_cast<const f32x4*>(p.alpha + cc) : one;
      const f32x4 a1 = p.alpha ? *reinterpret_cast<const f32x4*>(p.alpha + cc + 4) : one;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        sc[c] = s0[c]; sc[4 + c] = s1_[c];
        bi[c] = c0[c]; bi[4 + c] = c1[c];
        al[c] = a0[c]; al[4 + c] = a1[c];
      }
    }
#pragma unroll
    for (int jg = 0; jg < NGW; jg++) {
      const int m = (g0 + jg) * 16 + l16;
      const bool valid = (jg < ng) & pix_ok(m);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float f0 = acc4[jg][0][e], f1 = acc4[jg][1][e];      // (float temporaries: see conv_s2r.hip)
        auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(f0), __float_as_uint(f1), false, false);
        v[e] = __uint_as_float(sw[0]);
        v[4 + e] = __uint_as_float(sw[1]);
      }
      const long po = pix_off(m) * 3 + cc;
      f32x4 b0 = {bi[0], bi[1], bi[2], bi[3]}, b1 = {bi[4], bi[5], bi[6], bi[7]};
      if (p.bias9 && valid) {
        const int my = m >> 4, mx = m & 15;
        const int gy = MOS ? my % PER : y0 + my, gx = MOS ? mx % PER : x0 + mx;
        const float* bp = p.bias + border_class(gy, gx, p.GH, p.GW) * p.coutp + cc;
        b0 = *reinterpret_cast<const f32x4*>(bp);
        b1 = *reinterpret_cast<const f32x4*>(bp + 4);
      }
      u32x4 rh = {0, 0, 0, 0}, rl = {0, 0, 0, 0};
      if (p.residual && valid) {
        rh = *reinterpret_cast<const u32x4*>(p.residual + po);
        rl = *reinterpret_cast<const u32x4*>(p.residual + po + p.coutp);
      }
      u32x4 oh, ol;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float z[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int c = 2 * i + k;
          float y = v[c] * sc[c] + (c < 4 ? b0[c & 3] : b1[c & 3]);
          if (act_here) y = y > 0.f ? y : y * al[c];
          if (p.residual) {
            y += __uint_as_float(k ? (rh[i] & 0xffff0000u) : (rh[i] << 16)) + __uint_as_float(k ? (rl[i] & 0xffff0000u) : (rl[i] << 16));
            if (act_after) y = y > 0.f ? y : y * al[c];
          }
          z[k] = y;
        }
        const unsigned short h0 = f2bf(z[0]), h1 = f2bf(z[1]);
        oh[i] = (unsigned int)h0 | ((unsigned int)h1 << 16);
        ol[i] = (unsigned int)f2bf(z[0] - bf2f(h0)) | ((unsigned int)f2bf(z[1] - bf2f(h1)) << 16);
      }
      if (valid) {
        *reinterpret_cast<u32x4*>(p.out + po) = oh;
        *reinterpret_cast<u32x4*>(p.out + po + p.coutp) = ol;
        *reinterpret_cast<u32x4*>(p.out + po + 2 * p.coutp) = oh;
      }
    }
    return;
  }
  const bool direct = FDIR || (!FUSE && p.residual == nullptr);
#pragma unroll
  for (int jg = 0; jg < NGW; jg++) {
    const int m = (g0 + jg) * 16 + l16;
    const bool valid = (jg < ng) & pix_ok(m);
    u32x2 pk[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float z = acc4[jg][g][j];
        if (!FUSE) z = z * sv[g][j] + bv[g][j];
        v[j] = z;
        if (!FUSE && valid) {
          s1[g][j] += z;
          s2[g][j] += z * z;
        }
      }
      pk[g][0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
      pk[g][1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
      if (!direct && jg < ng) *reinterpret_cast<u32x2*>(otile + m * OP + kb + 16 * g) = pk[g];
    }
    if (direct) {
      // v_permlane16_swap leaves every lane with 8 CONTIGUOUS channels (conv_halo.hip): rows 0 / 1 / 2 / 3 of the
      // wave -> channels 0-7 / 16-23 / 8-15 / 24-31 of its 32
      u32x4 o16;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        auto sw = __builtin_amdgcn_permlane16_swap(pk[0][e], pk[1][e], false, false);
        o16[e] = sw[0]; o16[2 + e] = sw[1];
      }
      if (valid) {
        *reinterpret_cast<u32x4*>(p.out + pix_off(m) + n0 + cdir) = o16;
        if constexpr (FDIR)
          bnb_accum(bk, p.bnb.alpha != nullptr, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&o16)),
                    load8<unsigned short>(reinterpret_cast<const unsigned short*>(&xr[jg])), bq);
      }
    }
  }
  if (!direct) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
      const int idx = t + k * NT, m = idx / C8;
      if (idx < BM * C8 && pix_ok(m)) {
        u32x4 v = *reinterpret_cast<const u32x4*>(otile + m * OP + c8 * 8);
        const long o = pix_off(m) + n0 + c8 * 8;
        if (!FUSE && p.residual) {
          Vec8 a8 = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v));
          const Vec8 r8 = load8<unsigned short>(p.residual + o);
#pragma unroll
          for (int j = 0; j < 8; j++) a8.v[j] += r8.v[j];
          store8<unsigned short>(reinterpret_cast<unsigned short*>(&v), a8);
        }
        *reinterpret_cast<u32x4*>(p.out + o) = v;
        if constexpr (FUSE && !FDIR)
          bnb_accum(bk, p.bnb.alpha != nullptr, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v)),
                    load8<unsigned short>(reinterpret_cast<const unsigned short*>(&xr[k])), bq);
      }
    }
  }
  const long wg = (long)blockIdx.x + (long)gridDim.x * blockIdx.z;
  if constexpr (FUSE) {
    constexpr int G = FDIR ? 16 * NWM : NT / C8;       // threads that share an 8-channel chunk
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    MSML_LDS_REGION(red, G * 3 * BN * 4);
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if constexpr (FDIR) red[((mg * 16 + (lane & 15)) * 3 + q) * BN + cdir + j] = bq[q][j];
        else red[((t / C8) * 3 + q) * BN + (t % C8) * 8 + j] = bq[q][j];
      }
    __syncthreads();
    for (int i = t; i < 3 * BN; i += NT) {
      const int q = i / BN, c = i % BN;
      float sum = 0.f;
      for (int g = 0; g < G; g++) sum += red[(g * 3 + q) * BN + c];
      bnb_emit(p.bnb.partial, 1, wg, q, p.coutp, n0 + c, sum);
    }
  }
  if (!FUSE && p.stats) {
    // per-channel (sum, sumsq) of this workgroup's pixels: lanes hold pixels, the 16 partials of every lane meet in
    // LDS and each lane of the pixel-row group 0 adds up one (statistic, channel) in a fixed order
    __syncthreads();
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
    if (mg == 0) {
      const int which = lane >> 5, kl = lane & 31;     // channel kl = 16 g + 4 q + j lives in the 16 lanes 16 q + rr
      const int k = which * 16 + (kl >> 4) * 4 + (kl & 3), qq = (kl >> 2) & 3;
      float sum = 0.f;
#pragma unroll
      for (int gm = 0; gm < NWM; gm++)
#pragma unroll 8
        for (int rr = 0; rr < 16; rr++) sum += red[gm * KG * 64 * 33 + (qq * 16 + rr) * 33 + k];
      stats_emit(p.stats, 1, wg, which, p.coutp, n0 + kg * 32 + kl, sum);
    }
  }
#endif
}

template <int BN, int NWM, int GEOM, int MOS, bool FUSE, bool X3 = false>
static void launch_halo2(ConvHalo2Args& a, hipStream_t st) {
  constexpr int NGRP = MOS ? 2 * (MOS + 1) - 1 : 14, HPX = MOS ? (NGRP + 2) * 16 + 8 : 16 * 16, BM = NGRP * 16;
  size_t lds = 2 * (size_t)HPX * 128 + 8 * 8192;        // two halo images + eight private weight rings
  const size_t olds = (size_t)BM * (BN + 8) * 2;        // transposed output tile
  const size_t slds = (size_t)8 * 64 * 33 * 4;          // statistics meet
  const size_t rlds = (size_t)(512 / (BN / 8)) * 3 * BN * 4;   // fused BatchNorm sums meet
  if (olds > lds) lds = olds;
  if (slds > lds) lds = slds;
  if (rlds > lds) lds = rlds;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_halo2<BN, NWM, GEOM, MOS, FUSE, X3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int tiles = MOS ? cdiv(a.N, MOS == 4 ? 6 : 4) : a.N * a.tpy * a.tpx;
  dim3 grid(tiles, a.coutp / BN, GEOM == 2 ? 4 : 1);
  k_conv_halo2<BN, NWM, GEOM, MOS, FUSE, X3><<<grid, dim3(512), lds, st>>>(a);
}

// geom: 0 stride 1, 1 stride-2 forward, 2 stride-2 backward-data.  Returns the tiling (0 none, 1 plain 14 x 14 tiles,
// 2 mosaic of four 7 x 7 images, 3 mosaic of six 4 x 4 images) this family takes the shape with.
int msml_conv_halo2_tiling(int c0p, int kop, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                           int pad_h, int pad_w, int transposed, int x3) {
  static const bool off = getenv("MSML_NO_HALO2_CONV") != nullptr;
  if (off) return 0;
  if (R != 3 || S != 3 || pad_h != 1 || pad_w != 1) return 0;
  if (c0p % 64 != 0 || c0p < 128 || coutp % 128 != 0 || kop < coutp) return 0;
  int gh, gw;                                           // the GEMM grid
  if (stride == 1) { if (P != H || Q != W) return 0; gh = H; gw = W; }
  else if (stride == 2 && !transposed) { if ((H & 1) || (W & 1) || P != H / 2 || Q != W / 2) return 0; gh = P; gw = Q; }
  else if (stride == 2) { if (P != 2 * H || Q != 2 * W) return 0; gh = H; gw = W; }      // (dY is the launch's input)
  else return 0;
  const long in_bytes = (long)N * H * W * c0p * 2, w_bytes = (long)kop * 9 * c0p * 2;
  if (in_bytes >= 0x70000000L || w_bytes >= 0x70000000L || (long)N * P * Q * coutp * 2 >= 0x7fffffffL * 2) return 0;
  static const bool no_mos = getenv("MSML_NO_HALO2_MOSAIC") != nullptr;
  static const bool no_s2 = getenv("MSML_NO_HALO2_S2") != nullptr;
  // mosaics run 128-channel tiles: below ~160 workgroups (256 -> 256 @ 7x7: 128, 128 -> 128 @ 7x7: 64, the forward of
  // 256 @ 14 -> 7: 128) the im2col kernel's 64-row tiles fill the chip better (tools/bench_small.py: 37.4 -> 34.9 us,
  // 17.6 -> 19.7 us, 36.1 -> 37.7 us); a backward-data launch has four slices per tile
  // (split-bf16 inference: the kernel choice must not depend on N -- an image's embedding is bit-identical whatever batch
  // it is extracted in, tests/test_verification.py -- so the mosaics take every batch there; 512-channel layers only)
  const long cblk = coutp / 128, slices = (stride == 2 && transposed) ? 4 : 1;
  const bool fill7 = x3 ? coutp >= 512 : cdiv(N, 4) * cblk * slices >= 160;
  const bool fill4 = x3 ? coutp >= 512 : cdiv(N, 6) * cblk >= 160;
  if (gh == 7 && gw == 7) return (no_mos || (stride == 2 && no_s2) || !fill7) ? 0 : 2;
  if (gh == 4 && gw == 4 && stride == 1) return (no_mos || !fill4) ? 0 : 3;
  if (stride == 1 || no_s2) return 0;                   // stride-1 maps with real tiles: conv_halo.hip
  const long tiles = (long)N * cdiv(gh, 14) * cdiv(gw, 14);
  if ((long)N * gh * gw * 10 < tiles * 224 * 7) return 0;           // < 70 % real GEMM rows: im2col kernel wins
  return 1;
}

// Tried by msml_conv_fast_dispatch before the im2col kernel; false = shape / epilogue not covered here.
bool msml_conv_halo2_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out, int coutp,
                              float* stats, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                              int pad_w, int transposed, hipStream_t st, const float* scale, const float* alpha,
                              const void* residual, int res_first, const BnBwdFuse* bnb, int* bnb_rows, int x3) {
  const int tiling = msml_conv_halo2_tiling(c0p, kop, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed, x3);
  if (!tiling) return false;
  // split-bf16 inference (c0p = 3 x the logical channels): forward launches, no statistics; MSML_NO_HALO2_X3=1 (read per
  // call) leaves them on the im2col kernel
  if (x3 && (transposed || bnb || stats || getenv("MSML_NO_HALO2_X3") != nullptr)) return false;
  if (x3 && (long)N * P * Q * coutp * 6 >= 0x7fffffffL * 2) return false;
  if (msml_tl_bias9 && !(x3 && stride == 1)) return false;
  if (!x3 && (alpha || res_first)) return false;
  if (bnb && (bias || scale)) return false;
  if (stats && !msml_tl_stats_acc) return false;
  if (bnb && (!bnb->acc || residual || stats)) return false;
  if (residual && stats) return false;
  const int geom = stride == 1 ? 0 : (transposed ? 2 : 1);
  ConvHalo2Args a;
  a.in = (const unsigned short*)in0; a.in_bytes = (unsigned int)((long)N * H * W * c0p * 2); a.C = c0p;
  a.N = N; a.IH = H; a.IW = W;
  a.GH = geom == 1 ? P : H; a.GW = geom == 1 ? Q : W;
  a.OH = P; a.OW = Q;
  a.tpy = cdiv(a.GH, 14); a.tpx = cdiv(a.GW, 14);
  a.flip = transposed;
  a.wp = (const unsigned short*)wp; a.w_bytes = (unsigned int)((long)kop * 9 * c0p * 2); a.Ktot = 9 * c0p;
  a.out = (unsigned short*)out; a.coutp = coutp;
  a.bias = bias; a.scale = scale;
  a.residual = (const unsigned short*)residual;
  a.stats = stats;
  a.bnb = BnBwdFuse{};
  if (bnb) a.bnb = *bnb;
  a.alpha = alpha; a.res_first = res_first; a.bias9 = (x3 && bias) ? msml_tl_bias9 : 0;
  a.zrev = 1;        // (tools/bench_small.py, three interleaved pairs: 256 @ 28 -> 56 79.4 / 79.0 / 78.2 -> 76.7 / 77.7 / 77.9 us,
                     //  128 @ 28 32.0 -> 31.3, the other shapes inside the noise; outputs bit-identical)
  const bool mos = tiling >= 2;
  const bool wide = coutp % 256 == 0 && !mos;           // mosaic: N / 4 tiles -- 128-channel tiles fill the chip sooner
  if (bnb_rows) *bnb_rows = (mos ? cdiv(N, tiling == 3 ? 6 : 4) : N * a.tpy * a.tpx) * (geom == 2 ? 4 : 1);
#define H2_CASE(GEOM, MOS)                                                              \
  if (bnb) { if (wide) launch_halo2<256, 1, GEOM, MOS, true>(a, st); else launch_halo2<128, 2, GEOM, MOS, true>(a, st); } \
  else { if (wide) launch_halo2<256, 1, GEOM, MOS, false>(a, st); else launch_halo2<128, 2, GEOM, MOS, false>(a, st); }
  if (x3) {
    if (tiling == 3) launch_halo2<128, 2, 0, 4, false, true>(a, st);
    else if (mos && geom == 0) launch_halo2<128, 2, 0, 7, false, true>(a, st);
    else if (mos) launch_halo2<128, 2, 1, 7, false, true>(a, st);
    else if (wide) launch_halo2<256, 1, 1, 0, false, true>(a, st);
    else launch_halo2<128, 2, 1, 0, false, true>(a, st);
    return true;
  }
  if (tiling == 3) {
    if (bnb) launch_halo2<128, 2, 0, 4, true>(a, st); else launch_halo2<128, 2, 0, 4, false>(a, st);
  } else if (mos) {
    if (geom == 0) { if (bnb) launch_halo2<128, 2, 0, 7, true>(a, st); else launch_halo2<128, 2, 0, 7, false>(a, st); }
    else if (geom == 1) { if (bnb) return false; launch_halo2<128, 2, 1, 7, false>(a, st); }
    else { if (bnb) launch_halo2<128, 2, 2, 7, true>(a, st); else launch_halo2<128, 2, 2, 7, false>(a, st); }
  } else if (geom == 1) {
    if (bnb) return false;
    if (wide) launch_halo2<256, 1, 1, 0, false>(a, st); else launch_halo2<128, 2, 1, 0, false>(a, st);
  } else {
    H2_CASE(2, 0)
  }
#undef H2_CASE
  return true;
}
