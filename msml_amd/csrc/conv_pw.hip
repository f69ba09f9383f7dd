// Pointwise (1x1, stride 1) convolutions for bf16 NHWC tensors -- the FM bottlenecks' conv1 / conv3
// (backbones/fm/fmoperator.py:53-68 of the reference: C -> C/2 -> C/2 -> C at 56x56 ... 14x14), the im2col'd stems
// (stem.hip: a 1x1 conv over 32-channel patches) and their backward-data convs.  Same contract as msml_conv2d /
// msml_conv2d_acc / msml_conv2d_fused / msml_conv2d_bnbwd for the cases it takes (conv_fast.hip tries it first).
//
// These launches are HBM-bound (32 -> 64 channels @ 112x112: 617 MB for 26 GFLOP) and K is one or two im2col stages
// deep, so the general kernel's workgroup = (load tile, wait, 8 MFMAs, transpose through LDS, store) has nothing to
// overlap inside and sits at 1.9-4.1 TB/s against a 5.1-5.8 TB/s device copy (tools/bench_pw.py).  Here the input is
// a plain [M][CIN] matrix and every WAVE is an independent persistent worker over 32-pixel blocks:
//   * global memory is only touched in whole rows: lane l <-> 16-B chunk l of the block's contiguous bytes, both for
//     the input (the next unit's chunks are requested into registers before the current unit is computed: 4-8 KB per
//     wave in flight) and for the output, the residual and the saved BatchNorm input of the fused epilogues;
//   * the MFMA operand layout (lane = (pixel, k half): adjacent lanes hold DIFFERENT pixels) is produced by a trip
//     through a wave-private LDS scratch: registers -> swizzled pixel rows -> ds_read_b128 fragments, and back for the
//     output tile (v_permlane32_swap, 16-B writes in fragment layout, linear 16-B reads).  One wave, LDS operations in
//     program order: no barrier, no hand-placed s_waitcnt -- the compiler tracks every wait.  (The first version moved
//     fragments between global memory and registers directly: every 16-B access was a memory request of its own,
//     four times the request count, and only the 112x112 stem launch came out ahead of the general kernel);
//   * the packed weight sits in LDS for the life of the workgroup; D = W_frag x X_frag (channels in the accumulator
//     rows); the residual add and the BatchNorm backward sums run in the copy-out layout, where a lane keeps ONE
//     8-channel chunk for the whole kernel (its coefficients live in registers);
//   * statistics / BatchNorm backward sums stay in registers until the worker has no unit left, are folded over the
//     lanes with DPP adds (a __shfl_xor tree = five ds_bpermute per value was a fixed ~15 us at the end of every
//     statistics launch) and over the workgroup's waves in LDS: one f64 atomic per channel and workgroup.
// The arithmetic (MFMA shape and k order, rounding points of the epilogues) is k_conv_fast's: outputs are bit-identical.
#include <stdlib.h>

#include <mutex>

#include "common.h"

struct ConvPwArgs {
  const unsigned short* in;        // [M][CIN]
  const unsigned short* wp;        // packed [>= COUT rows][ktot]
  unsigned short* out;             // [M][COUT]
  const float* bias;               // per-channel shift or null
  const float* scale;              // per-channel scale or null
  const unsigned short* residual;  // PW_ADD: [M][COUT], added to the ROUNDED conv output (msml_conv2d_fused's order)
  double* stats;                   // PW_STATS: accumulator double[MSML_ACC_ROWS][2][COUT]
  BnBwdFuse bnb;                   // PW_BNB (accumulator mode): sums into double[MSML_ACC_ROWS][3][COUT]
  long M;
  int nblk;                        // 32-pixel blocks
  int ktot;                        // row pitch of wp (elements)
};

enum { PW_PLAIN = 0, PW_STATS = 1, PW_ADD = 2, PW_BNB = 3 };

// LDS map of a workgroup (4 waves): packed weights | epilogue table | end-of-kernel sums | per-wave scratch
template <int CIN, int COUT>
struct PwLds {
  static constexpr int CINB = CIN * 2, NCO = COUT / 32, NCOW = NCO == 1 ? 1 : 2, COUTW = 32 * NCOW;
  static constexpr int XINB = 32 * CINB, XOUTB = 32 * COUTW * 2, SCR = XINB + XOUTB;
  static constexpr int W = 0, TAB = COUT * CINB, RED = TAB + 2 * COUT * 4, SCRATCH = RED + 4 * 3 * 64 * 4;
  static constexpr int TOTAL = SCRATCH + 4 * SCR;
};

template <int CIN, int COUT, int MODE, bool AFF>
__global__ void __launch_bounds__(256) k_conv_pw(const ConvPwArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using L = PwLds<CIN, COUT>;
  constexpr int CINB = CIN * 2, CH = CIN / 8, KK = CIN / 16, NCO = COUT / 32;
  // a work unit = NB consecutive 32-pixel blocks (8 KB of input per wave in flight) x NCOW 32-channel tiles; the NCG
  // units of a block group go to adjacent waves, which share its input through L1 / L2
  constexpr int NB = CIN >= 128 ? 1 : (MODE == PW_STATS ? 64 : 128) / CIN;   // (STATS: 64 accumulator registers)
  constexpr int NCOW = L::NCOW, NCG = NCO / NCOW, COUTW = L::COUTW;
  constexpr int C8W = COUTW / 8, OPB = COUTW * 2, NOJ = 32 * C8W / 64;    // output chunks per pixel / lane
  static_assert(NCG == 1 || NCG == 2 || NCG == 4, "channel groups must divide the worker count");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem + L::W;                               // [COUT][CINB], 16-B chunks swizzled by the row
  float* ktab = reinterpret_cast<float*>(smem + L::TAB);         // AFF: [2][COUT] scale, shift of the epilogue
  float* red = reinterpret_cast<float*>(smem + L::RED);          // [4 waves][NQ][COUTW]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  char* xin = smem + L::SCRATCH + wave * L::SCR;        // this wave's input block [32 px][CINB], chunks swizzled
  char* xout = xin + L::XINB;                           // ... and its output tile [32 px][COUTW], chunks swizzled
  MSML_LDS_REGION(Ws, COUT * CINB);
  MSML_LDS_REGION(ktab, 2 * COUT * 4);
  MSML_LDS_REGION(red, 4 * 3 * 64 * 4);
  MSML_LDS_REGION(xin, L::XINB);
  MSML_LDS_REGION(xout, L::XOUTB);
  // the 16 rows one ds_read_b128 lane group touches must land on 16 different 16-B bank slots (conv_line.hip)
  auto key = [](int row) -> int { return CINB == 64 ? ((row >> 2) & 3) : (CINB == 128 ? ((row >> 1) & 7) : (row & 15)); };
  // output tile: the key of pixel p must not depend on which of a lane's NOJ linear chunks p belongs to, so that a lane
  // of the copy-out loop keeps ONE channel chunk (p = j * 64 / C8W + lane / C8W): 64-B rows -> (p >> 2) & 3 (conflict-free
  // for the fragment-layout writes), 128-B rows -> p & 7 (2-way on those writes, which are 2 % of the LDS traffic)
  auto okey = [](int row) -> int { return C8W == 4 ? ((row >> 2) & 3) : (row & 7); };

  for (int i = t; i < COUT * CH; i += 256) {
    const int row = i / CH, c = i - row * CH;
    const u32x4 v = *reinterpret_cast<const u32x4*>(p.wp + (long)row * p.ktot + c * 8);
    *reinterpret_cast<u32x4*>(Ws + row * CINB + ((c ^ key(row)) << 4)) = v;
  }
  if (AFF) {
    for (int c = t; c < COUT; c += 256) {
      ktab[c] = p.scale ? p.scale[c] : 1.f;
      ktab[COUT + c] = p.bias ? p.bias[c] : 0.f;
    }
  }
  __syncthreads();

  const int gw = blockIdx.x * 4 + wave, tw = gridDim.x * 4;      // tw % NCG == 0: a worker keeps its channel group
  const int ngrp = (p.nblk + NB - 1) / NB, nunits = ngrp * NCG;
  const int cg = gw % NCG, cbase = cg * COUTW;
  const bool has_alpha = MODE == PW_BNB && p.bnb.alpha != nullptr;
  // copy-out loop: lane <-> linear 16-B chunk (j * 64 + lane) of the tile = pixel j * (64 / C8W) + lane / C8W, stored
  // chunk lane % C8W = logical chunk cl (the same for every j)
  const int opl = lane / C8W;
  const int cl = (lane % C8W) ^ okey(opl);
  BnbCoef bk;
  if (MODE == PW_BNB) bk = bnb_load_coef(p.bnb, cbase + cl * 8);

  f32x4 s1[MODE == PW_STATS ? 4 * NCOW : 1], s2[MODE == PW_STATS ? 4 * NCOW : 1];      // channels cbase + 32 i + 8 g + 4 h + j
#pragma unroll
  for (int g = 0; g < (MODE == PW_STATS ? 4 * NCOW : 1); g++) s1[g] = s2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bq[3][8];                                       // PW_BNB: sums of channels cbase + 8 cl + j
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) bq[q][j] = 0.f;

  // coalesced: lane <-> 16-B chunk (j * 64 + lane) of the block's contiguous 32 x CINB bytes
  auto load_x = [&](int grp, u32x4 (&dst)[NB][KK]) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const long pb = ((long)grp * NB + b) * 32;
      const char* src = reinterpret_cast<const char*>(p.in) + pb * CINB + lane * 16;
#pragma unroll
      for (int j = 0; j < KK; j++)
        dst[b][j] = (pb + (j * 64 + lane) / CH < p.M) ? *reinterpret_cast<const u32x4*>(src + j * 1024) : u32x4{0, 0, 0, 0};
    }
  };

  u32x4 xc[NB][KK], xn[NB][KK];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int kk = 0; kk < KK; kk++) xc[b][kk] = xn[b][kk] = u32x4{0, 0, 0, 0};
  int u = gw;
  if (u < nunits) load_x(u / NCG, xc);
  for (; u < nunits; u += tw) {
    const int nu = u + tw;
    if (nu < nunits) load_x(nu / NCG, xn);
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const long pb = ((long)(u / NCG) * NB + b) * 32;
      const bool valid = pb + r32 < p.M;
      // second operands of the epilogue (residual / saved BatchNorm input), in the copy-out layout: they fly during the MFMAs
      u32x4 e2[(MODE == PW_ADD || MODE == PW_BNB) ? NOJ : 1];
      if (MODE == PW_ADD || MODE == PW_BNB) {
        const unsigned short* src = MODE == PW_ADD ? p.residual : p.bnb.x;
#pragma unroll
        for (int j = 0; j < NOJ; j++) {
          const long m = pb + j * (64 / C8W) + opl;
          e2[j] = m < p.M ? *reinterpret_cast<const u32x4*>(src + m * COUT + cbase + cl * 8) : u32x4{0, 0, 0, 0};
        }
      }
      // input block: registers (coalesced) -> LDS (pixel rows) -> MFMA fragments; one wave, LDS operations in order
#pragma unroll
      for (int j = 0; j < KK; j++) {
        const int idx = j * 64 + lane, pl = idx / CH, c = idx % CH;
        *reinterpret_cast<u32x4*>(xin + pl * CINB + ((c ^ key(pl)) << 4)) = xc[b][j];
      }
      __builtin_amdgcn_wave_barrier();
      f32x16 acc[NCOW];
#pragma unroll
      for (int i = 0; i < NCOW; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; kk++) {
        const u32x4 xb = *reinterpret_cast<const u32x4*>(xin + r32 * CINB + (((kk * 2 + h) ^ key(r32)) << 4));
#pragma unroll
        for (int i = 0; i < NCOW; i++) {
          const int row = cbase + 32 * i + r32;
          const u32x4 a = *reinterpret_cast<const u32x4*>(Ws + row * CINB + (((kk * 2 + h) ^ key(row)) << 4));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, xb),
                                                           acc[i], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NCOW; i++) {
        u32x2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
          float v[4];
          f32x4 sv = {1.f, 1.f, 1.f, 1.f}, bv = {0.f, 0.f, 0.f, 0.f};
          if (AFF) {
            sv = *reinterpret_cast<const f32x4*>(ktab + cbase + 32 * i + 8 * g + 4 * h);
            bv = *reinterpret_cast<const f32x4*>(ktab + COUT + cbase + 32 * i + 8 * g + 4 * h);
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            v[j] = AFF ? acc[i][g * 4 + j] * sv[j] + bv[j] : acc[i][g * 4 + j];
            if (MODE == PW_STATS && valid) {
              s1[4 * i + g][j] += v[j];
              s2[4 * i + g][j] += v[j] * v[j];
            }
          }
          pk[g][0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
          pk[g][1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
        }
        // lane (pixel r32, half h): channels 8 h .. + 7 and 16 + 8 h .. + 7 of tile i after the swaps (conv_line.hip)
        u32x4 ch[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          auto r01 = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
          auto r23 = __builtin_amdgcn_permlane32_swap(pk[2][e], pk[3][e], false, false);
          ch[0][e] = r01[0]; ch[0][2 + e] = r01[1];
          ch[1][e] = r23[0]; ch[1][2 + e] = r23[1];
        }
#pragma unroll
        for (int c = 0; c < 2; c++)
          *reinterpret_cast<u32x4*>(xout + r32 * OPB + (((4 * i + 2 * c + h) ^ okey(r32)) << 4)) = ch[c];
      }
      __builtin_amdgcn_wave_barrier();
      // copy-out: 64 consecutive 16-B chunks per instruction (whole pixel rows), residual / BatchNorm sums in this layout
#pragma unroll
      for (int j = 0; j < NOJ; j++) {
        const long m = pb + j * (64 / C8W) + opl;
        u32x4 v = *reinterpret_cast<const u32x4*>(xout + (j * 64 + lane) * 16);
        if (m < p.M) {
          if (MODE == PW_ADD) {
            Vec8 a = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v));
            const Vec8 r = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&e2[j]));
#pragma unroll
            for (int q = 0; q < 8; q++) a.v[q] += r.v[q];
            store8<unsigned short>(reinterpret_cast<unsigned short*>(&v), a);
          }
          if (MODE == PW_BNB)
            bnb_accum(bk, has_alpha, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v)),
                      load8<unsigned short>(reinterpret_cast<const unsigned short*>(&e2[j])), bq);
          *reinterpret_cast<u32x4*>(p.out + m * COUT + cbase + cl * 8) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);                // one block at a time: interleaving the NB blocks costs > 256 VGPRs
    }
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int kk = 0; kk < KK; kk++) xc[b][kk] = xn[b][kk];
  }

  // ---- the worker's sums: over its lanes (fixed shuffle tree), then over the waves of the workgroup that share a
  // channel group (LDS, fixed order), then one f64 atomic per channel and workgroup (all workers finish together: four
  // times fewer atomics queue on each accumulator address)
  if (MODE == PW_STATS || MODE == PW_BNB) {
    constexpr int NQ = MODE == PW_STATS ? 2 : 3;
    if (MODE == PW_STATS) {
#pragma unroll
      for (int g = 0; g < 4 * NCOW; g++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float a = wave_half_sum(s1[g][j]), b = wave_half_sum(s2[g][j]);
          if (r32 == 0) {
            red[(wave * NQ + 0) * COUTW + 8 * g + 4 * h + j] = a;      // (g = 4 i + g': channel 32 i + 8 g' + 4 h + j)
            red[(wave * NQ + 1) * COUTW + 8 * g + 4 * h + j] = b;
          }
        }
    } else {
      // lanes that share the chunk cl = (lane % C8W) ^ okey(lane / C8W): flip the same bit on both sides of the XOR
      // (8 chunks: lane bits k and k + 3; 4 chunks: bits k and k + 4), and the remaining pixel bits (4 chunks: 2, 3)
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          float a = bq[q][j];
          if (C8W == 8) {
            a += __shfl_xor(a, 9, 64);
            a += __shfl_xor(a, 18, 64);
            a += __shfl_xor(a, 36, 64);
          } else {
            a += __shfl_xor(a, 17, 64);
            a += __shfl_xor(a, 34, 64);
            a += __shfl_xor(a, 4, 64);
            a += __shfl_xor(a, 8, 64);
          }
          if (lane < C8W) red[(wave * NQ + q) * COUTW + 8 * cl + j] = a;      // (okey(0) = 0: cl = lane)
        }
    }
    __syncthreads();
    for (int e = t; e < NCG * NQ * COUTW; e += 256) {
      const int g = e / (NQ * COUTW), q = (e / COUTW) % NQ, c = e % COUTW;     // wave w of this workgroup serves channel group w % NCG
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; w++)
        if (w % NCG == g) sum += red[(w * NQ + q) * COUTW + c];
      if (MODE == PW_STATS) stats_emit(reinterpret_cast<float*>(p.stats), 1, blockIdx.x, q, COUT, g * COUTW + c, sum);
      else bnb_emit(p.bnb.partial, 1, blockIdx.x, q, COUT, g * COUTW + c, sum);
    }
  }
#endif
}

template <int CIN, int COUT, int MODE, bool AFF>
static void pw_launch(const ConvPwArgs& a, hipStream_t st) {
  using L = PwLds<CIN, COUT>;
  constexpr int NCG = (COUT / 32) / L::NCOW;
  constexpr int NB = CIN >= 128 ? 1 : (MODE == PW_STATS ? 64 : 128) / CIN;   // (STATS: 64 accumulator registers)
  const size_t lds = L::TOTAL;
  if (lds > 64 * 1024) {
    static std::once_flag once;
    std::call_once(once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_pw<CIN, COUT, MODE, AFF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
  }
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n;
  }
  // persistent workers: up to 4 workgroups of 4 waves per CU (the register / LDS footprint of the instantiation may
  // allow fewer -- the rest queue behind them); never more workers than units
  static const int per_cu = getenv("MSML_PW_WGS_PER_CU") ? atoi(getenv("MSML_PW_WGS_PER_CU")) : 4;
  long grid = (long)cus * per_cu;
  const long ngrp = (a.nblk + NB - 1) / NB;
  const long need = (ngrp * NCG + 3) / 4;
  if (grid > need) grid = need;
  // (knob: at least `upw` units per worker for STATS / BNB launches, whose sums leave every workgroup as f64 atomics at
  // the end of the kernel; measured neutral once the in-wave reduction ran on DPP -- default 1 = as many workers as fit)
  static const int upw = getenv("MSML_PW_UPW") ? atoi(getenv("MSML_PW_UPW")) : 1;
  if (MODE == PW_STATS || MODE == PW_BNB) {
    long g2 = need / upw;
    if (g2 < cus) g2 = cus;
    if (grid > g2) grid = g2;
  }
  if (grid < 1) grid = 1;
  k_conv_pw<CIN, COUT, MODE, AFF><<<dim3((unsigned)grid), dim3(256), lds, st>>>(a);
}

template <int MODE, bool AFF>
static bool pw_shape(const ConvPwArgs& a, int cin, int cout, hipStream_t st) {
#define PW_CASE(CI, CO) if (cin == CI && cout == CO) { pw_launch<CI, CO, MODE, AFF>(a, st); return true; }
  // (128 -> 256 / 256 -> 128 @ 14x14 measured SLOWER here, 38 against 16-21 us: 64 KB of weights per workgroup for 3-6 units
  // per worker on 38 MB tensors -- they stay on the general kernel)
  PW_CASE(32, 64) PW_CASE(64, 32) PW_CASE(64, 64) PW_CASE(64, 128) PW_CASE(128, 64)
#undef PW_CASE
  return false;
}

// Tried by msml_conv_fast_dispatch before everything else; false = not a case this kernel takes.
bool msml_conv_pw_dispatch(const void* in0, int c0p, const void* wp, int kop, int ktot, const float* bias, void* out,
                           int coutp, float* stats, int stats_acc, int N, int H, int W, int P, int Q, int R, int S,
                           int stride, int pad_h, int pad_w, hipStream_t st, const float* scale, const float* alpha,
                           const void* residual, int res_first, const BnBwdFuse* bnb) {
  // Measured against the general kernel on cold data (tools/bench_pw.py, batch 256, one box, us): forward + statistics
  // 32 -> 64 @ 112x112 177 -> 125 (4.9 TB/s; a device copy of the same bytes: 5.2-5.8), @ 56x56 51 -> 41, 64 -> 32 @ 56x56
  // 38 -> 29, the 28x28 layers 28 -> 25-29; backward-data 64 -> 32 @ 56x56 41 -> 29, 32 -> 64 33 -> 27 (5.6 TB/s),
  // 128 -> 64 @ 28x28 24 -> 17.  (The first version loaded and stored in MFMA operand layout -- adjacent lanes = different
  // pixels, every 16-B access a memory request of its own -- and only won on the 112x112 stem; see DESIGN section 8 g.)
  // MSML_PW_CONV=0: the general kernel everywhere (read per call: the tests compare both kernels in one process).
  const char* pol = getenv("MSML_PW_CONV");
  if (pol && pol[0] == '0') return false;
  if (R != 1 || S != 1 || stride != 1 || pad_h != 0 || pad_w != 0 || P != H || Q != W) return false;
  if (alpha || (residual && res_first)) return false;
  if (stats && !stats_acc) return false;                // partial-row statistics stay on the general kernel
  if (bnb && (!bnb->acc || bias || scale || residual || stats)) return false;
  if (stats && residual) return false;
  if (kop < coutp || ktot < c0p) return false;
  const long M = (long)N * H * W;
  if (M * (c0p > coutp ? c0p : coutp) * 2 >= 0x7fffff00L) return false;
  ConvPwArgs a;
  a.in = (const unsigned short*)in0; a.wp = (const unsigned short*)wp; a.out = (unsigned short*)out;
  a.bias = bias; a.scale = scale; a.residual = (const unsigned short*)residual;
  a.stats = reinterpret_cast<double*>(stats);
  a.bnb = BnBwdFuse{};
  if (bnb) a.bnb = *bnb;
  a.M = M; a.nblk = (int)((M + 31) / 32); a.ktot = ktot;
  const bool aff = bias || scale;
  if (bnb) return pw_shape<PW_BNB, false>(a, c0p, coutp, st);
  if (residual) return aff ? pw_shape<PW_ADD, true>(a, c0p, coutp, st) : pw_shape<PW_ADD, false>(a, c0p, coutp, st);
  if (stats) return aff ? pw_shape<PW_STATS, true>(a, c0p, coutp, st) : pw_shape<PW_STATS, false>(a, c0p, coutp, st);
  return aff ? pw_shape<PW_PLAIN, true>(a, c0p, coutp, st) : pw_shape<PW_PLAIN, false>(a, c0p, coutp, st);
}
