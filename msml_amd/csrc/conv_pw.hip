// Pointwise (1x1, stride 1) convolutions for bf16 NHWC tensors -- the FM bottlenecks' conv1 / conv3
// (backbones/fm/fmoperator.py:53-68 of the reference: C -> C/2 -> C/2 -> C at 56x56 ... 14x14), the im2col'd stems
// (stem.hip: a 1x1 conv over 32-channel patches) and their backward-data convs.  Same contract as msml_conv2d /
// msml_conv2d_acc / msml_conv2d_fused / msml_conv2d_bnbwd for the cases it takes (conv_fast.hip tries it first).
//
// These launches are HBM-bound (32 -> 64 channels @ 112x112: 617 MB for 26 GFLOP) and K is one or two im2col stages
// deep, so the general kernel's workgroup = (load tile, wait, 8 MFMAs, transpose through LDS, store) has nothing to
// overlap inside and sits at 1.9-4.1 TB/s against a 5.1-5.8 TB/s device copy (tools/bench_pw.py).  Here the input is
// a plain [M][CIN] matrix: every wave is an independent persistent worker over 32-pixel blocks -- X fragments go
// straight from global memory to VGPRs in MFMA layout (lane (pixel r32, k-half h) <- 16 B of row r32), the next
// block's loads are issued before the current block's MFMAs, the packed weight sits in LDS for the life of the
// workgroup, D = W_frag x X_frag (channels in the accumulator rows) lets a lane pair store 32 contiguous channels of
// a pixel after v_permlane32_swap.  No barrier after the weights have landed; statistics / BatchNorm backward sums
// stay in registers until the worker has no block left (one f64 atomic per channel and wave).
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

template <int CIN, int COUT, int MODE, bool AFF>
__global__ void __launch_bounds__(256) k_conv_pw(const ConvPwArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CINB = CIN * 2, CH = CIN / 8, KK = CIN / 16, NCO = COUT / 32;
  // a work unit = NB consecutive 32-pixel blocks (8 KB of input per wave in flight) x NCOW 32-channel tiles (one where
  // the per-lane sums of the STATS / BNB epilogues live in registers: 32 / 48 per tile); the NCG units of a block
  // group go to adjacent waves, which share its input through L1 / L2
  constexpr int NB = CIN >= 128 ? 1 : 128 / CIN;
  constexpr int NCOW = (MODE == PW_BNB || NCO == 1) ? 1 : 2;
  constexpr int NCG = NCO / NCOW;
  static_assert(NCG == 1 || NCG == 2 || NCG == 4, "channel groups must divide the worker count");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem;                                      // [COUT][CINB], 16-B chunks swizzled by the row
  // PW_BNB: [5][COUT] scale, shift, alpha, invstd, -mean invstd; AFF: [2][COUT] scale, shift of the epilogue
  float* ktab = reinterpret_cast<float*>(smem + COUT * CINB);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  // the 16 rows one ds_read_b128 lane group touches must land on 16 different 16-B bank slots (conv_line.hip)
  auto key = [](int row) -> int { return CINB == 64 ? ((row >> 2) & 3) : (CINB == 128 ? ((row >> 1) & 7) : (row & 15)); };

  for (int i = t; i < COUT * CH; i += 256) {
    const int row = i / CH, c = i - row * CH;
    const u32x4 v = *reinterpret_cast<const u32x4*>(p.wp + (long)row * p.ktot + c * 8);
    *reinterpret_cast<u32x4*>(Ws + row * CINB + ((c ^ key(row)) << 4)) = v;
  }
  if (MODE == PW_BNB) {
    for (int c = t; c < COUT; c += 256) {
      const float is = p.bnb.invstd[c];
      ktab[c] = p.bnb.scale[c];
      ktab[COUT + c] = p.bnb.shift[c];
      ktab[2 * COUT + c] = p.bnb.alpha ? p.bnb.alpha[c] : 1.f;
      ktab[3 * COUT + c] = is;
      ktab[4 * COUT + c] = -p.bnb.mean[c] * is;
    }
  } else if (AFF) {
    for (int c = t; c < COUT; c += 256) {
      ktab[c] = p.scale ? p.scale[c] : 1.f;
      ktab[COUT + c] = p.bias ? p.bias[c] : 0.f;
    }
  }
  __syncthreads();

  const int gw = blockIdx.x * 4 + wave, tw = gridDim.x * 4;      // tw % NCG == 0: a worker keeps its channel group
  const int ngrp = (p.nblk + NB - 1) / NB, nunits = ngrp * NCG;
  const int cg = gw % NCG, cbase = cg * NCOW * 32;
  const bool has_alpha = MODE == PW_BNB && p.bnb.alpha != nullptr;

  f32x4 s1[MODE == PW_STATS ? 4 * NCOW : 1], s2[MODE == PW_STATS ? 4 * NCOW : 1];      // channels cbase + 32 i + 8 g + 4 h + j
#pragma unroll
  for (int g = 0; g < (MODE == PW_STATS ? 4 * NCOW : 1); g++) s1[g] = s2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bq[MODE == PW_BNB ? 2 : 1][3][8];               // PW_BNB: [lo / hi chunk][sum][channel]
#pragma unroll
  for (int c = 0; c < (MODE == PW_BNB ? 2 : 1); c++)
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int j = 0; j < 8; j++) bq[c][q][j] = 0.f;

  auto load_x = [&](int grp, u32x4 (&dst)[NB][KK]) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const long m = ((long)grp * NB + b) * 32 + r32;
      const char* src = reinterpret_cast<const char*>(p.in) + m * CINB + h * 16;
      const bool ok = m < p.M;
#pragma unroll
      for (int kk = 0; kk < KK; kk++)
        dst[b][kk] = ok ? *reinterpret_cast<const u32x4*>(src + kk * 32) : u32x4{0, 0, 0, 0};
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
      const long m = ((long)(u / NCG) * NB + b) * 32 + r32;
      const bool valid = m < p.M;
      // second operands of the epilogue fly during the MFMAs: lane (pixel r32, half h) ends up with channels
      // 8 h .. 8 h + 7 and 16 + 8 h .. of every 32-channel tile
      u32x4 e2[(MODE == PW_ADD || MODE == PW_BNB) ? NCOW : 1][2];
      if (MODE == PW_ADD || MODE == PW_BNB) {
        const unsigned short* src = (MODE == PW_ADD ? p.residual : p.bnb.x) + m * COUT + cbase + 8 * h;
#pragma unroll
        for (int i = 0; i < NCOW; i++)
#pragma unroll
          for (int c = 0; c < 2; c++)
            e2[i][c] = valid ? *reinterpret_cast<const u32x4*>(src + 32 * i + 16 * c) : u32x4{0, 0, 0, 0};
      }
      f32x16 acc[NCOW];
#pragma unroll
      for (int i = 0; i < NCOW; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; kk++) {
#pragma unroll
        for (int i = 0; i < NCOW; i++) {
          const int row = cbase + 32 * i + r32;
          const u32x4 a = *reinterpret_cast<const u32x4*>(Ws + row * CINB + (((kk * 2 + h) ^ key(row)) << 4));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                           __builtin_bit_cast(bf16x8, xc[b][kk]), acc[i], 0, 0, 0);
        }
      }
      unsigned short* o = p.out + m * COUT + cbase + 8 * h;
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
        u32x4 ch[2];                                    // channels 8 h .. + 7 and 16 + 8 h .. + 7 of tile i
#pragma unroll
        for (int e = 0; e < 2; e++) {
          auto r01 = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
          auto r23 = __builtin_amdgcn_permlane32_swap(pk[2][e], pk[3][e], false, false);
          ch[0][e] = r01[0]; ch[0][2 + e] = r01[1];
          ch[1][e] = r23[0]; ch[1][2 + e] = r23[1];
        }
        if (MODE == PW_ADD) {
#pragma unroll
          for (int c = 0; c < 2; c++) {
            Vec8 a = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&ch[c]));
            const Vec8 r = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&e2[i][c]));
#pragma unroll
            for (int q = 0; q < 8; q++) a.v[q] += r.v[q];
            store8<unsigned short>(reinterpret_cast<unsigned short*>(&ch[c]), a);
          }
        }
        if (MODE == PW_BNB && valid) {
#pragma unroll
          for (int c = 0; c < 2; c++) {
            BnbCoef k;
            const float* tb = ktab + cbase + 32 * i + 16 * c + 8 * h;
#pragma unroll
            for (int j = 0; j < 8; j++) {
              k.sc[j] = tb[j]; k.sh[j] = tb[COUT + j]; k.al[j] = tb[2 * COUT + j];
              k.is[j] = tb[3 * COUT + j]; k.nm[j] = tb[4 * COUT + j];
            }
            bnb_accum(k, has_alpha, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&ch[c])),
                      load8<unsigned short>(reinterpret_cast<const unsigned short*>(&e2[i][c])), bq[c]);
          }
        }
        if (valid) {
          *reinterpret_cast<u32x4*>(o + 32 * i) = ch[0];
          *reinterpret_cast<u32x4*>(o + 32 * i + 16) = ch[1];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int kk = 0; kk < KK; kk++) xc[b][kk] = xn[b][kk];
  }

  // ---- the worker's sums: over the 32 pixel lanes of each half, then over the waves of the workgroup that share a
  // channel group (LDS), then one f64 atomic per channel and workgroup (all workers finish together: four times fewer
  // atomics queue on each accumulator address)
  if (MODE == PW_STATS || MODE == PW_BNB) {
    constexpr int NQ = MODE == PW_STATS ? 2 : 3;
    constexpr int CW = 32 * NCOW;                       // channels of one worker
    float* red = ktab + 5 * COUT;                       // [4 waves][NQ][CW]
    if (MODE == PW_STATS) {
#pragma unroll
      for (int g = 0; g < 4 * NCOW; g++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float a = s1[g][j], b = s2[g][j];
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
          }
          if (r32 == 0) {
            red[(wave * NQ + 0) * CW + 8 * g + 4 * h + j] = a;      // (g = 4 i + g': channel 32 i + 8 g' + 4 h + j)
            red[(wave * NQ + 1) * CW + 8 * g + 4 * h + j] = b;
          }
        }
    } else {
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
          for (int j = 0; j < 8; j++) {
            float a = bq[c][q][j];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (r32 == 0) red[(wave * NQ + q) * CW + 16 * c + 8 * h + j] = a;
          }
    }
    __syncthreads();
    for (int e = t; e < NCG * NQ * CW; e += 256) {
      const int g = e / (NQ * CW), q = (e / CW) % NQ, c = e % CW;     // wave w of this workgroup serves channel group w % NCG
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; w++)
        if (w % NCG == g) sum += red[(w * NQ + q) * CW + c];
      if (MODE == PW_STATS) stats_emit(reinterpret_cast<float*>(p.stats), 1, blockIdx.x, q, COUT, g * CW + c, sum);
      else bnb_emit(p.bnb.partial, 1, blockIdx.x, q, COUT, g * CW + c, sum);
    }
  }
#endif
}

template <int CIN, int COUT, int MODE, bool AFF>
static void pw_launch(const ConvPwArgs& a, hipStream_t st) {
  constexpr int NCO = COUT / 32, NCOW = (MODE == PW_BNB || NCO == 1) ? 1 : 2, NCG = NCO / NCOW;
  constexpr int NB = CIN >= 128 ? 1 : 128 / CIN;
  const size_t lds = (size_t)COUT * CIN * 2 + 5 * COUT * 4 + 4 * 3 * 64 * 4;
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n;
  }
  // persistent workers: up to 4 workgroups of 4 waves per CU (the register footprint of the instantiation may allow
  // fewer -- the rest queue behind them); never more workers than units
  static const int per_cu = getenv("MSML_PW_WGS_PER_CU") ? atoi(getenv("MSML_PW_WGS_PER_CU")) : 4;
  long grid = (long)cus * per_cu;
  const long ngrp = (a.nblk + NB - 1) / NB;
  const long need = (ngrp * NCG + 3) / 4;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  k_conv_pw<CIN, COUT, MODE, AFF><<<dim3((unsigned)grid), dim3(256), lds, st>>>(a);
}

template <int MODE, bool AFF>
static bool pw_shape(const ConvPwArgs& a, int cin, int cout, hipStream_t st) {
#define PW_CASE(CI, CO) if (cin == CI && cout == CO) { pw_launch<CI, CO, MODE, AFF>(a, st); return true; }
  PW_CASE(32, 64) PW_CASE(64, 32) PW_CASE(64, 64) PW_CASE(64, 128) PW_CASE(128, 64)
#undef PW_CASE
  return false;
}

// Tried by msml_conv_fast_dispatch before everything else; false = not a case this kernel takes.
bool msml_conv_pw_dispatch(const void* in0, int c0p, const void* wp, int kop, int ktot, const float* bias, void* out,
                           int coutp, float* stats, int stats_acc, int N, int H, int W, int P, int Q, int R, int S,
                           int stride, int pad_h, int pad_w, hipStream_t st, const float* scale, const float* alpha,
                           const void* residual, int res_first, const BnBwdFuse* bnb) {
  // Measured against the general kernel on cold data (tools/bench_pw.py, batch 256, one box): forward + statistics
  // 32 -> 64 @ 112x112 186 -> 157 us, @ 56x56 50 -> 47 us; 64 -> 32 @ 56x56 37 -> 37; the 28x28 layers 26-28 -> 29-37 us
  // (1.5 work units per worker: the weight copy of 1 024 workgroups is exposed); backward-data launches +-1 us.  A lane's
  // 16-B loads / stores of one row are one REQUEST each here (adjacent lanes hold different pixels -- the MFMA operand
  // map), four times the request count of the general kernel's LDS-staged 64-B runs, which caps this kernel near
  // 4 TB/s.  Default policy: the launches where it wins (32 -> 64 forward with statistics on >= 2^18 pixels: the
  // im2col'd stems and the first FM stage's conv3); MSML_PW_CONV=all takes every case it supports (the tests), =0 none.
  // (read per call: the tests compare both kernels in one process)
  const char* pol = getenv("MSML_PW_CONV");
  const bool all = pol && pol[0] == 'a';
  if (pol && pol[0] == '0') return false;
  if (!all && !(c0p == 32 && coutp == 64 && stats && !residual && !bnb && (long)N * H * W >= (1L << 18))) return false;
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
