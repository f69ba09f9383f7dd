// 3x3 / stride-1 / pad-1 convolution with 32 input and 32 output channels on bf16 NHWC maps -- conv2 of the first FM
// stage's bottlenecks (backbones/fm/fmoperator.py:53-68 of the reference: 64 -> 32 -> 32 -> 64 at 56x56), forward
// (+ statistics accumulator) and backward-data (+ the BatchNorm backward sums of the layer in front).  Same contract as
// msml_conv2d_acc / msml_conv2d / msml_conv2d_bnbwd_acc for the cases it takes (conv_fast.hip tries it first).
//
// With 64-B pixels the layer is HBM-bound (103 MB for 14.8 GFLOP per launch at batch 256) and the im2col kernel sits at
// 1.5-1.6 TB/s on it (tools/bench_pw.py).  The structure is the pointwise kernel's (conv_pw.hip): every WAVE is an
// independent worker -- here over a strip of image rows -- global memory is touched in whole rows only (lane = 16-B
// chunk of the row's contiguous bytes), the MFMA fragment layout comes from a wave-private LDS scratch, no barrier, no
// hand-placed wait.  A worker keeps a ring of three input rows (with zero halo columns) in LDS: moving one output row
// down costs one new input row, requested two rows ahead into registers; the nine taps are shifted reads of that
// ring.  The packed weight (18 KB) lives in REGISTERS for the whole kernel (18 fragments = 72 VGPRs per lane): the
// LDS port only serves the pixel fragments.  Epilogues as in conv_pw.hip (statistics before the bf16 rounding, sums
// in the copy-out layout, DPP folds, one f64 atomic per channel and workgroup); arithmetic = k_conv_fast's (taps in
// order, 32 channels per tap as two 16-deep MFMAs): outputs bit-identical.
#include <stdlib.h>

#include <mutex>

#include "common.h"

struct ConvR32Args {
  const unsigned short* in;        // [N][H][W][32]
  const unsigned short* wp;        // packed [32 rows][ktot], k = tap * 32 + c
  unsigned short* out;             // [N][H][W][32]
  double* stats;                   // R32_STATS: accumulator double[MSML_ACC_ROWS][2][32]
  BnBwdFuse bnb;                   // R32_BNB (accumulator mode): sums into double[MSML_ACC_ROWS][3][32]
  int N, H, W, ktot;
  int rs, strips, nunits;          // rows per strip, strips per image, N * strips
};

enum { R32_PLAIN = 0, R32_STATS = 1, R32_BNB = 3 };

// FLIP: backward-data launch (tap (r, s) reads the image at (1 - r, 1 - s) instead of (r - 1, s - 1))
template <int MODE, bool FLIP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_conv_r32(const ConvR32Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int W = p.W, H = p.H;
  const int slotb = (W + 2) * 64, scr = 3 * slotb + W * 64;      // a wave's three input rows (+ halo columns) + its output row
  float* red = reinterpret_cast<float*>(smem);          // [4 waves][3][32]
  float* ktab = reinterpret_cast<float*>(smem + 1536);  // R32_BNB: [5][32] scale, shift, alpha, invstd, -mean invstd
  char* ring = smem + 2304 + wave * scr;
  char* outt = ring + 3 * slotb;
  MSML_LDS_REGION(red, 4 * 3 * 32 * 4);
  MSML_LDS_REGION(ktab, 5 * 32 * 4);
  MSML_LDS_REGION(ring, 3 * slotb);
  MSML_LDS_REGION(outt, W * 64);
  // 64-B pixels: the 16 pixels one ds_read_b128 lane group touches need 16 different (quarter-bank, chunk) slots at
  // every tap shift -> key = (pixel >> 2) & 3 (conv_line.hip); the same key keeps a copy-out lane on ONE channel chunk
  auto key = [](int col) -> int { return (col >> 2) & 3; };

  // weights -> registers: A operand of tap t9, k half kk = rows r32, k = t9 * 32 + kk * 16 + h * 8 .. + 7
  u32x4 wf[9][2];
#pragma unroll
  for (int t9 = 0; t9 < 9; t9++)
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
      wf[t9][kk] = *reinterpret_cast<const u32x4*>(p.wp + (long)r32 * p.ktot + t9 * 32 + kk * 16 + h * 8);
  // halo columns 0 and W + 1 of the three slots stay zero for the life of the worker
  if (lane < 24) {
    const int slot = lane >> 3, col = (lane & 4) ? W + 1 : 0;
    *reinterpret_cast<u32x4*>(ring + slot * slotb + col * 64 + (lane & 3) * 16) = u32x4{0, 0, 0, 0};
  }

  const int nch = W * 4;                                // 16-B chunks of one row
  // copy-out / row-load layout: lane <-> chunk j * 64 + lane = pixel 16 j + lane / 4, stored chunk lane & 3
  const int cl = (lane & 3) ^ ((lane >> 4) & 3);        // logical channel chunk of this lane's output chunks (every j)
  const bool has_alpha = MODE == R32_BNB && p.bnb.alpha != nullptr;
  if (MODE == R32_BNB) {                                // (the table is read per use: 40 coefficient registers less per lane)
    if (t < 32) {
      const float is = p.bnb.invstd[t];
      ktab[t] = p.bnb.scale[t];
      ktab[32 + t] = p.bnb.shift[t];
      ktab[64 + t] = p.bnb.alpha ? p.bnb.alpha[t] : 1.f;
      ktab[96 + t] = is;
      ktab[128 + t] = -p.bnb.mean[t] * is;
    }
    __syncthreads();
  }
  f32x4 s1[MODE == R32_STATS ? 4 : 1], s2[MODE == R32_STATS ? 4 : 1];      // channels 8 g + 4 h + j
#pragma unroll
  for (int g = 0; g < (MODE == R32_STATS ? 4 : 1); g++) s1[g] = s2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bq[3][8];
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) bq[q][j] = 0.f;

  auto load_row = [&](int n, int yy, bool want, u32x4 (&dst)[4]) {
    const bool ok = want && yy >= 0 && yy < H;
    const char* src = reinterpret_cast<const char*>(p.in) + ((long)(n * H + (ok ? yy : 0)) * W) * 64 + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; j++)
      dst[j] = (ok && j * 64 + lane < nch) ? *reinterpret_cast<const u32x4*>(src + j * 1024) : u32x4{0, 0, 0, 0};
  };
  auto put_row = [&](int yy, const u32x4 (&src)[4]) {      // row yy -> slot (yy + 3) % 3, pixels at columns 1 .. W
    char* s = ring + ((yy + 3) % 3) * slotb;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = j * 64 + lane, col = (idx >> 2) + 1;
      if (idx < nch) *reinterpret_cast<u32x4*>(s + col * 64 + (((idx & 3) ^ key(col)) << 4)) = src[j];
    }
  };

  const int gw = blockIdx.x * 4 + wave, tw = gridDim.x * 4;
  for (int u = gw; u < p.nunits; u += tw) {
    const int n = u / p.strips, y0 = (u - n * p.strips) * p.rs;
    const int y1 = y0 + p.rs < H ? y0 + p.rs : H;
    u32x4 ra[4], rb[4];
    __builtin_amdgcn_wave_barrier();
    load_row(n, y0 - 1, true, ra);
    load_row(n, y0, true, rb);
    put_row(y0 - 1, ra);
    put_row(y0, rb);
    load_row(n, y0 + 1, true, ra);
    put_row(y0 + 1, ra);
    load_row(n, y0 + 2, y0 + 2 <= y1, ra);              // rows y0 + 2, y0 + 3 fly during the first output rows
    load_row(n, y0 + 3, y0 + 3 <= y1, rb);

    // one output row: `nx` holds input row y + 2 (requested two rows ago); afterwards it is refilled with row y + 4
    auto row = [&](int y, u32x4 (&nx)[4]) {
      const long pix0 = (long)(n * H + y) * W;
      const int so[3] = {((y + 2) % 3) * slotb, (y % 3) * slotb, ((y + 1) % 3) * slotb};      // rows y - 1, y, y + 1
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int bl = 0; bl < 2; bl++) {
        if (bl * 32 >= W) break;
        const int x = bl * 32 + r32;
        const bool valid = x < W;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
        for (int t9 = 0; t9 < 9; t9++) {
          const int r = t9 / 3, s = t9 % 3;
          const int rr = FLIP ? 2 - r : r, col = x + (FLIP ? 2 - s : s);      // window row 0 .. 2, ring column (image x + s - 1, + 1)
          const char* px = ring + so[rr] + col * 64;
          const int k = key(col);
#pragma unroll
          for (int kk = 0; kk < 2; kk++) {
            const u32x4 xb = *reinterpret_cast<const u32x4*>(px + (((kk * 2 + h) ^ k) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[t9][kk]),
                                                          __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
          }
        }
        // (running both blocks through ONE tap loop -- two independent accumulator chains -- measured no faster: the
        // worker waits on memory, not on MFMA latency)
        u32x2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            v[j] = acc[g * 4 + j];
            if (MODE == R32_STATS && valid) {
              s1[g][j] += v[j];
              s2[g][j] += v[j] * v[j];
            }
          }
          pk[g][0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
          pk[g][1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
        }
        // lane (pixel x, half h): channels 8 h .. + 7 and 16 + 8 h .. + 7 after the swaps (conv_line.hip)
        u32x4 ch[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          auto r01 = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
          auto r23 = __builtin_amdgcn_permlane32_swap(pk[2][e], pk[3][e], false, false);
          ch[0][e] = r01[0]; ch[0][2 + e] = r01[1];
          ch[1][e] = r23[0]; ch[1][2 + e] = r23[1];
        }
        if (valid) {
#pragma unroll
          for (int c = 0; c < 2; c++)
            *reinterpret_cast<u32x4*>(outt + x * 64 + (((2 * c + h) ^ key(x)) << 4)) = ch[c];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // (the saved BatchNorm input of this row, in the copy-out layout; requested after the MFMAs: 16 registers that
      // would otherwise live across them put the instantiation over 256 VGPRs)
      u32x4 e2[MODE == R32_BNB ? 4 : 1];
      if (MODE == R32_BNB) {
#pragma unroll
        for (int j = 0; j < 4; j++)
          e2[j] = j * 64 + lane < nch ? *reinterpret_cast<const u32x4*>(p.bnb.x + (pix0 + 16 * j + (lane >> 2)) * 32 + cl * 8)
                                      : u32x4{0, 0, 0, 0};
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int idx = j * 64 + lane;
        if (idx < nch) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(outt + idx * 16);
          if (MODE == R32_BNB) {
            BnbCoef bk;
            const float* tb = ktab + cl * 8;
#pragma unroll
            for (int q = 0; q < 8; q++) {
              bk.sc[q] = tb[q]; bk.sh[q] = tb[32 + q]; bk.al[q] = tb[64 + q]; bk.is[q] = tb[96 + q]; bk.nm[q] = tb[128 + q];
            }
            bnb_accum(bk, has_alpha, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v)),
                      load8<unsigned short>(reinterpret_cast<const unsigned short*>(&e2[j])), bq);
          }
          *reinterpret_cast<u32x4*>(p.out + (pix0 + (idx >> 2)) * 32 + cl * 8) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
      put_row(y + 2, nx);                               // over row y - 1, which no later output row reads
      load_row(n, y + 4, y + 4 <= y1, nx);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int y = y0; y < y1; y += 2) {
      row(y, ra);
      if (y + 1 < y1) row(y + 1, rb);
    }
  }

  if (MODE == R32_STATS || MODE == R32_BNB) {
    constexpr int NQ = MODE == R32_STATS ? 2 : 3;
    if (MODE == R32_STATS) {
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float a = wave_half_sum(s1[g][j]), b = wave_half_sum(s2[g][j]);
          if (r32 == 0) {
            red[(wave * NQ + 0) * 32 + 8 * g + 4 * h + j] = a;
            red[(wave * NQ + 1) * 32 + 8 * g + 4 * h + j] = b;
          }
        }
    } else {
      // lanes that share cl = (lane & 3) ^ ((lane >> 4) & 3): flip bit k of both sides (lane bits k and k + 4), and the
      // pixel bits 2, 3
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          float a = bq[q][j];
          a += __shfl_xor(a, 17, 64);
          a += __shfl_xor(a, 34, 64);
          a += __shfl_xor(a, 4, 64);
          a += __shfl_xor(a, 8, 64);
          if (lane < 4) red[(wave * NQ + q) * 32 + 8 * lane + j] = a;      // (lanes 0 .. 3: cl = lane)
        }
    }
    __syncthreads();
    if (t < NQ * 32) {
      const int q = t >> 5, c = t & 31;
      const float sum = (red[(0 * NQ + q) * 32 + c] + red[(1 * NQ + q) * 32 + c]) +
                        (red[(2 * NQ + q) * 32 + c] + red[(3 * NQ + q) * 32 + c]);
      if (MODE == R32_STATS) stats_emit(reinterpret_cast<float*>(p.stats), 1, blockIdx.x, q, 32, c, sum);
      else bnb_emit(p.bnb.partial, 1, blockIdx.x, q, 32, c, sum);
    }
  }
#endif
}

template <int MODE, bool FLIP>
static void r32_launch(ConvR32Args& a, hipStream_t st) {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n;
  }
  // strips: at least ~8 workers per CU, at most 14 rows each (two halo rows are re-read per strip)
  static const int rows_env = getenv("MSML_R32_ROWS") ? atoi(getenv("MSML_R32_ROWS")) : 0;
  int rs = 14;
  while (rs > 4 && (long)a.N * ((a.H + rs - 1) / rs) < 8L * cus) rs = (rs + 1) / 2;
  if (rows_env > 0) rs = rows_env;
  a.rs = rs;
  a.strips = (a.H + rs - 1) / rs;
  a.nunits = a.N * a.strips;
  const size_t lds = 2304 + 4 * (size_t)(3 * (a.W + 2) * 64 + a.W * 64) + 1024;     // (+ masked lanes' reads past the last row)
  if (lds > 64 * 1024) {                                // rows wider than 58 pixels: above the default dynamic-LDS limit
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_r32<MODE, FLIP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    });
  }
  long grid = (a.nunits + 3) / 4;
  if (grid > 2L * cus) grid = 2L * cus;
  k_conv_r32<MODE, FLIP><<<dim3((unsigned)grid), dim3(256), lds, st>>>(a);
}

// Tried by msml_conv_fast_dispatch first; false = not a case this kernel takes.
bool msml_conv_r32_dispatch(const void* in0, int c0p, const void* wp, int kop, int ktot, const float* bias, void* out,
                            int coutp, float* stats, int stats_acc, int N, int H, int W, int P, int Q, int R, int S,
                            int stride, int pad_h, int pad_w, int transposed, hipStream_t st, const float* scale,
                            const float* alpha, const void* residual, const BnBwdFuse* bnb) {
  if (getenv("MSML_NO_R32_CONV")) return false;           // (read per call: the test compares both kernels in one process)
  if (c0p != 32 || coutp != 32 || R != 3 || S != 3 || stride != 1 || pad_h != 1 || pad_w != 1 || P != H || Q != W) return false;
  if (bias || scale || alpha || residual) return false;
  if (stats && !stats_acc) return false;
  if (bnb && (!bnb->acc || stats)) return false;
  if (W < 8 || W > 62 || H < 2 || kop < 32 || ktot < 288) return false;
  if ((long)N * H * W * 64 >= 0x7fffff00L) return false;
  ConvR32Args a;
  a.in = (const unsigned short*)in0; a.wp = (const unsigned short*)wp; a.out = (unsigned short*)out;
  a.stats = reinterpret_cast<double*>(stats);
  a.bnb = BnBwdFuse{};
  if (bnb) a.bnb = *bnb;
  a.N = N; a.H = H; a.W = W; a.ktot = ktot;
  if (transposed) {
    if (bnb) r32_launch<R32_BNB, true>(a, st);
    else if (stats) r32_launch<R32_STATS, true>(a, st);
    else r32_launch<R32_PLAIN, true>(a, st);
  } else {
    if (bnb) r32_launch<R32_BNB, false>(a, st);
    else if (stats) r32_launch<R32_STATS, false>(a, st);
    else r32_launch<R32_PLAIN, false>(a, st);
  }
  return true;
}
