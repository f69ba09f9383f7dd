// Weight-gradient implicit GEMM on MFMA (gfx950).
//
//   dW[a][b][r][s] = sum_{n,py,px} U[n,py,px,a] * V[n, py*stride - pad_h + r, px*stride - pad_w + s, b]
//
// U is the operand on the natural pixel grid (P x Q), V the shifted one (H x W):
//   Conv2d wgrad:           U = dY (a = cout), V = X  (b = cin)   -> dW[cout][cin][r][s]
//   ConvTranspose2d wgrad:  U = X  (a = cin),  V = dY (b = cout)  -> dWt[cin][cout][r][s]
//   Linear / PartialFC:     1x1 (or HxW valid) windows of the same form.
// Replaces the autograd weight gradients of every conv/deconv/linear cited in conv_igemm.hip
// and of headers/partial_fc.py:169 (sub_weight.grad).
//
// GEMM view: rows = a, cols = (tap, b), K = pixels (N*P*Q, up to 3.2 M) -> split-K over pixel
// ranges; every split writes an f32 partial slab ws[split][a][tap][b] with plain stores and a
// second kernel sums the slabs in a fixed order (deterministic, no float atomics) and scatters
// into the parameter's own [A][Btot][R][S] layout.
//
// Both operands are pixel-major (channels contiguous), i.e. K-major: bf16 fragments are read
// from LDS with the gfx950 transposing read ds_read_b64_tr_b16 (rows = pixels, 16-channel
// blocks); LDS rows are padded to a pitch == 64 (mod 256) bytes so the four pixel rows of a
// block fall in disjoint bank windows.  f32 fragments are plain ds_read_b32 (one k per lane).
#include <stdlib.h>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct WgradArgs {
  const void* u; int up;      // U: [N][P][Q][up]
  const void* v; int vp;      // V: [N][H][W][vp]
  int N, H, W, P, Q, R, S, stride, pad_h, pad_w;
  float* ws;                  // [splits][arows][taps][vp]
  int arows;                  // rows of a covered by the slab (= up)
  long Mpix;                  // N*P*Q
  int chunk;                  // pixels per split (multiple of 32)
  float rcp_pq, rcp_q;
};

__device__ __forceinline__ void divmod(int m, int d, float rcp, int& q, int& r) {
  q = (int)((float)m * rcp);
  r = m - q * d;
  if (r < 0) { r += d; q--; }
  if (r >= d) { r -= d; q++; }
}

template <typename T, int BA, int BB>
struct WTile {
  static constexpr int ES = sizeof(T);
  // pitch in bytes per pixel row of the U / V tiles
  static constexpr int PU = BA * ES + (ES == 2 ? 64 : 0);
  static constexpr int PV = BB * ES + (ES == 2 ? 64 : 0);
  static constexpr int STAGE = 32 * (PU + PV);
};

__device__ __forceinline__ s16x8 tr_frag(const char* base, int pitch, int chan0, int pix0, int lane) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const char* addr = base + (pix0 + 8 * (g >> 1) + q) * pitch + (chan0 + 16 * (g & 1) + 4 * pp) * 2;
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(addr));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(addr + 4 * pitch));
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

template <typename T, int BA, int BB>
__global__ void __launch_bounds__(256) k_conv_wgrad(const WgradArgs p) {
  using WT = WTile<T, BA, BB>;
  constexpr int CH = 16 / sizeof(T);           // elements per 16-B chunk
  constexpr int CU = BA / CH, CV = BB / CH;    // chunks per pixel row
  constexpr int PPU = 256 / CU, PPV = 256 / CV;   // pixels covered per pass
  constexpr int NU = 32 / PPU > 0 ? 32 / PPU : 1, NV = 32 / PPV > 0 ? 32 / PPV : 1;
  constexpr bool UPART = PPU > 32, VPART = PPV > 32;
  constexpr int TM = BA / 64, TN = BB / 64;    // 2x2 waves, each (BA/2) x (BB/2)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  MSML_LDS_REGION(smem, 2 * WT::STAGE);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int btiles = (p.vp + BB - 1) / BB;
  const int a0 = blockIdx.x * BA;
  const int tap = blockIdx.y / btiles, b0 = (blockIdx.y % btiles) * BB;
  const int r = tap / p.S, s = tap % p.S;
  const int split = blockIdx.z;
  const long k_begin = (long)split * p.chunk;
  long k_end = k_begin + p.chunk;
  if (k_end > p.Mpix) k_end = p.Mpix;
  const int nsteps = k_begin < k_end ? (int)((k_end - k_begin + 31) / 32) : 0;
  const int PQ = p.P * p.Q;

  const int ucc = t % CU, upx = t / CU;
  const int vcc = t % CV, vpx = t / CV;
  const bool u_chan_ok = a0 + ucc * CH < p.up;
  const bool v_chan_ok = b0 + vcc * CH < p.vp;
  const T* U = reinterpret_cast<const T*>(p.u);
  const T* V = reinterpret_cast<const T*>(p.v);

  u32x4 ru[NU], rv[NV];
  auto gload = [&](int step) {
    const long kb = k_begin + (long)step * 32;
#pragma unroll
    for (int i = 0; i < NU; i++) {
      int j = upx + i * PPU;
      long m = kb + j;
      u32x4 val = {0u, 0u, 0u, 0u};
      if ((!UPART || j < 32) && m < k_end && u_chan_ok)
        val = *reinterpret_cast<const u32x4*>(U + m * p.up + a0 + ucc * CH);
      ru[i] = val;
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      int j = vpx + i * PPV;
      long m = kb + j;
      u32x4 val = {0u, 0u, 0u, 0u};
      if ((!VPART || j < 32) && m < k_end && v_chan_ok) {
        int n, rem, py, px;
        divmod((int)m, PQ, p.rcp_pq, n, rem);
        divmod(rem, p.Q, p.rcp_q, py, px);
        int iy = py * p.stride - p.pad_h + r, ix = px * p.stride - p.pad_w + s;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          val = *reinterpret_cast<const u32x4*>(V + ((long)(n * p.H + iy) * p.W + ix) * p.vp + b0 +
                                                vcc * CH);
      }
      rv[i] = val;
    }
  };
  auto lstore = [&](int buf) {
    char* ub = smem + buf * WT::STAGE;
    char* vb = ub + 32 * WT::PU;
#pragma unroll
    for (int i = 0; i < NU; i++) {
      int j = upx + i * PPU;
      if (!UPART || j < 32) *reinterpret_cast<u32x4*>(ub + j * WT::PU + ucc * 16) = ru[i];
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      int j = vpx + i * PPV;
      if (!VPART || j < 32) *reinterpret_cast<u32x4*>(vb + j * WT::PV + vcc * 16) = rv[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int arow0 = wm * (BA / 2), bcol0 = wn * (BB / 2);

  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < nsteps; step++) {
    const bool more = step + 1 < nsteps;
    if (more) gload(step + 1);
    const char* ub = smem + cur * WT::STAGE;
    const char* vb = ub + 32 * WT::PU;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        s16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) a[i] = tr_frag(ub, WT::PU, arow0 + 32 * i, 16 * kk, lane);
#pragma unroll
        for (int j = 0; j < TN; j++) b[j] = tr_frag(vb, WT::PV, bcol0 + 32 * j, 16 * kk, lane);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
      }
    } else {
      const int i32 = lane & 31, h = lane >> 5;
#pragma unroll 4
      for (int kk = 0; kk < 16; kk++) {
        float a[TM], b[TN];
        const int pix = 2 * kk + h;
#pragma unroll
        for (int i = 0; i < TM; i++)
          a[i] = *reinterpret_cast<const float*>(ub + pix * WT::PU + (arow0 + 32 * i + i32) * 4);
#pragma unroll
        for (int j = 0; j < TN; j++)
          b[j] = *reinterpret_cast<const float*>(vb + pix * WT::PV + (bcol0 + 32 * j + i32) * 4);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // partial slab: ws[split][a][tap][b]
  const int h = lane >> 5, c32 = lane & 31;
  const int taps = p.R * p.S;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int b = b0 + bcol0 + 32 * j + c32;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int a = a0 + arow0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (a < p.arows && b < p.vp)
          p.ws[(((long)split * p.arows + a) * taps + tap) * p.vp + b] = acc[i][j][e];
      }
    }
}

// dst[a][boff + b][tap] = sum_split ws[split][a][tap][b]   (a < A, b < Breal)
// One workgroup per (output row a, tap, block of 64 columns b); its four waves are split lanes: wave w
// sums the splits w, w + 4, w + 8, ... with eight independent partial sums (eight 256-B coalesced slab
// loads in flight per wave: the first version kept two, and the kernel is bound by bytes in flight,
// not by bandwidth), the four lanes are combined through LDS.  Every addition has a fixed place in
// the tree: deterministic.
// (blockIdx.y = layer of a grouped launch: its slabs start at ws + layer * splits * slab, its gradient is dsts.p[layer])
#define WR_MAXGROUP 8
struct DwPtrs { float* p[WR_MAXGROUP]; };
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ ws, const DwPtrs dsts,
                                                      int splits, int arows, int taps, int vp, int A,
                                                      int Breal, int Btot, int boff, int accumulate) {
  __shared__ float red[4][64];
  const long slab = (long)arows * taps * vp;
  ws += (long)blockIdx.y * splits * slab;
  float* __restrict__ dst = dsts.p[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bblocks = (vp + 63) >> 6;
  const int at = blockIdx.x / bblocks, b = ((blockIdx.x - at * bblocks) << 6) + lane;   // at = a * taps + tap
  float q[8];
#pragma unroll
  for (int k = 0; k < 8; k++) q[k] = 0.f;
  if (b < vp) {
    const float* p = ws + (long)at * vp + b;
    int sp = wave;
    for (; sp + 28 < splits; sp += 32) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = p[(long)(sp + 4 * k) * slab];
#pragma unroll
      for (int k = 0; k < 8; k++) q[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (sp + 4 * k < splits) q[k] += p[(long)(sp + 4 * k) * slab];
  }
  red[wave][lane] = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
  __syncthreads();
  if (wave == 0 && b < Breal) {
    const float sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int a = at / taps, tap = at - a * taps;
    const long o = ((long)a * Btot + boff + b) * taps + tap;
    dst[o] = accumulate ? dst[o] + sum : sum;
  }
}

// Variant for large outputs (>= 512 (row, column block) pairs): one workgroup per (a, block of 64
// columns) walks over ALL taps (wave w: taps w, w + 4, ...), each lane summing its splits with eight
// loads in flight, and the results cross LDS so that the parameter-layout run
// dst[a][b0 .. b0 + 63][0 .. taps) leaves as one contiguous stream instead of 4-B elements `taps` floats
// apart.  256 -> 256 @ 14x14: weight gradient + reduce 97 -> 87 us (the per-tap kernel: 89).
#define WR_MAXTAPS 49
__global__ void __launch_bounds__(256) k_wgrad_reduce_rows(const float* __restrict__ ws, const DwPtrs dsts,
                                                           int splits, int arows, int taps, int vp, int A,
                                                           int Breal, int Btot, int boff, int accumulate) {
  __shared__ float tile[WR_MAXTAPS * 65];
  const long slab = (long)arows * taps * vp;
  ws += (long)blockIdx.y * splits * slab;
  float* __restrict__ dst = dsts.p[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bblocks = (vp + 63) >> 6;
  const int a = blockIdx.x / bblocks, b0 = (blockIdx.x - a * bblocks) << 6;
  const int b = b0 + lane;
  for (int tap = wave; tap < taps; tap += 4) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = 0.f;
    if (b < vp) {
      const float* p = ws + ((long)a * taps + tap) * vp + b;
      int sp = 0;
      for (; sp + 8 <= splits; sp += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = p[(long)(sp + k) * slab];
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] += v[k];
      }
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (sp + k < splits) acc[k] += p[(long)(sp + k) * slab];
    }
    tile[tap * 65 + lane] = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
  }
  __syncthreads();
  const int nb = Breal - b0 < 64 ? Breal - b0 : 64;    // real columns of this block
  float* out = dst + ((long)a * Btot + boff + b0) * taps;
  for (int j = threadIdx.x; j < nb * taps; j += 256) {
    const int bl = j / taps, tap = j - bl * taps;
    const float sum = tile[tap * 65 + bl];
    out[j] = accumulate ? out[j] + sum : sum;
  }
}

// (the choice depends on the layer shape only, so a layer always sums in the same order)
static void wgrad_reduce_launch_group(const float* ws, float* const* dw, int group, int splits, int up, int taps, int vp,
                                      int A, int Breal, int Btot, int boff, int accumulate, hipStream_t st) {
  DwPtrs d;
  for (int i = 0; i < WR_MAXGROUP; i++) d.p[i] = dw[i < group ? i : 0];
  const int rowblocks = A * ((vp + 63) / 64);
  if (rowblocks >= 512 && taps > 1 && taps <= WR_MAXTAPS)
    k_wgrad_reduce_rows<<<dim3(rowblocks, group), 256, 0, st>>>(ws, d, splits, up, taps, vp, A, Breal, Btot, boff, accumulate);
  else
    k_wgrad_reduce<<<dim3(rowblocks * taps, group), 256, 0, st>>>(ws, d, splits, up, taps, vp, A, Breal, Btot, boff,
                                                                 accumulate);
}

static void wgrad_reduce_launch(const float* ws, float* dw, int splits, int up, int taps, int vp, int A, int Breal,
                                int Btot, int boff, int accumulate, hipStream_t st) {
  wgrad_reduce_launch_group(ws, &dw, 1, splits, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
}

bool msml_wgrad_fast_launch(const void* u, int up, const void* v, int vp, float* ws, int N, int H, int W,
                            int P, int Q, int R, int S, int stride, int pad_h, int pad_w, int ba, int bb,
                            int ntw, int splits, int chunk, hipStream_t st, float* dw_direct, int A, int Breal,
                            int Btot, int boff, int accumulate);

bool msml_wgrad_fast_launch_group(const void* const* u, int up, const void* const* v, int vp, float* ws, int N, int H,
                                  int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w, int ba, int bb,
                                  int ntw, int group, int splits, int chunk, hipStream_t st, float* const* dw_direct,
                                  int A, int Breal, int Btot, int boff, int accumulate);

int msml_wgrad_halo_splits(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q, int R, int S,
                           int stride, int pad_h, int pad_w);
int msml_wgrad_halo_s2_splits(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q, int R, int S, int stride,
                              int pad_h, int pad_w);
bool msml_wgrad_halo_s2_launch(const void* u, int up, const void* v, int vp, float* ws, int N, int H, int W, int P, int Q,
                               int splits, hipStream_t st);
bool msml_wgrad_halo_launch(const void* u, int up, const void* v, int vp, float* ws, int N, int H, int W,
                            int splits, hipStream_t st, const BnIn* xin = nullptr);
int msml_wgrad_halo_group_splits(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q, int R, int S,
                                 int stride, int pad_h, int pad_w, int group);
bool msml_wgrad_halo_launch_group(const void* const* u, int up, const void* const* v, int vp, float* ws, int N, int H,
                                  int W, int group, int splits, hipStream_t st, const BnIn* xin);

int msml_fc_wgrad_launch(const void* u, int up, const void* v, int vp, float* dw, int A, int Breal, int Btot, int boff,
                         int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                         int accumulate, hipStream_t st);      // fc_wgrad.hip
int msml_wgrad_n32_splits(int up, int vp, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                          int pad_w);
bool msml_wgrad_n32_launch(const void* u, const void* v, float* ws, int N, int H, int W, int P, int Q, int R,
                           int stride, int splits, hipStream_t st);
int msml_wgrad_line_splits(int up, int vp, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                           int pad_w);
bool msml_wgrad_line_launch(const void* u, const void* v, int vp, float* ws, int N, int H, int R, int splits,
                            hipStream_t st);

// taps handled by one workgroup of the bf16 fast kernel (narrow V operands share the U tile)
static int wgrad_ntw(int vp, int taps) {
  static const bool off = getenv("MSML_WGRAD_NO_MULTITAP") != nullptr;
  // 128-channel layers: 3 taps x one 64-channel chunk per workgroup (125 -> 93 us at 128->128@28x28);
  // neutral from 256 channels on (the wider tile's slab traffic eats the fill saving)
  static const int wide = getenv("MSML_WGRAD_MULTITAP_WIDE") ? atoi(getenv("MSML_WGRAD_MULTITAP_WIDE")) : 128;
  if (off || taps < 3) return 1;
  if (vp <= 64) return 3;
  return (wide && vp % 64 == 0 && vp <= wide) ? 3 : 1;     // 3 taps x one 64-channel chunk per workgroup
}

static int pick_tile(int c) { return c > 64 ? 128 : 64; }

// number of pixel splits: enough workgroups to fill 256 CUs a few times over, chunks >= 256 px.
// Every workgroup writes one f32 tile (<= 64 KB) of partial sums, so the slab traffic is
// workgroups x tile bytes whatever the layer: with few output tiles (64 / 128-channel layers,
// |dW| of a few hundred KB) half as many workgroups win, with many tiles (512 channels) 768;
// measured per shape with tools/bench_conv.py (MSML_WGRAD_WGS overrides).
static int pick_splits(long mpix, int out_tiles) {
  static const long forced = getenv("MSML_WGRAD_WGS") ? atol(getenv("MSML_WGRAD_WGS")) : 0;
  static const long minchunk = getenv("MSML_WGRAD_MINCHUNK") ? atol(getenv("MSML_WGRAD_MINCHUNK")) : 256;
  long target = 1024;
  if (out_tiles <= 9 && mpix <= (1L << 20)) target = 512;
  else if (out_tiles > 48) target = 768;
  if (forced > 0) target = forced;
  long want = (target + out_tiles - 1) / out_tiles;
  long max_by_chunk = (mpix + minchunk - 1) / minchunk;
  long s = want < max_by_chunk ? want : max_by_chunk;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return (int)s;
}

extern "C" long msml_conv_wgrad_workspace(int up, int vp, int N, int P, int Q, int R, int S) {
  int ba = pick_tile(up), bb = pick_tile(vp);
  int tiles = cdiv(up, ba) * cdiv(vp, bb) * R * S;
  int splits = pick_splits((long)N * P * Q, tiles);      // upper bound over both dtypes' tilings
  int nt = wgrad_ntw(vp, R * S);
  if (nt > 1) {
    int s2 = pick_splits((long)N * P * Q, cdiv(up, ba) * cdiv(R * S, nt) * cdiv(vp, 64));
    if (s2 > splits) splits = s2;
  }
  // the halo kernel (wgrad_halo.hip) may use up to one split per CU-resident workgroup
  if (R == 3 && S == 3 && splits < 512) {
    int hs = msml_wgrad_halo_splits(up, vp, up, vp, N, P, Q, P, Q, 3, 3, 1, 1, 1);
    if (hs > splits) splits = hs;
  }
  // narrow-operand kernel (wgrad_n32.hip): H / W are not known here, assume its largest split count
  if (up == 32 && vp == 32 && (R * S == 16 || R * S == 9) && splits < 512) splits = 512;
  if (up == 32 && (vp == 32 || vp == 64) && R * S == 7 && splits < 512) splits = 512;          // line kernel
  return (long)splits * up * R * S * vp * (long)sizeof(float);
}

// 1 when the bf16 weight gradient of this shape runs on the narrow-operand kernel (tests, profiling labels)
extern "C" int msml_conv_wgrad_kernel_is_n32(int up, int vp, int N, int H, int W, int P, int Q, int R, int S, int stride,
                                             int pad_h, int pad_w) {
  return (msml_wgrad_n32_splits(up, vp, N, H, W, P, Q, R, S, stride, pad_h, pad_w) > 0 ||
          msml_wgrad_line_splits(up, vp, N, H, W, P, Q, R, S, stride, pad_h, pad_w) > 0) ? 1 : 0;
}

extern "C" int msml_conv_wgrad(const void* u, int up, const void* v, int vp, float* dw, int A,
                               int Breal, int Btot, int boff, int N, int H, int W, int P, int Q,
                               int R, int S, int stride, int pad_h, int pad_w, int accumulate,
                               void* workspace, long ws_bytes, int dtype, void* stream) {
  MSML_CHECK(u && v && dw && workspace, MSML_ERR_SHAPE, "conv_wgrad: null pointer");
  MSML_CHECK(up > 0 && up % 8 == 0 && vp > 0 && vp % 8 == 0 && A > 0 && A <= up && Breal > 0 &&
                 Breal <= vp && boff >= 0 && boff + Breal <= Btot,
             MSML_ERR_SHAPE, "conv_wgrad: bad channels up=%d vp=%d A=%d Breal=%d Btot=%d boff=%d", up,
             vp, A, Breal, Btot, boff);
  MSML_CHECK(N > 0 && H > 0 && W > 0 && P > 0 && Q > 0 && R > 0 && S > 0 && stride >= 1,
             MSML_ERR_SHAPE, "conv_wgrad: bad dims");
  MSML_CHECK((long)N * P * Q < (1L << 24), MSML_ERR_UNSUPPORTED,
             "conv_wgrad: N*P*Q = %ld exceeds 2^24 (float-reciprocal pixel decode)", (long)N * P * Q);
  long need = msml_conv_wgrad_workspace(up, vp, N, P, Q, R, S);
  MSML_CHECK(ws_bytes >= need, MSML_ERR_WORKSPACE, "conv_wgrad: workspace %ld < %ld bytes", ws_bytes, need);
  WgradArgs a;
  a.u = u; a.up = up; a.v = v; a.vp = vp;
  a.N = N; a.H = H; a.W = W; a.P = P; a.Q = Q; a.R = R; a.S = S;
  a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.ws = (float*)workspace;
  a.arows = up;
  a.Mpix = (long)N * P * Q;
  a.rcp_pq = 1.0f / (float)(P * Q);
  a.rcp_q = 1.0f / (float)Q;
  const int ba = pick_tile(up), bb = pick_tile(vp);
  const int atiles = cdiv(up, ba), btiles = cdiv(vp, bb), taps = R * S;
  const int ntw = dtype == MSML_BF16 ? wgrad_ntw(vp, taps) : 1;
  const int splits = pick_splits(a.Mpix, ntw > 1 ? atiles * cdiv(taps, ntw) * cdiv(vp, 64) : atiles * btiles * taps);
  a.chunk = (int)(((a.Mpix + splits - 1) / splits + 63) / 64 * 64);
  dim3 grid(atiles, btiles * taps, splits);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSML_BF16) {
    if (msml_fc_wgrad_launch(u, up, v, vp, dw, A, Breal, Btot, boff, N, H, W, P, Q, R, S, stride, pad_h, pad_w,
                             accumulate, st)) {
      MSML_LAUNCH_OK("conv_wgrad(fc)");
      return MSML_OK;
    }
    const int ns = msml_wgrad_n32_splits(up, vp, N, H, W, P, Q, R, S, stride, pad_h, pad_w);
    if (ns > 0 && msml_wgrad_n32_launch(u, v, a.ws, N, H, W, P, Q, R, stride, ns, st)) {
      MSML_LAUNCH_OK("conv_wgrad(n32)");
      wgrad_reduce_launch(a.ws, dw, ns, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
      MSML_LAUNCH_OK("conv_wgrad_reduce");
      return MSML_OK;
    }
    const int ls = msml_wgrad_line_splits(up, vp, N, H, W, P, Q, R, S, stride, pad_h, pad_w);
    if (ls > 0 && msml_wgrad_line_launch(u, v, vp, a.ws, N, H, R, ls, st)) {
      MSML_LAUNCH_OK("conv_wgrad(line)");
      wgrad_reduce_launch(a.ws, dw, ls, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
      MSML_LAUNCH_OK("conv_wgrad_reduce");
      return MSML_OK;
    }
    const int hs2 = msml_wgrad_halo_s2_splits(up, vp, A, Breal, N, H, W, P, Q, R, S, stride, pad_h, pad_w);
    if (hs2 > 0 && (long)hs2 * up * taps * vp * (long)sizeof(float) <= (long)ws_bytes &&
        msml_wgrad_halo_s2_launch(u, up, v, vp, a.ws, N, H, W, P, Q, hs2, st)) {
      MSML_LAUNCH_OK("conv_wgrad(halo, stride 2)");
      wgrad_reduce_launch(a.ws, dw, hs2, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
      MSML_LAUNCH_OK("conv_wgrad_reduce");
      return MSML_OK;
    }
    const int hs = msml_wgrad_halo_splits(up, vp, A, Breal, N, H, W, P, Q, R, S, stride, pad_h, pad_w);
    if (hs > 0 && msml_wgrad_halo_launch(u, up, v, vp, a.ws, N, H, W, hs, st)) {
      MSML_LAUNCH_OK("conv_wgrad(halo)");
      wgrad_reduce_launch(a.ws, dw, hs, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
      MSML_LAUNCH_OK("conv_wgrad_reduce");
      return MSML_OK;
    }
  }
  if (dtype == MSML_BF16 && !getenv("MSML_NO_FAST_WGRAD") &&
      msml_wgrad_fast_launch(u, up, v, vp, a.ws, N, H, W, P, Q, R, S, stride, pad_h, pad_w, ba, bb, ntw,
                             splits, a.chunk, st, dw, A, Breal, Btot, boff, accumulate)) {
    MSML_LAUNCH_OK("conv_wgrad(fast)");
    if (splits == 1) return MSML_OK;                   // written straight into dw
    wgrad_reduce_launch(a.ws, dw, splits, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
    MSML_LAUNCH_OK("conv_wgrad_reduce");
    return MSML_OK;
  }

#define WG_LAUNCH(T, BA_, BB_)                                                    \
  {                                                                               \
    typedef WTile<T, BA_, BB_> WT_;                                               \
    k_conv_wgrad<T, BA_, BB_><<<grid, dim3(256), 2 * WT_::STAGE, st>>>(a);        \
  }
#define WG_CASE(T)                                            \
  if (ba == 128 && bb == 128) WG_LAUNCH(T, 128, 128)          \
  else if (ba == 128) WG_LAUNCH(T, 128, 64)                   \
  else if (bb == 128) WG_LAUNCH(T, 64, 128)                   \
  else WG_LAUNCH(T, 64, 64)
  if (dtype == MSML_F32) { WG_CASE(float) }
  else if (dtype == MSML_BF16) { WG_CASE(unsigned short) }
  else {
    msml_set_error("conv_wgrad: unsupported dtype %d", dtype);
    return MSML_ERR_DTYPE;
  }
  MSML_LAUNCH_OK("conv_wgrad");
  wgrad_reduce_launch(a.ws, dw, splits, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
  MSML_LAUNCH_OK("conv_wgrad_reduce");
  return MSML_OK;
}

// Splits per layer of the im2col kernel when `group` layers share a launch: the workgroup budget of ONE layer's launch
// spread over the group (>= 1; 1 = every layer's tile goes straight into its dW, no slabs, no reduce).
static int fast_group_splits(long mpix, int out_tiles, int group) {
  const int one = pick_splits(mpix, out_tiles);
  int per = one / group;
  return per < 1 ? 1 : per;
}

// Largest number of same-shape layers msml_conv_wgrad_group accepts for this shape (1: no grouping).
extern "C" int msml_conv_wgrad_group_max(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q, int R,
                                         int S, int stride, int pad_h, int pad_w) {
  if (msml_wgrad_halo_splits(up, vp, A, Breal, N, H, W, P, Q, R, S, stride, pad_h, pad_w) > 0) {
    int g = 1;
    while (g * 2 <= WR_MAXGROUP &&
           msml_wgrad_halo_group_splits(up, vp, A, Breal, N, H, W, P, Q, R, S, stride, pad_h, pad_w, g * 2) >= 2)
      g *= 2;
    return g;
  }
  // im2col kernel (small maps, the shapes the strip / halo kernel leaves): grouping pays while a layer still has
  // split-K slabs to shed, i.e. up to the split count of the one-layer launch
  static const bool off = getenv("MSML_NO_FAST_WGRAD_GROUP") != nullptr;
  if (off || R != 3 || S != 3 || stride != 1 || up % 8 || vp % 8) return 1;
  if (msml_wgrad_n32_splits(up, vp, N, H, W, P, Q, R, S, stride, pad_h, pad_w) > 0) return 1;
  const int ba = pick_tile(up), bb = pick_tile(vp), taps = R * S, ntw = wgrad_ntw(vp, taps);
  const int tiles = ntw > 1 ? cdiv(up, ba) * cdiv(taps, ntw) * cdiv(vp, 64) : cdiv(up, ba) * cdiv(vp, bb) * taps;
  const int one = pick_splits((long)N * P * Q, tiles);
  int g = 1;
  while (g * 2 <= WR_MAXGROUP && g * 2 <= one) g *= 2;
  return g;
}

// Weight gradients of `group` layers of ONE shape in one launch pair (strip / halo kernel, or the im2col kernel for
// the shapes it serves, + fixed-order reduce): every layer gets 1 / group of the workgroups, so the split-K slab bytes
// written and re-read PER LAYER fall by the same factor (256 -> 256 @ 14x14: 75 MB of slabs for a 2.4 MB gradient at
// group 1; 512 -> 512 @ 7x7 at group 4: no slabs at all, the tiles go straight into dW).  u / v / dw: host arrays of
// `group` device pointers.  Results depend on `group` only through the split partition (fixed for a given group).
extern "C" int msml_conv_wgrad_group(const void* const* u, const void* const* v, float* const* dw, int group, int up,
                                     int vp, int A, int Breal, int Btot, int boff, int N, int H, int W, int P, int Q,
                                     int R, int S, int stride, int pad_h, int pad_w, int accumulate, void* workspace,
                                     long ws_bytes, int dtype, void* stream) {
  MSML_CHECK(u && v && dw && workspace && group >= 1 && group <= WR_MAXGROUP, MSML_ERR_SHAPE,
             "conv_wgrad_group: bad arguments (group %d)", group);
  for (int i = 0; i < group; i++)
    MSML_CHECK(u[i] && v[i] && dw[i], MSML_ERR_SHAPE, "conv_wgrad_group: null pointer in layer %d", i);
  MSML_CHECK(dtype == MSML_BF16, MSML_ERR_UNSUPPORTED, "conv_wgrad_group: bf16 only");
  MSML_CHECK(boff >= 0 && boff + Breal <= Btot && A <= up && Breal <= vp, MSML_ERR_SHAPE, "conv_wgrad_group: bad channel range");
  hipStream_t st = (hipStream_t)stream;
  const int per = msml_wgrad_halo_group_splits(up, vp, A, Breal, N, H, W, P, Q, R, S, stride, pad_h, pad_w, group);
  if (per > 0) {
    const long need = (long)group * per * up * 9 * vp * (long)sizeof(float);
    MSML_CHECK(ws_bytes >= need, MSML_ERR_WORKSPACE, "conv_wgrad_group: workspace %ld < %ld bytes", ws_bytes, need);
    msml_wgrad_halo_launch_group(u, up, v, vp, (float*)workspace, N, H, W, group, per, st, nullptr);
    MSML_LAUNCH_OK("conv_wgrad_group(halo)");
    wgrad_reduce_launch_group((const float*)workspace, dw, group, per, up, 9, vp, A, Breal, Btot, boff, accumulate, st);
    MSML_LAUNCH_OK("conv_wgrad_group_reduce");
    return MSML_OK;
  }
  MSML_CHECK(R == 3 && S == 3 && stride == 1 && (long)N * P * Q < (1L << 24), MSML_ERR_UNSUPPORTED,
             "conv_wgrad_group: 3x3 / stride-1 layers only");
  const int ba = pick_tile(up), bb = pick_tile(vp), taps = R * S, ntw = wgrad_ntw(vp, taps);
  const int tiles = ntw > 1 ? cdiv(up, ba) * cdiv(taps, ntw) * cdiv(vp, 64) : cdiv(up, ba) * cdiv(vp, bb) * taps;
  const long mpix = (long)N * P * Q;
  const int splits = fast_group_splits(mpix, tiles, group);
  const int chunk = (int)(((mpix + splits - 1) / splits + 63) / 64 * 64);
  const long need = splits == 1 ? 0 : (long)group * splits * up * taps * vp * (long)sizeof(float);
  MSML_CHECK(ws_bytes >= need, MSML_ERR_WORKSPACE, "conv_wgrad_group: workspace %ld < %ld bytes", ws_bytes, need);
  MSML_CHECK(msml_wgrad_fast_launch_group(u, up, v, vp, (float*)workspace, N, H, W, P, Q, R, S, stride, pad_h, pad_w, ba,
                                          bb, ntw, group, splits, chunk, st, dw, A, Breal, Btot, boff, accumulate),
             MSML_ERR_UNSUPPORTED, "conv_wgrad_group: tensors too large for 32-bit offsets");
  MSML_LAUNCH_OK("conv_wgrad_group(fast)");
  if (splits > 1) {
    wgrad_reduce_launch_group((const float*)workspace, dw, group, splits, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
    MSML_LAUNCH_OK("conv_wgrad_group_reduce");
  }
  return MSML_OK;
}


// ---------------------------------------------------------------- split-K GEMM (head dX) -----
bool msml_conv_fast_splitk(const void* in0, int c0p, const void* wp, int kop, float* ws, int coutp, int N,
                           int ksplits, hipStream_t st);

// Weight gradient of a conv whose input was X = PReLU(v * x_scale + x_shift) (a training-mode
// BatchNorm in front of the conv that msml_conv2d_bnin applied on the fly): the strips of v are
// normalised in LDS, X never exists in HBM.  Only the shapes of the strip / halo kernel
// (msml_conv_wgrad_bnin_applies); MSML_ERR_UNSUPPORTED otherwise.
extern "C" int msml_conv_wgrad_bnin_applies(int up, int vp, int A, int Breal, int N, int H, int W, int P, int Q,
                                            int R, int S, int stride, int pad_h, int pad_w) {
  if (H == 7 && W == 7) return 0;      // (the image-pair strips of the 7 x 7 maps have no in-LDS BatchNorm variant)
  return msml_wgrad_halo_splits(up, vp, A, Breal, N, H, W, P, Q, R, S, stride, pad_h, pad_w) > 0 ? 1 : 0;
}

extern "C" int msml_conv_wgrad_bnin(const void* u, int up, const void* v, int vp, const float* x_scale,
                                    const float* x_shift, const float* x_alpha, float* dw, int A, int Breal,
                                    int Btot, int boff, int N, int H, int W, int P, int Q, int R, int S,
                                    int stride, int pad_h, int pad_w, int accumulate, void* workspace,
                                    long ws_bytes, void* stream) {
  MSML_CHECK(u && v && dw && workspace && x_scale && x_shift, MSML_ERR_SHAPE, "conv_wgrad_bnin: null pointer");
  MSML_CHECK(up > 0 && up % 8 == 0 && vp > 0 && vp % 8 == 0 && A > 0 && A <= up && Breal > 0 &&
                 Breal <= vp && boff >= 0 && boff + Breal <= Btot,
             MSML_ERR_SHAPE, "conv_wgrad_bnin: bad channels up=%d vp=%d A=%d Breal=%d Btot=%d boff=%d", up,
             vp, A, Breal, Btot, boff);
  MSML_CHECK(N > 0 && H > 0 && W > 0 && P > 0 && Q > 0, MSML_ERR_SHAPE, "conv_wgrad_bnin: bad dims");
  const int hs = msml_wgrad_halo_splits(up, vp, A, Breal, N, H, W, P, Q, R, S, stride, pad_h, pad_w);
  MSML_CHECK(hs > 0 && !(H == 7 && W == 7), MSML_ERR_UNSUPPORTED, "conv_wgrad_bnin: shape not covered by the strip kernel");
  long need = msml_conv_wgrad_workspace(up, vp, N, P, Q, R, S);
  MSML_CHECK(ws_bytes >= need, MSML_ERR_WORKSPACE, "conv_wgrad_bnin: workspace %ld < %ld bytes", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const BnIn xin{x_scale, x_shift, x_alpha};
  MSML_CHECK(msml_wgrad_halo_launch(u, up, v, vp, (float*)workspace, N, H, W, hs, st, &xin), MSML_ERR_UNSUPPORTED,
             "conv_wgrad_bnin: launch refused");
  MSML_LAUNCH_OK("conv_wgrad_bnin(halo)");
  const int taps = R * S;
  wgrad_reduce_launch((const float*)workspace, dw, hs, up, taps, vp, A, Breal, Btot, boff, accumulate, st);
  MSML_LAUNCH_OK("conv_wgrad_reduce");
  return MSML_OK;
}

extern "C" long msml_gemm_splitk_workspace(int M, int coutp, int K) {
  int stages = (K / 32 + 1) / 2;
  int ks = stages / 16 > 0 ? stages / 16 : 1;          // >= 16 stages (1024 k) per slice
  int tiles = cdiv(M, 128) * cdiv(coutp, 128);
  int want = (768 + tiles - 1) / tiles;
  if (ks > want) ks = want;
  return (long)ks * M * coutp * (long)sizeof(float);
}

extern "C" int msml_gemm_splitk(const void* a, int M, int K, const void* wp, int kop, float* out, int coutp,
                                void* workspace, long ws_bytes, int dtype, void* stream) {
  MSML_CHECK(a && wp && out && workspace && M > 0 && K > 0 && K % 32 == 0 && coutp % 8 == 0, MSML_ERR_SHAPE,
             "gemm_splitk: bad args M=%d K=%d coutp=%d", M, K, coutp);
  MSML_CHECK(dtype == MSML_BF16, MSML_ERR_UNSUPPORTED, "gemm_splitk: bf16 only");
  long need = msml_gemm_splitk_workspace(M, coutp, K);
  MSML_CHECK(ws_bytes >= need, MSML_ERR_WORKSPACE, "gemm_splitk: workspace %ld < %ld", ws_bytes, need);
  int ks = (int)(need / ((long)M * coutp * sizeof(float)));
  hipStream_t st = (hipStream_t)stream;
  MSML_CHECK(msml_conv_fast_splitk(a, K, wp, kop, (float*)workspace, coutp, M, ks, st), MSML_ERR_UNSUPPORTED,
             "gemm_splitk: shape not supported by the fast kernel");
  MSML_LAUNCH_OK("gemm_splitk");
  wgrad_reduce_launch((const float*)workspace, out, ks, M, 1, coutp, M, coutp, coutp, 0, 0, st);
  MSML_LAUNCH_OK("gemm_splitk_reduce");
  return MSML_OK;
}
