// Boundary layout kernels: NCHW f32 <-> NHWC (channel-padded) storage, weight packing.
#include "common.h"

// One thread per (n, h, w, 8-channel chunk); reads are strided over C (small C at the
// boundary: 3 input channels / 2 output channels), writes are 16 B (bf16) / 32 B (f32).
template <typename T>
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, T* __restrict__ dst, int N, int C,
                               int HW, int Cp) {
  long total = (long)N * HW * (Cp / 8);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    int c8 = (int)(i % (Cp / 8));
    long pix = i / (Cp / 8);
    int n = (int)(pix / HW);
    int hw = (int)(pix % HW);
    Vec8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int c = c8 * 8 + j;
      v.v[j] = c < C ? src[((long)n * C + c) * HW + hw] : 0.f;
    }
    store8<T>(dst + pix * Cp + c8 * 8, v);
  }
}

// NHWC -> NCHW through an LDS transpose tile: 64 pixels x 8 channels per wave step would
// be overkill for the tiny boundary tensors; lanes run along hw for coalesced f32 writes.
template <typename T>
__global__ void k_nhwc_to_nchw(const T* __restrict__ src, float* __restrict__ dst, int N, int C,
                               int HW, int Cp) {
  long total = (long)N * C * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    int hw = (int)(i % HW);
    long nc = i / HW;
    int c = (int)(nc % C);
    int n = (int)(nc / C);
    dst[i] = load1<T>(src + ((long)n * HW + hw) * Cp + c);
  }
}

extern "C" int msml_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int Cp,
                                 int dtype, void* stream) {
  MSML_CHECK(N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && Cp % 8 == 0, MSML_ERR_SHAPE,
             "nchw_to_nhwc: bad shape N=%d C=%d H=%d W=%d Cp=%d", N, C, H, W, Cp);
  long total = (long)N * H * W * (Cp / 8);
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  MSML_DISPATCH_DTYPE(dtype, "nchw_to_nhwc",
                      k_nchw_to_nhwc<DT><<<grid, 256, 0, (hipStream_t)stream>>>(
                          src, (DT*)dst, N, C, H * W, Cp);)
  MSML_LAUNCH_OK("nchw_to_nhwc");
  return MSML_OK;
}

extern "C" int msml_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int Cp,
                                 int dtype, void* stream) {
  MSML_CHECK(N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && Cp % 8 == 0, MSML_ERR_SHAPE,
             "nhwc_to_nchw: bad shape N=%d C=%d H=%d W=%d Cp=%d", N, C, H, W, Cp);
  long total = (long)N * C * H * W;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  MSML_DISPATCH_DTYPE(dtype, "nhwc_to_nchw",
                      k_nhwc_to_nchw<DT><<<grid, 256, 0, (hipStream_t)stream>>>(
                          (const DT*)src, dst, N, C, H * W, Cp);)
  MSML_LAUNCH_OK("nhwc_to_nchw");
  return MSML_OK;
}

// dst[ko][seg][r][s][c] <- w[a][b][r][s]; one thread per destination element (the packed
// weights are at most ~25 MB per layer and are rewritten once per optimizer step).
template <typename T>
__global__ void k_pack_weight(const float* __restrict__ w, T* __restrict__ dst, int A, int B, int R,
                              int S, int transpose, int C1, int C1p, int C2, int C2p, int KOp,
                              int K0, int Ktot) {
  long total = (long)KOp * Ktot;
  int KO = transpose ? B : A;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    int k = (int)(i % Ktot);
    int ko = (int)(i / Ktot);
    int seg = k >= K0;
    int kk = seg ? k - K0 : k;
    int cp = seg ? C2p : C1p;
    int cn = seg ? C2 : C1;
    int c = kk % cp;
    int tap = kk / cp;
    float v = 0.f;
    if (ko < KO && tap < R * S && c < cn) {
      int r = tap / S, s = tap % S;
      int ci = c + (seg ? C1 : 0);
      int a = transpose ? ci : ko;
      int b = transpose ? ko : ci;
      v = w[(((long)a * B + b) * R + r) * S + s];
    }
    store1<T>(dst + i, v);
  }
}

extern "C" int msml_pack_weight(const float* w, void* dst, int A, int B, int R, int S,
                                int transpose, int C1, int C1p, int C2, int C2p, int KOp,
                                int dtype, void* stream) {
  int CI = transpose ? A : B;
  int KO = transpose ? B : A;
  MSML_CHECK(A > 0 && B > 0 && R > 0 && S > 0 && C1 > 0 && C2 >= 0 && C1 + C2 == CI &&
                 C1p >= C1 && C1p % 8 == 0 && C2p >= C2 && C2p % 8 == 0 && KOp >= KO,
             MSML_ERR_SHAPE, "pack_weight: bad shape A=%d B=%d R=%d S=%d C1=%d/%d C2=%d/%d KOp=%d",
             A, B, R, S, C1, C1p, C2, C2p, KOp);
  int K0 = (R * S * C1p + 31) / 32 * 32;
  int K1 = C2 > 0 ? (R * S * C2p + 31) / 32 * 32 : 0;
  int Ktot = K0 + K1;
  long total = (long)KOp * Ktot;
  int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  MSML_DISPATCH_DTYPE(dtype, "pack_weight",
                      k_pack_weight<DT><<<grid, 256, 0, (hipStream_t)stream>>>(
                          w, (DT*)dst, A, B, R, S, transpose, C1, C1p, C2, C2p, KOp, K0, Ktot);)
  MSML_LAUNCH_OK("pack_weight");
  return MSML_OK;
}


// ---------------------------------------------------------------- batched / sub-block packing -
// One launch packs every weight of a model: desc[i] = 16 x int64
//   {w, dst, Afull, Bfull, a_off, A, b_off, B, R, S, transpose, C1, C1p, C2, C2p, KOp}
// (w, dst are device pointers).  The sub-block (a_off, A) x (b_off, B) of w[Afull][Bfull][R][S]
// is packed exactly like msml_pack_weight packs a whole tensor, so backward-data packs of one
// concat segment need no sliced copy of the parameter.
template <typename T>
__global__ void __launch_bounds__(256) k_pack_batched(const long* __restrict__ table, int count) {
  const long* d = table + (long)blockIdx.y * 16;
  const float* w = reinterpret_cast<const float*>(d[0]);
  T* dst = reinterpret_cast<T*>(d[1]);
  const int Bfull = (int)d[3], a_off = (int)d[4], A = (int)d[5], b_off = (int)d[6], B = (int)d[7];
  const int R = (int)d[8], S = (int)d[9], transpose = (int)d[10];
  const int C1 = (int)d[11], C1p = (int)d[12], C2 = (int)d[13], C2p = (int)d[14], KOp = (int)d[15];
  const int K0 = (R * S * C1p + 31) / 32 * 32;
  const int K1 = C2 > 0 ? (R * S * C2p + 31) / 32 * 32 : 0;
  const int Ktot = K0 + K1;
  const long total = (long)KOp * Ktot;
  const int KO = transpose ? B : A;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int k = (int)(i % Ktot);
    int ko = (int)(i / Ktot);
    int seg = k >= K0;
    int kk = seg ? k - K0 : k;
    int cp = seg ? C2p : C1p;
    int cn = seg ? C2 : C1;
    int c = kk % cp;
    int tap = kk / cp;
    float v = 0.f;
    if (ko < KO && tap < R * S && c < cn) {
      int r = tap / S, s2 = tap % S;
      int ci = c + (seg ? C1 : 0);
      int a = (transpose ? ci : ko) + a_off;
      int b = (transpose ? ko : ci) + b_off;
      v = w[(((long)a * Bfull + b) * R + r) * S + s2];
    }
    store1<T>(dst + i, v);
  }
}

extern "C" int msml_pack_weights_batched(const long* table, int count, int dtype, void* stream) {
  MSML_CHECK(table && count > 0, MSML_ERR_SHAPE, "pack_weights_batched: bad args");
  dim3 grid(192, count);
  MSML_DISPATCH_DTYPE(dtype, "pack_weights_batched",
                      k_pack_batched<DT><<<grid, 256, 0, (hipStream_t)stream>>>(table, count);)
  MSML_LAUNCH_OK("pack_weights_batched");
  return MSML_OK;
}


// Tiled variant for the per-step refresh: a workgroup moves a 32 x BT block of (a, b) index pairs
// with all R*S taps through LDS, so the f32 parameter is read in contiguous BT*R*S-float runs and
// the packed operand is written in contiguous channel runs (the element-wise kernel above reads
// with a stride of R*S floats and spends its time on integer division: 0.75 ms per step for the
// 55 M parameters of ires50-MSML).  Writes only real elements: the zero padding of dst (channel,
// K and row padding) must already be there (it never changes).
static inline int pack_bt(int RS) { return RS <= 9 ? 32 : (RS <= 16 ? 16 : 8); }
extern "C" int msml_pack_tiles(int A, int B, int R, int S) { return cdiv(A, 32) * cdiv(B, pack_bt(R * S)); }

// RSC > 0: compile-time R*S (the index decodes below divide by it; with a run-time divisor the
// kernel spent most of its 0.41 ms on integer division)
template <typename T, int RSC>
__device__ __forceinline__ void pack_tile_body(const long* __restrict__ d, int lt, float* tile) {
  const float* w = reinterpret_cast<const float*>(d[0]);
  T* dst = reinterpret_cast<T*>(d[1]);
  const int Bfull = (int)d[3], a_off = (int)d[4], A = (int)d[5], b_off = (int)d[6], B = (int)d[7];
  const int transpose = (int)d[10];
  const int C1 = (int)d[11], C1p = (int)d[12], C2 = (int)d[13], C2p = (int)d[14];
  const int RS = RSC > 0 ? RSC : (int)d[8] * (int)d[9], BT = RS <= 9 ? 32 : (RS <= 16 ? 16 : 8);
  const int K0 = (RS * C1p + 31) / 32 * 32;
  const int K1 = C2 > 0 ? (RS * C2p + 31) / 32 * 32 : 0;
  const int Ktot = K0 + K1;
  const int tiles_b = (B + BT - 1) / BT;
  MSML_LDS_REGION(tile, 32 * (BT * RS + 1) * 4);
  const int a0 = (lt / tiles_b) * 32, b0 = (lt % tiles_b) * BT;
  const int na = A - a0 < 32 ? A - a0 : 32, nb = B - b0 < BT ? B - b0 : BT;
  const int run = nb * RS, pitch = BT * RS + 1;
  // rows of the parameter are contiguous runs of nb * RS floats; a wave takes rows wave, wave + 4, ...
  // With a compile-time R*S all loads of the block are issued before the first LDS store (the
  // rolled loop waited for every load in turn: 36 dependent round trips per block)
  if (RSC > 0) {
    constexpr int MAXRUN = (RSC <= 9 ? 32 : (RSC <= 16 ? 16 : 8)) * (RSC > 0 ? RSC : 1), PER = (MAXRUN + 63) / 64;
    float v[8][PER];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int a = wv + 4 * i;
      const float* src = w + ((long)(a_off + a0 + (a < na ? a : 0)) * Bfull + b_off + b0) * RS;
#pragma unroll
      for (int k = 0; k < PER; k++) {
        const int j = lane + 64 * k;
        v[i][k] = (a < na && j < run) ? src[j] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int a = wv + 4 * i;
#pragma unroll
      for (int k = 0; k < PER; k++) {
        const int j = lane + 64 * k;
        if (a < na && j < run) tile[a * pitch + j] = v[i][k];
      }
    }
  } else {
    for (int a = threadIdx.x >> 6; a < na; a += 4) {
      const float* src = w + ((long)(a_off + a0 + a) * Bfull + b_off + b0) * RS;
      for (int j = threadIdx.x & 63; j < run; j += 64) tile[a * pitch + j] = src[j];
    }
  }
  __syncthreads();
  // the contiguous channel index of dst is b (forward packs) or a (transposed packs): make it the
  // fastest thread index
  const int nc = transpose ? na : nb, no = transpose ? nb : na;
  const int cbase = transpose ? a0 : b0;               // first channel (inside the concatenated input)
  if (nc % 8 == 0 && C1 % 8 == 0 && cbase % 8 == 0) {
    // 8 consecutive channels per thread: one 16-B (bf16) store
    const int n8 = nc / 8;
    for (int i = threadIdx.x; i < no * RS * n8; i += 256) {
      const int c8 = i % n8, rest = i / n8;
      const int oidx = rest / RS, tap = rest - oidx * RS;
      const int ko = transpose ? b0 + oidx : a0 + oidx;
      const int ci = cbase + c8 * 8;
      const int seg = ci >= C1;
      const int c = seg ? ci - C1 : ci;
      const int cp = seg ? C2p : C1p;
      Vec8 v;
#pragma unroll
      for (int j = 0; j < 8; j++)
        v.v[j] = transpose ? tile[(c8 * 8 + j) * pitch + oidx * RS + tap] : tile[oidx * pitch + (c8 * 8 + j) * RS + tap];
      store8<T>(dst + (long)ko * Ktot + (seg ? K0 : 0) + tap * cp + c, v);
    }
    return;
  }
  for (int i = threadIdx.x; i < no * RS * nc; i += 256) {
    const int cidx = i % nc, rest = i / nc;
    const int oidx = rest / RS, tap = rest - oidx * RS;
    const int a = transpose ? cidx : oidx, bb = transpose ? oidx : cidx;
    const int ko = transpose ? b0 + bb : a0 + a;       // packed row
    const int ci = transpose ? a0 + a : b0 + bb;       // channel inside the concatenated input
    const int seg = ci >= C1;
    const int c = seg ? ci - C1 : ci;
    const int cp = seg ? C2p : C1p;
    store1<T>(dst + (long)ko * Ktot + (seg ? K0 : 0) + tap * cp + c, tile[a * pitch + bb * RS + tap]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_pack_tiled(const long* __restrict__ table, const int* __restrict__ prefix,
                                                    int count) {
  extern __shared__ float tile[];                      // [32][BT * RS + 1]
  // entry of this block: last e with prefix[e] <= blockIdx.x.  prefix[count + t] (when the caller
  // appended it) is that entry for tile t directly -- the binary search is a chain of ~8 dependent
  // loads in front of a 10 us block
  int lo = 0, hi = count - 1;
  if (prefix[count] == -1) {
    lo = prefix[count + 1 + blockIdx.x];
  } else {
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
  }
  const long* d = table + (long)lo * 16;
  const int RS = (int)d[8] * (int)d[9], lt = blockIdx.x - prefix[lo];
  if (RS == 9) pack_tile_body<T, 9>(d, lt, tile);
  else if (RS == 1) pack_tile_body<T, 1>(d, lt, tile);
  else if (RS == 49) pack_tile_body<T, 49>(d, lt, tile);
  else pack_tile_body<T, 0>(d, lt, tile);
}

extern "C" int msml_pack_weights_tiled(const long* table, const int* tile_prefix, int count, int total_tiles,
                                       int dtype, void* stream) {
  MSML_CHECK(table && tile_prefix && count > 0 && total_tiles > 0, MSML_ERR_SHAPE, "pack_weights_tiled: bad args");
  const size_t lds = (size_t)32 * (32 * 9 + 1 > 8 * 49 + 1 ? 32 * 9 + 1 : 8 * 49 + 1) * sizeof(float);
  MSML_DISPATCH_DTYPE(dtype, "pack_weights_tiled",
                      k_pack_tiled<DT><<<total_tiles, 256, lds, (hipStream_t)stream>>>(table, tile_prefix, count);)
  MSML_LAUNCH_OK("pack_weights_tiled");
  return MSML_OK;
}


// ---------------------------------------------------------------- stem im2col -----------------
// The stems convolve the 3-channel image (iresnet.py:209 `conv1`, unet.py:193): as an NHWC conv
// the 3 channels pad to 32 and the 3x3 kernel walks K = 288, 27 of which are real.  Instead the
// image is unfolded once: out[n][oy][ox][k] = x[n][c][oy*stride - pad + r][ox*stride - pad + s]
// with k = (r*S + s)*C + c < R*S*C <= KP (zeros above), and the stem becomes a 1x1 conv over
// KP = 32 channels (9x fewer MFMAs and operand bytes; its weight gradient likewise).
// CC / RR / SS > 0: compile-time channel and kernel sizes (the k -> (tap, c) decode is integer
// division: with run-time divisors the kernel was VALU-bound at 1.3 TB/s)
template <typename T, int CC, int RR, int SS>
__global__ void __launch_bounds__(256) k_stem_im2col(const float* __restrict__ x, T* __restrict__ out, int N, int C_,
                                                     int H, int W, int P, int Q, int R_, int S_, int stride,
                                                     int pad, int KP) {
  const int C = CC > 0 ? CC : C_, R = RR > 0 ? RR : R_, S = SS > 0 ? SS : S_;
  // one thread per (pixel, 8-channel chunk): consecutive threads store consecutive 16 B
  const int K8 = KP / 8;
  const long total = (long)N * P * Q * K8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k0 = (int)(i % K8) * 8;
    const long pix = i / K8;
    const int ox = (int)(pix % Q);
    const long t = pix / Q;
    const int oy = (int)(t % P), n = (int)(t / P);
    const float* xn = x + (long)n * C * H * W;
    Vec8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = k0 + j, tap = k / C, c = k - tap * C;
      const int r = tap / S, s2 = tap - r * S;
      const int iy = oy * stride - pad + r, ix = ox * stride - pad + s2;
      const bool ok = (tap < R * S) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
      v.v[j] = ok ? xn[((long)c * H + iy) * W + ix] : 0.f;
    }
    store8<T>(out + pix * KP + k0, v);
  }
}

// The stems' own case (3 channels, 3x3, pad 1, KP = 32, W <= 128) through LDS (round 6): the gather above issues eight scattered
// 4-B loads per 16-B store and ran at 1.9-2.1 TB/s (0.13 ms for the 243 MB of the FRB stem at batch 256, a quarter of which is
// algorithmic).  Here a workgroup owns ROWS output rows of one image: the 3 x (ROWS * STRIDE + 2) input rows are loaded once,
// coalesced, into an LDS tile with a zero halo (every input element crosses the memory system once per workgroup instead of
// nine times), and every thread assembles (pixel, 8-k chunk) units from LDS -- consecutive threads store consecutive 16 B.
template <typename T, int STRIDE>
__global__ void __launch_bounds__(256) k_stem_im2col_lds(const float* __restrict__ x, T* __restrict__ out, int N, int H, int W,
                                                         int P, int Q) {
  constexpr int ROWS = 4, NR = (ROWS - 1) * STRIDE + 3, WMAX = 128;
  __shared__ float tile[3][NR][WMAX + 2];
  const int t = threadIdx.x;
  const int tiles_y = (P + ROWS - 1) / ROWS;
  const int n = blockIdx.x / tiles_y, oy0 = (blockIdx.x % tiles_y) * ROWS;
  const int iy0 = oy0 * STRIDE - 1;
  const float* xn = x + (long)n * 3 * H * W;
  for (int i = t; i < 3 * NR * W; i += 256) {
    const int c = i / (NR * W), rem = i - c * (NR * W), row = rem / W, ix = rem - row * W;
    const int iy = iy0 + row;
    tile[c][row][ix + 1] = ((unsigned)iy < (unsigned)H) ? xn[((long)c * H + iy) * W + ix] : 0.f;
  }
  for (int i = t; i < 3 * NR * 2; i += 256) {
    const int c = i / (NR * 2), rem = i - c * (NR * 2), row = rem >> 1;
    tile[c][row][(rem & 1) ? W + 1 : 0] = 0.f;
  }
  __syncthreads();
  const int units = ROWS * Q * 4;
  for (int i = t; i < units; i += 256) {
    const int k0 = (i & 3) * 8, pix = i >> 2;
    const int orow = pix / Q, ox = pix - orow * Q;
    const int oy = oy0 + orow;
    if (oy >= P) break;
    Vec8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = k0 + j, tap = k / 3, c = k - tap * 3;
      const int r = tap / 3, s2 = tap - r * 3;
      v.v[j] = tap < 9 ? tile[c][orow * STRIDE + r][ox * STRIDE + s2] : 0.f;
    }
    store8<T>(out + (((long)n * P + oy) * Q + ox) * 32 + k0, v);
  }
}

extern "C" int msml_stem_im2col(const float* x, void* out, int N, int C, int H, int W, int P, int Q, int R,
                                int S, int stride, int pad, int KP, int dtype, void* stream) {
  MSML_CHECK(x && out && N > 0 && C > 0 && H > 0 && W > 0 && P > 0 && Q > 0 && R > 0 && S > 0 && stride > 0,
             MSML_ERR_SHAPE, "stem_im2col: bad dims");
  MSML_CHECK(KP % 8 == 0 && R * S * C <= KP, MSML_ERR_SHAPE, "stem_im2col: R*S*C = %d does not fit KP = %d",
             R * S * C, KP);
  const long total = (long)N * P * Q * (KP / 8);
  const int grid = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  const bool no_lds = getenv("MSML_NO_STEM_LDS") != nullptr;      // A/B switch, read per call (the test compares both kernels)
  if (!no_lds && C == 3 && R == 3 && S == 3 && pad == 1 && KP == 32 && W <= 128 && (stride == 1 || stride == 2) &&
      P == (H + 2 - 3) / stride + 1 && Q == (W + 2 - 3) / stride + 1) {
    const int lgrid = N * ((P + 3) / 4);
    if (stride == 1) {
      MSML_DISPATCH_DTYPE(dtype, "stem_im2col",
                          (k_stem_im2col_lds<DT, 1>)<<<lgrid, 256, 0, (hipStream_t)stream>>>(x, (DT*)out, N, H, W, P, Q);)
    } else {
      MSML_DISPATCH_DTYPE(dtype, "stem_im2col",
                          (k_stem_im2col_lds<DT, 2>)<<<lgrid, 256, 0, (hipStream_t)stream>>>(x, (DT*)out, N, H, W, P, Q);)
    }
    MSML_LAUNCH_OK("stem_im2col");
    return MSML_OK;
  }
  if (C == 3 && R == 3 && S == 3) {
    MSML_DISPATCH_DTYPE(dtype, "stem_im2col",
                        (k_stem_im2col<DT, 3, 3, 3>)<<<grid, 256, 0, (hipStream_t)stream>>>(x, (DT*)out, N, C, H, W, P,
                                                                                            Q, R, S, stride, pad, KP);)
  } else {
    MSML_DISPATCH_DTYPE(dtype, "stem_im2col",
                        (k_stem_im2col<DT, 0, 0, 0>)<<<grid, 256, 0, (hipStream_t)stream>>>(x, (DT*)out, N, C, H, W, P,
                                                                                            Q, R, S, stride, pad, KP);)
  }
  MSML_LAUNCH_OK("stem_im2col");
  return MSML_OK;
}


// ---------------------------------------------------------------- 2-D transpose ---------------
// dst[c][r] = src[r][c] for r < R, c < C; dst rows are ld_d long and zero-filled for r in
// [R, ld_d).  64 x 64 tiles through LDS (coalesced both ways).  Used for Wn^T of the PartialFC
// dX GEMM (K = classes must be the contiguous dimension of the MFMA operand).
template <typename T>
__global__ void __launch_bounds__(256) k_transpose(const T* __restrict__ src, int R, int C, int ld_s,
                                                   T* __restrict__ dst, int ld_d) {
  __shared__ T tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? src[(long)r * ld_s + c] : (T)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    int c = c0 + i, r = r0 + tx;
    if (c < C && r < ld_d) dst[(long)c * ld_d + r] = tile[tx][i];
  }
}

extern "C" int msml_transpose(const void* src, int R, int C, int ld_s, void* dst, int ld_d, int dtype,
                              void* stream) {
  MSML_CHECK(src && dst && R > 0 && C > 0 && ld_s >= C && ld_d >= R, MSML_ERR_SHAPE, "transpose: bad args");
  dim3 grid(cdiv(ld_d, 64), cdiv(C, 64));
  MSML_DISPATCH_DTYPE(dtype, "transpose",
                      k_transpose<DT><<<grid, 256, 0, (hipStream_t)stream>>>((const DT*)src, R, C, ld_s, (DT*)dst,
                                                                          ld_d);)
  MSML_LAUNCH_OK("transpose");
  return MSML_OK;
}
