// Fast path of the implicit-GEMM convolution for bf16 tensors whose input segments all have
// Cp % 32 == 0 (every 3x3 / 1x1 conv of the FRB, OSB encoder and FM operators except the
// 3-channel stems and the 18-channel seg segments).  Same math and ABI as conv_igemm.hip
// (k_conv_igemm is the general kernel and the f32 parity path); differences:
//   * K-stage of 64 (two 32-channel sub-steps) per barrier: 16 MFMAs per wave per barrier,
//   * a sub-step is (tap, 32-channel block), so a thread's gather address is
//     pixel_base(tap) + const: no per-chunk div/mod and no tap crossing inside a chunk row,
//   * XCD-aware tile order (neighbouring pixel tiles share halo rows in one XCD's L2),
//   * tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 16 B per lane): no VGPR staging
//     and no ds_write pass (the LDS store path, ~79 B/clk/CU, was the co-bottleneck of the
//     register-staged version).  An LDS-DMA wave-instruction writes 64 x 16 B linearly, so
//     the XOR swizzle is applied to the per-lane SOURCE chunk and again on the fragment read
//     (cdna_hip_programming.md rule 21); out-of-image taps read a zero page,
//   * bf16 results are transposed through LDS and stored as 16-B rows (not 2-B scalars).
#include "common.h"

__device__ unsigned int g_zero_page[64];     // 256 B of zeros: source of padded / dead lanes

struct ConvFastArgs {
  const unsigned short* in[2];
  int cp[2];          // channels per segment (multiple of 32)
  int nsub[2];        // sub-steps per segment = R*S*cp/32
  int nseg;
  int N, H, W, P, Q;
  int R, S, stride_shift, stride, pad_h, pad_w, transposed;
  const unsigned short* wp;
  int Ktot;
  void* out;
  int coutp;
  const float* bias;
  float* stats;
  long M;
  int tiles_m;
};

__device__ __forceinline__ int swz128(int row) { return (row >> 1) & 7; }   // 128-B rows

template <typename TOUT, int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(256) k_conv_fast(const ConvFastArgs p) {
  constexpr int NA = BM / 32, NB = BN >= 32 ? BN / 32 : 1;     // rows per thread (8 chunks/row)
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32;
  static_assert(WGM * WGN == 4 && TM >= 1 && TN >= 1, "tile config");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* As = reinterpret_cast<u32x4*>(smem);          // [2][BM][8]
  u32x4* Bs = As + 2 * BM * 8;                         // [2][BN][8]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = t >> 3;                             // 32 rows per pass, lane l -> LDS slot l
  const int chunk = (t & 7) ^ swz128(row0);            // logical 16-B chunk this lane fetches
  const int sub = chunk >> 2, cc = chunk & 3;

  // XCD-aware remap of the pixel-tile index (bijective for any tile count)
  int bid = blockIdx.x;
  {
    const int nt = p.tiles_m, q = nt >> 3, r8 = nt & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  }
  const long m0 = (long)bid * BM;
  const int n0 = blockIdx.y * BN;

  int pb[NA], y0[NA], x0[NA];
  const int PQ = p.P * p.Q;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    long m = m0 + row0 + i * 32;
    bool ok = m < p.M;
    int mm = ok ? (int)m : 0;
    int n = mm / PQ;
    int rem = mm - n * PQ;
    int oy = rem / p.Q, ox = rem - oy * p.Q;
    pb[i] = n * p.H * p.W;
    if (p.transposed) {
      y0[i] = oy + p.pad_h;
      x0[i] = ox + p.pad_w;
    } else {
      y0[i] = oy * p.stride - p.pad_h;
      x0[i] = ox * p.stride - p.pad_w;
    }
    if (!ok) y0[i] = -(1 << 20);                       // forces every tap out of range
  }
  const unsigned short* wrow = p.wp + (long)(n0 + row0) * p.Ktot + cc * 8;

  // this thread's sub-step stream: q = sub, sub + 2, ...
  int seg = 0, r = 0, s = 0, c32 = sub, qglob = sub;
  int nc32 = p.cp[0] >> 5;
  int seg_left = p.nsub[0];                            // sub-steps left in the current segment
  bool alive = true;
  const unsigned short* inp = p.in[0];
  int cpseg = p.cp[0];
  auto settle = [&]() {                                // normalise (c32, s, r, seg) after a jump
    while (alive) {
      while (c32 >= nc32 && r < p.R) {
        c32 -= nc32;
        if (++s == p.S) { s = 0; ++r; }
      }
      if (r < p.R) break;
      if (seg + 1 < p.nseg) {                          // carry the overshoot into the next segment
        seg++;
        r = 0; s = 0;
        nc32 = p.cp[seg] >> 5;
        cpseg = p.cp[seg];
        inp = p.in[seg];
      } else {
        alive = false;
      }
    }
  };
  (void)seg_left;
  settle();
  const int qtot = p.nsub[0] + (p.nseg > 1 ? p.nsub[1] : 0);
  const int stages = (qtot + 1) >> 1;

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero_page);
  // issue the LDS-DMA loads of the current sub-step state into stage buffer `buf`
  auto gissue = [&](int buf) {
    char* a = reinterpret_cast<char*>(As + buf * BM * 8) + wave * 1024;
    char* b = reinterpret_cast<char*>(Bs + buf * BN * 8) + wave * 1024;
    // all source addresses first (distinct registers), then the DMA issues back to back:
    // hipcc waits vmcnt(0) before it overwrites the address VGPRs of an in-flight LDS-DMA
    const unsigned short* sa[NA];
    const unsigned short* sb[NB];
    const long coff = (long)(c32 << 5) + cc * 8;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      int iy, ix;
      bool ok = alive;
      if (p.transposed) {
        int ty = y0[i] - r, tx = x0[i] - s;
        ok = ok & (ty >= 0) & (tx >= 0) & (((ty | tx) & (p.stride - 1)) == 0);
        iy = ty >> p.stride_shift;
        ix = tx >> p.stride_shift;
        ok = ok & (iy < p.H) & (ix < p.W);
      } else {
        iy = y0[i] + r;
        ix = x0[i] + s;
        ok = ok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      }
      const unsigned short* src = inp + ((long)(pb[i] + iy * p.W + ix) * cpseg + coff);
      sa[i] = ok ? src : zero;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const unsigned short* src = wrow + (long)i * 32 * p.Ktot + (long)qglob * 32;
      sb[i] = alive ? src : zero;
    }
#pragma unroll
    for (int i = 0; i < NA; i++)
      __builtin_amdgcn_global_load_lds((gptr_t)sa[i], (lptr_t)(a + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; i++)
      __builtin_amdgcn_global_load_lds((gptr_t)sb[i], (lptr_t)(b + i * 4096), 16, 0, 0);
  };
  auto advance = [&]() {
    c32 += 2;
    qglob += 2;
    if (qglob >= qtot) alive = false;
    settle();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int wm = wave / WGN, wn = wave % WGN;
  const int arow0 = wm * WTM, brow0 = wn * WTN;
  const int r32 = lane & 31, h = lane >> 5;

  if (qglob >= qtot) alive = false;
  gissue(0);
  __syncthreads();                                     // (hipcc drains vmcnt before the barrier)
  int cur = 0;
  for (int st = 0; st < stages; st++) {
    const bool more = st + 1 < stages;
    if (more) {
      advance();
      gissue(cur ^ 1);
    }
    const u32x4* A = As + cur * BM * 8;
    const u32x4* B = Bs + cur * BN * 8;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {                   // 4 x 16 k per 64-wide stage
      u32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) {
        int row = arow0 + i * 32 + r32;
        a[i] = A[row * 8 + ((kk * 2 + h) ^ swz128(row))];
      }
#pragma unroll
      for (int j = 0; j < TN; j++) {
        int row = brow0 + j * 32 + r32;
        b[j] = B[row * 8 + ((kk * 2 + h) ^ swz128(row))];
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---------------- epilogue (same contract as k_conv_igemm) -----------------------------------
  // bf16 results go through an LDS transpose tile [BM][BN + 8] and leave as 16-B row chunks;
  // f32 results (head logits) are stored directly.
  TOUT* outp = reinterpret_cast<TOUT*>(p.out);
  constexpr bool VIA_LDS = sizeof(TOUT) == 2;
  constexpr int OP = BN + 8;                           // tile pitch in elements
  unsigned short* otile = reinterpret_cast<unsigned short*>(smem);
  float s1v[TN], s2v[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int lcol = brow0 + j * 32 + r32;
    const int col = n0 + lcol;
    const bool cok = col < p.coutp;
    const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int row = arow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        long m = m0 + row;
        float v = acc[i][j][e] + bv;
        if (VIA_LDS) otile[row * OP + lcol] = f2bf(v);
        if (m < p.M && cok) {
          if (!VIA_LDS) store1<TOUT>(outp + m * p.coutp + col, v);
          s1 += v;
          s2 += v * v;
        }
      }
    }
    s1v[j] = s1;
    s2v[j] = s2;
  }
  if (VIA_LDS) {
    __syncthreads();
    constexpr int C8 = BN / 8;
    for (int idx = t; idx < BM * C8; idx += 256) {
      int row = idx / C8, c8 = idx % C8;
      long m = m0 + row;
      int col = n0 + c8 * 8;
      if (m < p.M && col < p.coutp) {
        u32x4 v = *reinterpret_cast<const u32x4*>(otile + row * OP + c8 * 8);
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.out) + m * p.coutp + col) = v;
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TN; j++) {
      float s1 = s1v[j] + __shfl_xor(s1v[j], 32, 64);
      float s2 = s2v[j] + __shfl_xor(s2v[j], 32, 64);
      if (h == 0) {
        red[(wm * 2 + 0) * BN + brow0 + j * 32 + r32] = s1;
        red[(wm * 2 + 1) * BN + brow0 + j * 32 + r32] = s2;
      }
    }
    __syncthreads();
    for (int i = t; i < 2 * BN; i += 256) {
      int which = i / BN, c = i % BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; w++) v += red[(w * 2 + which) * BN + c];
      int col = n0 + c;
      if (col < p.coutp) p.stats[((long)bid * 2 + which) * p.coutp + col] = v;
    }
  }
}

template <typename TOUT, int BM, int BN, int WGM, int WGN>
static void launch_fast(ConvFastArgs& a, hipStream_t st) {
  a.tiles_m = cdiv(a.M, BM);
  dim3 grid(a.tiles_m, cdiv(a.coutp, BN));
  size_t lds = (size_t)2 * (BM + BN) * 8 * 16;
  k_conv_fast<TOUT, BM, BN, WGM, WGN><<<grid, dim3(256), lds, st>>>(a);
}

// Called by msml_conv2d (conv_igemm.hip) when the fast-path conditions hold.  Returns false if
// this kernel does not apply.
bool msml_conv_fast_dispatch(const void* in0, int c0p, const void* in1, int c1p, const void* wp,
                             const float* bias, void* out, int coutp, float* stats, int N, int H,
                             int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                             int transposed, int in_dtype, int out_dtype, int bm, int bn,
                             hipStream_t st) {
  if (in_dtype != MSML_BF16) return false;
  if (c0p % 32 != 0 || (in1 && c1p % 32 != 0)) return false;
  ConvFastArgs a;
  a.in[0] = (const unsigned short*)in0;
  a.in[1] = (const unsigned short*)in1;
  a.cp[0] = c0p; a.cp[1] = c1p;
  a.nseg = in1 ? 2 : 1;
  a.nsub[0] = R * S * (c0p / 32);
  a.nsub[1] = in1 ? R * S * (c1p / 32) : 0;
  a.Ktot = 32 * (a.nsub[0] + a.nsub[1]);
  a.N = N; a.H = H; a.W = W; a.P = P; a.Q = Q; a.R = R; a.S = S;
  a.stride = stride;
  a.stride_shift = stride == 1 ? 0 : (stride == 2 ? 1 : 2);
  a.pad_h = pad_h; a.pad_w = pad_w; a.transposed = transposed;
  a.wp = (const unsigned short*)wp; a.out = out; a.coutp = coutp; a.bias = bias; a.stats = stats;
  a.M = (long)N * P * Q;
#define FAST_CASE(TO)                                          \
  if (bn == 128) launch_fast<TO, 128, 128, 2, 2>(a, st);       \
  else if (bn == 64) launch_fast<TO, 256, 64, 4, 1>(a, st);    \
  else launch_fast<TO, 256, 32, 4, 1>(a, st);
  if (out_dtype == MSML_BF16) { FAST_CASE(unsigned short) }
  else { FAST_CASE(float) }
#undef FAST_CASE
  (void)bm;
  return true;
}
