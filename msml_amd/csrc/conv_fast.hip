// Fast path of the implicit-GEMM convolution for bf16 tensors whose input segments all have
// Cp % 32 == 0 (every conv of the FRB, OSB and FM operators except the 3-channel stems and
// gcm1's 8-channel maps).  Same math and ABI as conv_igemm.hip (k_conv_igemm stays the general
// kernel and the exact-f32 parity path).  What makes it fast on gfx950:
//   * tiles go HBM/L2 -> LDS directly by LDS-DMA (`buffer_load_dwordx4 ... lds`, 16 B per lane):
//     no VGPR staging and no ds_write pass (the LDS store path, ~79 B/clk/CU, was the
//     co-bottleneck of the register-staged version).  A DMA wave-instruction writes 64 x 16 B
//     linearly, so the XOR swizzle is applied to the per-lane SOURCE chunk and again on the
//     ds_read_b128 fragment read (cdna_hip_programming.md rule 21),
//   * buffer addressing: 32-bit offsets against a wave-uniform descriptor, and an out-of-range
//     offset makes the DMA write ZEROS (verified on MI355X) -- conv padding, the zero-dilated
//     taps of a transposed gather and dead lanes cost one v_cndmask, no zero page, no 64-bit math,
//   * a K stage is 64 = two (tap, 32-channel) sub-steps, so a lane's gather offset is
//     (row_pixel + tap_delta) * Cp + const,
//   * fragment addresses are four per-lane LDS offsets + immediates (PMC: the first LDS-DMA
//     version issued 9.4 VALU per MFMA, mostly address arithmetic),
//   * XCD-aware tile order (neighbouring pixel tiles share halo rows in one XCD's L2),
//   * bf16 results are transposed through LDS and stored as 16-B row chunks.
#include <stdlib.h>

#include <mutex>

#include "common.h"

struct ConvFastArgs {
  const unsigned short* in[2];
  unsigned int in_bytes[2];
  int cp[2];          // channels per segment (multiple of 32)
  int nsub[2];        // sub-steps per segment = R*S*cp/32 (nsub[0] even when nseg == 2)
  int nseg;
  int N, H, W, P, Q;
  int R, S, stride_shift, stride, pad_h, pad_w, transposed;
  const unsigned short* wp;
  unsigned int w_bytes;
  int Ktot;
  void* out;
  int coutp;
  const float* bias;
  // inference epilogue fusion (eval-mode BatchNorm folded in): out = [prelu](acc * scale + bias
  // [+ residual]) [+ residual]; scale == nullptr -> 1
  const float* scale;
  const float* alpha;
  const unsigned short* residual;
  int res_first;
  int bias9;                      // X3: `bias` is float[9][coutp] by border class (common.h); applied in the copy-out loop
  float* stats;
  int stats_acc;      // accumulator mode (common.h): stats is double[MSML_ACC_ROWS][2][coutp]
  BnBwdFuse bnb;      // bnb.partial != nullptr: fused BatchNorm backward-reduce (common.h)
  long M;
  int tiles_m;
  int parity;         // transposed gather with stride 2: one launch slice (blockIdx.z) per output
                      // parity class, visiting only the taps that hit real input pixels
  int ksplits;        // > 1: split-K over stages, slice z writes f32 partials to out + z*split_stride
  long split_stride;
  int lds_stages;     // A-tile stages actually allocated in LDS (set by launch_fast): B tiles start behind them
};

// exact m / d, m % d for 0 <= m < 2^24 via a float reciprocal (integer division costs ~40
// VALU instructions; the prologue decodes up to 8 rows per thread and K can be as short as 9 stages)
__device__ __forceinline__ void fdivmod(int m, int d, int& q, int& r) {
  q = (int)((float)m * (1.0f / (float)d));
  r = m - q * d;
  if (r < 0) { r += d; q--; }
  if (r >= d) { r -= d; q++; }
}

#define OOB_OFFSET 0x7ffffff0u      // beyond any descriptor range: the DMA writes zeros

__device__ __forceinline__ int swz128(int row) { return (row >> 1) & 7; }   // 128-B rows

typedef __attribute__((address_space(3))) void* lptr_t;

// NST = 2: double buffer, loads of stage s+1 overlap the MFMAs of stage s, __syncthreads().
// NST = 3: ring with TWO stages in flight across the barrier: counted `s_waitcnt vmcnt(L)` (L =
//          DMA instructions per thread per stage) + raw s_barrier, never vmcnt(0) in the loop
//          (cdna_hip_programming.md "Pipelining across barriers").
// FUSE: backward-data launch with the fused BatchNorm backward-reduce epilogue (bf16 out, no
// bias / residual / statistics); a separate instantiation so the plain kernel's register
// allocation is not disturbed by the extra epilogue state.
// X3: split-bf16 output (x3.hip): the f32 results pass through an f32 LDS tile and leave as the three
// planes [hi | lo | hi] of a pixel with 3 * coutp channels; `residual` is read in the same format.
template <typename TOUT, int BM, int BN, int WGM, int WGN, int NST, bool FUSE = false, bool X3 = false>
__global__ void __launch_bounds__(WGM * WGN * 64, FUSE ? 2 : 1) k_conv_fast(const ConvFastArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist in the device pass only
  constexpr int NW = WGM * WGN, NT = NW * 64, RPP = NW * 8;    // waves, threads, rows per pass
  constexpr int NA = BM / RPP, NB = BN / RPP;                  // DMA instructions per thread
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32;
  static_assert(NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1 && (NST == 2 || NST == 3), "tile config");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ABYTES = BM * 128, BBYTES = BN * 128;  // one stage of A / of B
  char* As = smem;                                     // [NST][BM][128 B]
  char* Bs = smem + p.lds_stages * ABYTES;             // [stages][BN][128 B]
  MSML_LDS_REGION(As, p.lds_stages * ABYTES);
  MSML_LDS_REGION(Bs, p.lds_stages * BBYTES);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = t >> 3;                             // RPP rows per pass, lane l -> LDS slot l
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

  // tap walk: r = r0 + rstep * ri (ri < nr), s = s0 + rstep * si (si < ns).  Parity mode
  // (stride-2 transposed gather) keeps only the taps congruent to the class of the output
  // pixel: a 3x3 has 1 / 2 / 2 / 4 of 9 taps per class, a 4x4 has 4 of 16 -- no zero work.
  int cy = 0, cx = 0, r0 = 0, s0 = 0, rstep = 1, nr = p.R, ns = p.S;
  int Pc = p.P, Qc = p.Q;                              // output grid walked by this launch slice
  if (p.parity) {
    cy = blockIdx.z >> 1; cx = blockIdx.z & 1;
    r0 = (cy + p.pad_h) & 1; s0 = (cx + p.pad_w) & 1;
    rstep = 2;
    nr = p.R > r0 ? (p.R - r0 + 1) >> 1 : 0;
    ns = p.S > s0 ? (p.S - s0 + 1) >> 1 : 0;
    Pc = (p.P - cy + 1) >> 1; Qc = (p.Q - cx + 1) >> 1;
  }
  const long Mc = p.parity ? (long)p.N * Pc * Qc : p.M;
  const long prow = (long)bid + (p.parity ? (long)blockIdx.z * p.tiles_m : 0);   // fused-reduce row
  if (m0 >= Mc) {
    if constexpr (FUSE)                                // empty tile of a small parity class
      for (int i = t; i < 3 * BN; i += NT)
        if (n0 + i % BN < p.coutp && !p.bnb.acc) p.bnb.partial[(prow * 3 + i / BN) * p.coutp + n0 + i % BN] = 0.f;
    return;
  }
  const int taps = nr * ns;
  // output row index (pixel in the full P x Q grid) of class-local row m
  auto out_pixel = [&](long m) -> long {
    if (!p.parity) return m;
    int n, rem, oyc, oxc;
    fdivmod((int)m, Pc * Qc, n, rem);
    fdivmod(rem, Qc, oyc, oxc);
    return ((long)n * p.P + 2 * oyc + cy) * p.Q + 2 * oxc + cx;
  };

  // per staged row: pixel index of the window origin and the origin coordinates
  int rowpix[NA], y0[NA], x0[NA];
  const int PQ = Pc * Qc;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    long m = m0 + row0 + i * RPP;
    const bool ok = m < Mc;
    int mm = ok ? (int)m : 0;
    int n, rem, oy, ox;
    fdivmod(mm, PQ, n, rem);
    fdivmod(rem, Qc, oy, ox);
    if (p.parity) { oy = 2 * oy + cy; ox = 2 * ox + cx; }
    const int pb = n * p.H * p.W;
    if (p.transposed) {
      y0[i] = oy + p.pad_h;
      x0[i] = ox + p.pad_w;
      rowpix[i] = pb;
    } else {
      y0[i] = oy * p.stride - p.pad_h;
      x0[i] = ox * p.stride - p.pad_w;
      rowpix[i] = pb + y0[i] * p.W + x0[i];
    }
    if (!ok) y0[i] = -(1 << 20);                       // forces every tap out of range
  }

  __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in[0], 0, (int)p.in_bytes[0], 0x00020000);
  __amdgpu_buffer_rsrc_t rs_in1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.nseg > 1 ? p.in[1] : p.in[0]), 0,
                                                                    (int)(p.nseg > 1 ? p.in_bytes[1] : p.in_bytes[0]), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.w_bytes, 0x00020000);
  // byte offset of this lane's weight row / chunk
  const unsigned int wbase = (unsigned int)((n0 + row0) * p.Ktot + cc * 8) * 2u;
  const unsigned int wstep = (unsigned int)(RPP * p.Ktot) * 2u;

  // this thread's sub-step stream inside the current segment: q = sub, sub + 2, ...
  int ri = 0, si = 0, c32 = sub;
  int nc32 = p.cp[0] >> 5, cp2 = p.cp[0] * 2;          // channel blocks, bytes per pixel
  int wseg = 0;                                        // weight sub-step index of the segment start
  const int nsub0 = taps * (p.cp[0] >> 5), nsub1 = p.nseg > 1 ? taps * (p.cp[1] >> 5) : 0;
  const int qtot = nsub0 + nsub1;
  const int all_stages = (qtot + 1) >> 1;
  // split-K: this launch slice owns stages [st0, stages)
  int st0 = 0, stages = all_stages;
  if (p.ksplits > 1) {
    const int per = (all_stages + p.ksplits - 1) / p.ksplits;
    st0 = blockIdx.z * per;
    stages = st0 + per < all_stages ? st0 + per : all_stages;
    if (st0 > stages) st0 = stages;
    c32 = sub + 2 * st0;
  }
  const int seg0_stages = p.nseg > 1 ? (nsub0 >> 1) : all_stages;   // uniform segment switch
  auto settle = [&]() {
    while (c32 >= nc32) {
      c32 -= nc32;
      if (++si == ns) { si = 0; ++ri; }
    }
  };
  settle();

  // issue the LDS-DMA loads of stage `st` (this thread's current sub-step state) into `buf`
  auto gissue = [&](int st, int buf) {
    char* a = As + buf * ABYTES + wave * 1024;
    char* b = Bs + buf * BBYTES + wave * 1024;
    const bool alive = (2 * st + sub < qtot) & (ri < nr);
    const int r = r0 + rstep * ri, s = s0 + rstep * si;
    const unsigned int coff = (unsigned int)((c32 << 5) + cc * 8) * 2u;
    unsigned int oa[NA], ob[NB];
    if (p.transposed) {
#pragma unroll
      for (int i = 0; i < NA; i++) {
        const int ty = y0[i] - r, tx = x0[i] - s;
        const int iy = ty >> p.stride_shift, ix = tx >> p.stride_shift;
        const bool v = alive & ((ty | tx) >= 0) & (((ty | tx) & (p.stride - 1)) == 0) & (iy < p.H) & (ix < p.W);
        const unsigned int off = (unsigned int)(rowpix[i] + iy * p.W + ix) * (unsigned int)cp2 + coff;
        oa[i] = v ? off : OOB_OFFSET;
      }
    } else {
      const int dpix = r * p.W + s;
#pragma unroll
      for (int i = 0; i < NA; i++) {
        const bool v = alive & ((unsigned)(y0[i] + r) < (unsigned)p.H) & ((unsigned)(x0[i] + s) < (unsigned)p.W);
        const unsigned int off = (unsigned int)(rowpix[i] + dpix) * (unsigned int)cp2 + coff;
        oa[i] = v ? off : OOB_OFFSET;
      }
    }
    // packed-weight column of this sub-step: [segment][r][s][c32] blocks of 32 (64 B)
    const unsigned int wq = wbase + (unsigned int)(wseg + (r * p.S + s) * nc32 + c32) * 64u;
#pragma unroll
    for (int i = 0; i < NB; i++) ob[i] = alive ? wq + i * wstep : OOB_OFFSET;
    if (st < seg0_stages) {
#pragma unroll
      for (int i = 0; i < NA; i++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in0, (lptr_t)(a + i * NW * 1024), 16, oa[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NA; i++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in1, (lptr_t)(a + i * NW * 1024), 16, oa[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(b + i * NW * 1024), 16, ob[i], 0, 0, 0);
  };
  // move this thread's state from stage st to stage st + 1
  auto advance = [&](int st) {
    if (st + 1 == seg0_stages && p.nseg > 1) {         // (uniform) first stage of segment 1
      ri = 0; si = 0; c32 = sub;
      wseg = p.R * p.S * (p.cp[0] >> 5);
      nc32 = p.cp[1] >> 5;
      cp2 = p.cp[1] * 2;
    } else {
      c32 += 2;
    }
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
  // fragment offsets: row (arow0 + r32), 16-B chunk ((2 kk + h) ^ swz); rows + 32 i keep the same
  // swizzle (32 >> 1 == 0 mod 8), so tiles i / j and the stage buffer are immediates
  int aoff[4], boff[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    aoff[kk] = (arow0 + r32) * 128 + (((kk * 2 + h) ^ swz128(arow0 + r32)) << 4);
    boff[kk] = (brow0 + r32) * 128 + (((kk * 2 + h) ^ swz128(brow0 + r32)) << 4);
  }

  // fragment reads of step kk+1 are issued before the MFMAs of step kk (register double buffer)
  auto compute = [&](int buf) {
    const char* A = As + buf * ABYTES;
    const char* B = Bs + buf * BBYTES;
    u32x4 a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; i++) a[0][i] = *reinterpret_cast<const u32x4*>(A + aoff[0] + i * 4096);
#pragma unroll
    for (int j = 0; j < TN; j++) b[0][j] = *reinterpret_cast<const u32x4*>(B + boff[0] + j * 4096);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {                   // 4 x 16 k per 64-wide stage
      const int cb = kk & 1, nb = cb ^ 1;
      if (kk + 1 < 4) {
#pragma unroll
        for (int i = 0; i < TM; i++) a[nb][i] = *reinterpret_cast<const u32x4*>(A + aoff[kk + 1] + i * 4096);
#pragma unroll
        for (int j = 0; j < TN; j++) b[nb][j] = *reinterpret_cast<const u32x4*>(B + boff[kk + 1] + j * 4096);
      }
#ifndef MSML_NO_SCHED_FENCE
      __builtin_amdgcn_sched_barrier(0);               // keep the reads of step kk + 1 ahead of these MFMAs
#endif
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, a[cb][i]), __builtin_bit_cast(bf16x8, b[cb][j]), acc[i][j], 0, 0, 0);
#ifndef MSML_NO_SCHED_FENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };

  if constexpr (NST == 2) {
    gissue(st0, 0);
    __syncthreads();                                   // (hipcc drains vmcnt before the barrier)
    int cur = 0;
    for (int st = st0; st < stages; st++) {
      if (st + 1 < stages) {
        advance(st);
#ifndef MSML_ABLATE_LOADS
        gissue(st + 1, cur ^ 1);
#endif
      }
#ifndef MSML_ABLATE_COMPUTE
      compute(cur);
#endif
      __syncthreads();
      cur ^= 1;
    }
  } else {
    constexpr int L = NA + NB;                         // DMA instructions per thread per stage
    gissue(0, 0);
    if (stages > 1) {
      advance(0);
      gissue(1, 1);
    }
    int cur = 0, nxt = 2;                              // buffer of stage st, of stage st + 2
    for (int st = 0; st < stages; st++) {
      if (st + 1 < stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // stage st landed for all waves; buffer
      if (st + 2 < stages) {                           // `nxt` (read in st-1) is free again
        advance(st + 1);
        gissue(st + 2, nxt);
      }
      compute(cur);
      cur = cur == 2 ? 0 : cur + 1;
      nxt = nxt == 2 ? 0 : nxt + 1;
    }
    __syncthreads();
  }

  // ---------------- epilogue (same contract as k_conv_igemm) -----------------------------------
  // bf16 results go through an LDS transpose tile [BM][BN + 8] and leave as 16-B row chunks;
  // f32 results (head logits) are stored directly.
#ifdef MSML_ABLATE_EPILOGUE
  if (p.M >= 0) return;
#endif
  TOUT* outp = reinterpret_cast<TOUT*>(p.out) + (p.ksplits > 1 ? blockIdx.z * p.split_stride : 0);
  constexpr bool VIA_LDS = sizeof(TOUT) == 2;
  constexpr int OP = BN + 8;                           // tile pitch in elements
  constexpr int OPF = BN + 4;                          // pitch of the f32 tile (X3)
  unsigned short* otile = reinterpret_cast<unsigned short*>(smem);
  float* otf = reinterpret_cast<float*>(smem);
  if (X3) MSML_LDS_REGION(otf, BM * OPF * 4);
  else if (VIA_LDS) MSML_LDS_REGION(otile, BM * OP * 2);
  // fused BatchNorm backward-reduce: the saved BatchNorm input of this thread's copy-out chunks is
  // requested first, so the loads fly during the accumulator -> LDS transpose
  constexpr int C8 = BN / 8, ITERS = FUSE ? BM * C8 / NT : 1;
  const int c8 = t % C8, ccol = n0 + c8 * 8;
  u32x4 xr[ITERS];
  unsigned int oo[ITERS];                              // element offsets (tensors < 2^31 elements)
  constexpr unsigned int NO_CHUNK = 0xffffffffu;
  if constexpr (FUSE) {
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
      const long m = m0 + (t + k * NT) / C8;
      oo[k] = (m < Mc && ccol < p.coutp) ? (unsigned int)(out_pixel(m) * p.coutp + ccol) : NO_CHUNK;
#ifdef FAST_ABL_BNB_LOAD
      xr[k] = u32x4{0, 0, 0, 0};
#else
      xr[k] = oo[k] != NO_CHUNK ? *reinterpret_cast<const u32x4*>(p.bnb.x + oo[k]) : u32x4{0, 0, 0, 0};
#endif
    }
  }
  float s1v[TN], s2v[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int lcol = brow0 + j * 32 + r32;
    const int col = n0 + lcol;
    const bool cok = col < p.coutp;
    const float bv = (p.bias && cok && !(X3 && p.bias9)) ? p.bias[col] : 0.f;
    const float sv = (p.scale && cok) ? p.scale[col] : 1.f;
    const bool act_here = p.alpha && !(p.residual && p.res_first) && !(X3 && p.bias9);
    const float av = (act_here && cok) ? p.alpha[col] : 1.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int row = arow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        long m = m0 + row;
        float v = acc[i][j][e] * sv + bv;
        if (act_here) v = v > 0.f ? v : v * av;
        if constexpr (X3) otf[row * OPF + lcol] = v;
        else if (VIA_LDS) otile[row * OP + lcol] = f2bf(v);
        if (m < Mc && cok) {
          if (!VIA_LDS) store1<TOUT>(outp + out_pixel(m) * p.coutp + col, v);
          s1 += v;
          s2 += v * v;
        }
      }
    }
    s1v[j] = s1;
    s2v[j] = s2;
  }
  if (VIA_LDS) {
    static_assert(NT % C8 == 0 && (BM * C8) % NT == 0, "a thread keeps one 8-channel chunk in the copy-out loop");
    const int col = ccol;
    if constexpr (FUSE) {
      // fused BatchNorm backward-reduce (x chunks were fetched before the transpose)
      const bool cok = col < p.coutp;
      __syncthreads();
      BnbCoef bk;
      if (cok) bk = bnb_load_coef(p.bnb, col);
      float bq[3][8];
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) bq[q][j] = 0.f;
#pragma unroll
      for (int k = 0; k < ITERS; k++) {
        if (oo[k] != NO_CHUNK) {
          const int row = (t + k * NT) / C8;
          u32x4 v = *reinterpret_cast<const u32x4*>(otile + row * OP + c8 * 8);
          *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.out) + oo[k]) = v;
#ifndef FAST_ABL_BNB_ACC
          bnb_accum(bk, p.bnb.alpha != nullptr, load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v)),
                    load8<unsigned short>(reinterpret_cast<const unsigned short*>(&xr[k])), bq);
#endif
        }
      }
      // the NT / C8 threads that share a channel chunk meet in LDS; fixed-order sums
      constexpr int G = NT / C8;
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);
      MSML_LDS_REGION(red, G * 3 * BN * 4);
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) red[((t / C8) * 3 + q) * BN + c8 * 8 + j] = bq[q][j];
      __syncthreads();
      for (int i = t; i < 3 * BN; i += NT) {
        const int q = i / BN, c = i % BN;
        float sum = 0.f;
        for (int g = 0; g < G; g++) sum += red[(g * 3 + q) * BN + c];
        if (n0 + c < p.coutp) bnb_emit(p.bnb.partial, p.bnb.acc, prow, q, p.coutp, n0 + c, sum);
      }
    } else if constexpr (X3) {
      __syncthreads();
      unsigned short* o16 = reinterpret_cast<unsigned short*>(p.out);
      for (int idx = t; idx < BM * C8; idx += NT) {
        const int row = idx / C8;
        const long m = m0 + row;
        if (m < Mc && col < p.coutp) {
          Vec8 a = load8<float>(otf + row * OPF + c8 * 8);
          const long opix = out_pixel(m);
          const long o = opix * (3L * p.coutp) + col;
          if (p.bias9) {                               // border-class shift (+ the PReLU that follows it)
            const int ox = (int)(opix % p.Q), oy = (int)((opix / p.Q) % p.P);
            const Vec8 b9 = load8<float>(p.bias + border_class(oy, ox, p.P, p.Q) * p.coutp + col);
            const bool act9 = p.alpha && !(p.residual && p.res_first);
#pragma unroll
            for (int q = 0; q < 8; q++) {
              float z = a.v[q] + b9.v[q];
              if (act9) z = z > 0.f ? z : z * p.alpha[col + q];
              a.v[q] = z;
            }
          }
          if (p.residual) {
            const Vec8 rh = load8<unsigned short>(p.residual + o), rl = load8<unsigned short>(p.residual + o + p.coutp);
#pragma unroll
            for (int q = 0; q < 8; q++) {
              float z = a.v[q] + (rh.v[q] + rl.v[q]);
              if (p.res_first && p.alpha) z = z > 0.f ? z : z * p.alpha[col + q];
              a.v[q] = z;
            }
          }
          const Vec8 hi = round8<unsigned short>(a);
          Vec8 lo;
#pragma unroll
          for (int q = 0; q < 8; q++) lo.v[q] = a.v[q] - hi.v[q];
          store8<unsigned short>(o16 + o, hi);
          store8<unsigned short>(o16 + o + p.coutp, lo);
          store8<unsigned short>(o16 + o + 2 * p.coutp, hi);
        }
      }
    } else {
      __syncthreads();
      for (int idx = t; idx < BM * C8; idx += NT) {
        const int row = idx / C8;
        const long m = m0 + row;
        if (m < Mc && col < p.coutp) {
          u32x4 v = *reinterpret_cast<const u32x4*>(otile + row * OP + c8 * 8);
          const long o = out_pixel(m) * p.coutp + col;
          if (p.residual) {                            // fused residual (+ PReLU after it)
            Vec8 a = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&v));
            Vec8 r = load8<unsigned short>(p.residual + o);
#pragma unroll
            for (int q = 0; q < 8; q++) {
              float z = a.v[q] + r.v[q];
              if (p.res_first && p.alpha) z = z > 0.f ? z : z * p.alpha[col + q];
              a.v[q] = z;
            }
            store8<unsigned short>(reinterpret_cast<unsigned short*>(&v), a);
          }
          *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.out) + o) = v;
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    MSML_LDS_REGION(red, WGM * 2 * BN * 4);
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
    // one (sum, sumsq) row pair per SROWS pixel rows: 128 when BN >= 128 (so the row count does
    // not depend on BM = 128 / 256), the whole tile otherwise
    constexpr int SROWS = (BN >= 128 && BM >= 128) ? 128 : BM;   // (BM = 64: accumulator-mode launches only)
    constexpr int NH = BM / SROWS, WPH = WGM / NH;     // halves per tile, wave rows per half
    for (int i = t; i < NH * 2 * BN; i += NT) {
      int hf = i / (2 * BN), which = (i / BN) & 1, c = i % BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WPH; w++) v += red[((hf * WPH + w) * 2 + which) * BN + c];
      int col = n0 + c;
      long srow = (long)bid * NH + hf;
      if (col < p.coutp && srow * SROWS < Mc) stats_emit(p.stats, p.stats_acc, srow, which, p.coutp, col, v);
    }
  }
#endif
}

template <typename TOUT, int BM, int BN, int WGM, int WGN, int NST, bool FUSE = false, bool X3 = false>
static void launch_fast(ConvFastArgs& a, hipStream_t st) {
  // parity mode: tiles sized for the largest class (cy = cx = 0); smaller classes exit early
  const long mtile = a.parity ? (long)a.N * ((a.P + 1) / 2) * ((a.Q + 1) / 2) : a.M;
  a.tiles_m = cdiv(mtile, BM);
  dim3 grid(a.tiles_m, cdiv(a.coutp, BN), a.parity ? 4 : (a.ksplits > 1 ? a.ksplits : 1));
  size_t lds = (size_t)NST * (BM + BN) * 128;
  // K of one stage (1x1 convs over <= 64 channels, the im2col'd stems): the ring never advances, so only
  // one buffer is needed -- 2 - 3 workgroups fit a CU instead of 1 - 2 and their load / epilogue phases
  // overlap (these launches are pure streaming: X in, Y out)
  static const bool one_stage_ok = getenv("MSML_CONV_NO_ONE_STAGE") == nullptr;
  a.lds_stages = NST;
  if (one_stage_ok && NST == 2 && a.ksplits <= 1 && (a.nsub[0] + a.nsub[1] + 1) / 2 <= 1) {
    a.lds_stages = 1;
    lds = (size_t)(BM + BN) * 128;
  }
  size_t olds = X3 ? (size_t)BM * (BN + 4) * 4 : (sizeof(TOUT) == 2 ? (size_t)BM * (BN + 8) * 2 : 0);
  if (olds > lds) lds = olds;
  // the fused BatchNorm backward sums meet in LDS as [threads per channel chunk][3][BN] floats: 24.5 KB for the 64-row
  // tiles -- more than ONE stage of those tiles holds (a single-stage 1x1 launch on 64 x 64 / 64 x 128 tiles wrote past
  // its allocation and returned wrong sums; found by the pointwise-kernel test of round 4, no layer of the MSML
  // networks takes that combination)
  const size_t rlds = FUSE ? (size_t)(WGM * WGN * 64 / (BN / 8)) * 3 * BN * 4 : 0;
#ifdef MSML_LDS_GUARD
  static const bool drop_rlds = getenv("MSML_LDS_GUARD_DROP_RLDS") != nullptr;   // guard self-test: the round-3 allocation
  if (rlds > lds && !drop_rlds) lds = rlds;
#else
  if (rlds > lds) lds = rlds;
#endif
  if (lds > 64 * 1024) {                               // above the default dynamic-LDS limit
    static std::once_flag once;
    std::call_once(once, [] {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_fast<TOUT, BM, BN, WGM, WGN, NST, FUSE, X3>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
  }
  k_conv_fast<TOUT, BM, BN, WGM, WGN, NST, FUSE, X3><<<grid, dim3(WGM * WGN * 64), lds, st>>>(a);
}

bool msml_conv_halo_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out,
                             int coutp, float* stats, int N, int H, int W, int P, int Q, int R, int S,
                             int stride, int pad_h, int pad_w, int transposed, hipStream_t st,
                             const float* scale, const float* alpha, const void* residual, int res_first,
                             const BnBwdFuse* bnb, int* bnb_rows, const BnIn* xin = nullptr, int x3 = 0,
                             const BnBwdIn* bin = nullptr);

// Called by msml_conv2d (conv_igemm.hip) when the fast-path conditions hold.  Returns false if
// this kernel does not apply.
bool msml_conv_ws_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out,
                           int coutp, float* stats, int N, int H, int W, int P, int Q, int R, int S,
                           int stride, int pad_h, int pad_w, int transposed, hipStream_t st,
                           const float* scale, const float* alpha, const void* residual, int res_first,
                           const BnBwdFuse* bnb, int* bnb_rows, const BnIn* xin = nullptr);

bool msml_conv_r32_dispatch(const void* in0, int c0p, const void* wp, int kop, int ktot, const float* bias, void* out,
                            int coutp, float* stats, int stats_acc, int N, int H, int W, int P, int Q, int R, int S,
                            int stride, int pad_h, int pad_w, int transposed, hipStream_t st, const float* scale,
                            const float* alpha, const void* residual, const BnBwdFuse* bnb);

bool msml_conv_pw_dispatch(const void* in0, int c0p, const void* wp, int kop, int ktot, const float* bias, void* out,
                           int coutp, float* stats, int stats_acc, int N, int H, int W, int P, int Q, int R, int S,
                           int stride, int pad_h, int pad_w, hipStream_t st, const float* scale, const float* alpha,
                           const void* residual, int res_first, const BnBwdFuse* bnb);

bool msml_conv_halo2_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out, int coutp,
                              float* stats, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                              int pad_w, int transposed, hipStream_t st, const float* scale, const float* alpha,
                              const void* residual, int res_first, const BnBwdFuse* bnb, int* bnb_rows, int x3 = 0);

bool msml_conv_s2r_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out, int coutp,
                            float* stats, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                            int transposed, hipStream_t st, const float* scale, const float* alpha, const void* residual,
                            const BnBwdFuse* bnb, int* bnb_rows);

bool msml_conv_s2r_x3_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out, int coutp, int N,
                               int H, int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w, int transposed,
                               hipStream_t st, const float* scale, const float* alpha, const void* residual, int res_first);

bool msml_conv_fast_dispatch(const void* in0, int c0p, const void* in1, int c1p, const void* wp, int kop,
                             const float* bias, void* out, int coutp, float* stats, int N, int H,
                             int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                             int transposed, int in_dtype, int out_dtype, int bn, hipStream_t st,
                             const float* scale, const float* alpha, const void* residual, int res_first,
                             const BnBwdFuse* bnb, int* bnb_rows) {
  if (in_dtype != MSML_BF16) return false;
  const bool x3 = out_dtype == MSML_BF16X3;            // split-bf16 output planes (x3.hip)
  if (x3 && (bnb || stats)) return false;
  if (bnb && (out_dtype != MSML_BF16 || stats)) return false;
  if ((scale || alpha || residual) && out_dtype != MSML_BF16 && !x3) return false;
  if (c0p % 32 != 0 || (in1 && c1p % 32 != 0)) return false;
  // 32 -> 32 channel 3x3 layers (conv2 of the first FM stage's bottlenecks): conv_r32.hip
  if (!in1 && out_dtype == MSML_BF16 &&
      msml_conv_r32_dispatch(in0, c0p, wp, kop, R * S * c0p, bias, out, coutp, stats, stats ? msml_tl_stats_acc : 0, N, H,
                             W, P, Q, R, S, stride, pad_h, pad_w, transposed, st, scale, alpha, residual, bnb)) {
    if (bnb_rows) *bnb_rows = 1;
    return true;
  }
  // 1x1 / stride-1 layers (FM bottlenecks, im2col'd stems and their backward-data convs): conv_pw.hip
  if (!in1 && out_dtype == MSML_BF16 &&
      msml_conv_pw_dispatch(in0, c0p, wp, kop, R * S * c0p, bias, out, coutp, stats, stats ? msml_tl_stats_acc : 0, N, H,
                            W, P, Q, R, S, stride, pad_h, pad_w, st, scale, alpha, residual, res_first, bnb)) {
    if (bnb_rows) *bnb_rows = 1;
    return true;
  }
  // (64 -> 64 stride-1 forward / plain backward-data: the register-weights kernel first, conv_s2r.hip)
  if (stride == 1 && !in1 && !bnb && out_dtype == MSML_BF16 &&
      msml_conv_s2r_dispatch(in0, c0p, wp, kop, bias, out, coutp, stats, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed,
                             st, scale, alpha, residual, bnb, bnb_rows))
    return true;
  if (!in1 && out_dtype == MSML_BF16 &&
      msml_conv_ws_dispatch(in0, c0p, wp, kop, bias, out, coutp, stats, N, H, W, P, Q, R, S, stride, pad_h,
                            pad_w, transposed, st, scale, alpha, residual, res_first, bnb, bnb_rows))
    return true;
  if (!in1 && out_dtype == MSML_BF16 &&
      msml_conv_halo_dispatch(in0, c0p, wp, kop, bias, out, coutp, stats, N, H, W, P, Q, R, S, stride, pad_h,
                              pad_w, transposed, st, scale, alpha, residual, res_first, bnb, bnb_rows))
    return true;
  // stride-2 3x3 layers (forward through parity planes, backward-data per output class) and 7x7 maps (2 x 2 image
  // mosaics): conv_halo2.hip
  if (!in1 && out_dtype == MSML_BF16 &&
      msml_conv_halo2_dispatch(in0, c0p, wp, kop, bias, out, coutp, stats, N, H, W, P, Q, R, S, stride, pad_h, pad_w,
                               transposed, st, scale, alpha, residual, res_first, bnb, bnb_rows))
    return true;
  // 64 -> 64 channel stride-2 3x3 layers (weights in registers, persistent): conv_s2r.hip
  if (!in1 && out_dtype == MSML_BF16 &&
      msml_conv_s2r_dispatch(in0, c0p, wp, kop, bias, out, coutp, stats, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed,
                             st, scale, alpha, residual, bnb, bnb_rows))
    return true;
  // split-bf16 inference: the 64 -> 64 channel 3x3 / stride-1 layers with both weight planes in registers (conv_s2r.hip)
  if (x3 && !in1 &&
      msml_conv_s2r_x3_dispatch(in0, c0p, wp, kop, bias, out, coutp, N, H, W, P, Q, R, S, stride, pad_h, pad_w, transposed, st,
                                scale, alpha, residual, res_first))
    return true;
  // split-bf16 inference: stride-2 layers and 7x7 / 4x4 maps (conv_halo2.hip)
  if (x3 && !in1 &&
      msml_conv_halo2_dispatch(in0, c0p, wp, kop, bias, out, coutp, nullptr, N, H, W, P, Q, R, S, stride, pad_h, pad_w,
                               transposed, st, scale, alpha, residual, res_first, nullptr, nullptr, 1))
    return true;
  // split-bf16 inference: 3x3 / stride-1 layers of the 28x28 / 14x14 stages on the halo kernel (c0p is 3 x logical)
  if (x3 && !in1 &&
      msml_conv_halo_dispatch(in0, c0p, wp, kop, bias, out, coutp, stats, N, H, W, P, Q, R, S, stride, pad_h,
                              pad_w, transposed, st, scale, alpha, residual, res_first, nullptr, nullptr, nullptr, 1))
    return true;
  ConvFastArgs a;
  a.in[0] = (const unsigned short*)in0;
  a.in[1] = (const unsigned short*)in1;
  a.cp[0] = c0p; a.cp[1] = c1p;
  a.nseg = in1 ? 2 : 1;
  a.nsub[0] = R * S * (c0p / 32);
  a.nsub[1] = in1 ? R * S * (c1p / 32) : 0;
  if (a.nseg == 2 && (a.nsub[0] & 1)) return false;    // segment switch must be stage-aligned
  a.Ktot = 32 * (a.nsub[0] + a.nsub[1]);
  const long in0_bytes = (long)N * H * W * c0p * 2, in1_bytes = (long)N * H * W * c1p * 2;
  const long w_bytes = (long)kop * a.Ktot * 2;
  if ((long)N * P * Q >= (1L << 24)) return false;      // float-reciprocal pixel decode
  if (in0_bytes >= 0x7fffff00L || in1_bytes >= 0x7fffff00L || w_bytes >= 0x7fffff00L) return false;
  a.in_bytes[0] = (unsigned int)in0_bytes;
  a.in_bytes[1] = (unsigned int)in1_bytes;
  a.w_bytes = (unsigned int)w_bytes;
  a.N = N; a.H = H; a.W = W; a.P = P; a.Q = Q; a.R = R; a.S = S;
  a.stride = stride;
  a.stride_shift = stride == 1 ? 0 : (stride == 2 ? 1 : 2);
  a.pad_h = pad_h; a.pad_w = pad_w; a.transposed = transposed;
  a.wp = (const unsigned short*)wp; a.out = out; a.coutp = coutp; a.bias = bias; a.stats = stats;
  a.stats_acc = stats ? msml_tl_stats_acc : 0;
  a.scale = scale; a.alpha = alpha; a.residual = (const unsigned short*)residual; a.res_first = res_first;
  a.bias9 = (x3 && bias) ? msml_tl_bias9 : 0;
  a.bnb = BnBwdFuse{};
  if (bnb) a.bnb = *bnb;
  a.M = (long)N * P * Q;
  a.ksplits = 1;
  a.split_stride = 0;
  // stride-2 transposed gather by output parity class (needs stage-aligned segment switches)
  a.parity = 0;
  if (transposed && stride == 2 && !getenv("MSML_CONV_NO_PARITY") && stats == nullptr) {
    bool ok = true;
    if (a.nseg == 2)
      for (int c = 0; c < 4 && ok; c++) {
        int r0 = ((c >> 1) + pad_h) & 1, s0 = ((c & 1) + pad_w) & 1;
        int nr = R > r0 ? (R - r0 + 1) / 2 : 0, ns = S > s0 ? (S - s0 + 1) / 2 : 0;
        ok = ((nr * ns * (c0p / 32)) & 1) == 0;
      }
    a.parity = ok ? 1 : 0;
  }
  // big tile only when it still fills the 256 CUs
  static const int use_big = getenv("MSML_CONV_BIG_TILE") ? atoi(getenv("MSML_CONV_BIG_TILE")) : 0;
  const bool big = use_big && bn == 128 && (long)cdiv(a.M, 256) * cdiv(coutp, 128) >= 256;
  // Small maps (7x7 / 4x4 stages, the OSB's deepest levels): the default 128- / 256-row tile leaves the chip half
  // empty (512 -> 512 @ 4x4: 128 workgroups, 512 -> 8 @ 4x4: 16), a 64-row tile doubles / quadruples the workgroup
  // count.  Statistics / fused BatchNorm sums only in accumulator mode (the partial-row formats are sized by the
  // default tile), no split-K, bf16 output.
  static const bool small_ok = getenv("MSML_CONV_NO_SMALL_M") == nullptr;
  const long def_wgs = (long)cdiv(a.parity ? (long)N * ((P + 1) / 2) * ((Q + 1) / 2) : a.M, bn == 128 ? 128 : 256) *
                       cdiv(coutp, bn) * (a.parity ? 4 : 1);
  static const long small_wgs = getenv("MSML_CONV_SMALL_M_WGS") ? atol(getenv("MSML_CONV_SMALL_M_WGS")) : 200;
  const bool small_m = small_ok && !x3 && out_dtype == MSML_BF16 && def_wgs <= small_wgs && a.M >= 2048 &&
                       (!stats || a.stats_acc) && (!bnb || bnb->acc);
  // split-bf16 inference on the small maps (7x7 / 4x4: the deep OSB levels and their GCMs, the 256- / 128-channel 7x7
  // layers): 64-row tiles whatever the batch -- the kernel choice of this mode must not depend on N (conv_halo2.hip) -- a
  // 512 -> 32 @ 4x4 line conv is 16 workgroups of 256 rows x a K of 10 752 otherwise
  static const bool x3_small = getenv("MSML_NO_X3_SMALL_M") == nullptr;
  if (x3 && x3_small && !a.parity && P * Q <= 64 && !transposed) {
    if (bn == 128) launch_fast<unsigned short, 64, 128, 2, 2, 2, false, true>(a, st);
    else if (bn == 64) launch_fast<unsigned short, 64, 64, 2, 2, 2, false, true>(a, st);
    else launch_fast<unsigned short, 64, 32, 2, 1, 2, false, true>(a, st);
    return true;
  }
  if (small_m) {
    if (bnb) {
      if (bias || residual || scale || alpha) return false;
      if (bn == 128) launch_fast<unsigned short, 64, 128, 2, 2, 2, true>(a, st);
      else if (bn == 64) launch_fast<unsigned short, 64, 64, 2, 2, 2, true>(a, st);
      else launch_fast<unsigned short, 64, 32, 2, 1, 2, true>(a, st);
    } else {
      if (bn == 128) launch_fast<unsigned short, 64, 128, 2, 2, 2>(a, st);
      else if (bn == 64) launch_fast<unsigned short, 64, 64, 2, 2, 2>(a, st);
      else launch_fast<unsigned short, 64, 32, 2, 1, 2>(a, st);
    }
    if (bnb_rows) *bnb_rows = a.tiles_m * (a.parity ? 4 : 1);
    return true;
  }
#define FAST_CASE(TO)                                                   \
  if (big && use_big == 3) launch_fast<TO, 256, 128, 4, 2, 3>(a, st);   \
  else if (big) launch_fast<TO, 256, 128, 4, 2, 2>(a, st);              \
  else if (bn == 128) launch_fast<TO, 128, 128, 2, 2, 2>(a, st);        \
  else if (bn == 64) launch_fast<TO, 256, 64, 4, 1, 2>(a, st);          \
  else launch_fast<TO, 256, 32, 4, 1, 2>(a, st);
  if (bnb) {
    if (bias || residual || scale || alpha) return false;
    if (bn == 128) launch_fast<unsigned short, 128, 128, 2, 2, 2, true>(a, st);
    else if (bn == 64) launch_fast<unsigned short, 256, 64, 4, 1, 2, true>(a, st);
    else launch_fast<unsigned short, 256, 32, 4, 1, 2, true>(a, st);
  } else if (x3) {
    if (bn == 128) launch_fast<unsigned short, 128, 128, 2, 2, 2, false, true>(a, st);
    else if (bn == 64) launch_fast<unsigned short, 256, 64, 4, 1, 2, false, true>(a, st);
    else launch_fast<unsigned short, 256, 32, 4, 1, 2, false, true>(a, st);
  } else if (out_dtype == MSML_BF16) { FAST_CASE(unsigned short) }
  else { FAST_CASE(float) }
#undef FAST_CASE
  if (bnb_rows) *bnb_rows = a.tiles_m * (a.parity ? 4 : 1);
  return true;
}


// Split-K variant for skinny GEMMs with a huge K (PartialFC dX = dcos[N x C] . Wn[C x E]:
// K = classes, only a handful of output tiles): `ksplits` launch slices each accumulate a K
// range into an f32 slab (ws[z][M][coutp]); the caller sums the slabs in a fixed order.
bool msml_conv_fast_splitk(const void* in0, int c0p, const void* wp, int kop, float* ws, int coutp, int N,
                           int ksplits, hipStream_t st) {
  if (c0p % 32 != 0) return false;
  ConvFastArgs a;
  a.in[0] = (const unsigned short*)in0; a.in[1] = nullptr;
  a.cp[0] = c0p; a.cp[1] = 0;
  a.nseg = 1;
  a.nsub[0] = c0p / 32; a.nsub[1] = 0;
  a.Ktot = c0p;
  const long in_bytes = (long)N * c0p * 2, w_bytes = (long)kop * a.Ktot * 2;
  if (in_bytes >= 0x7fffff00L || w_bytes >= 0x7fffff00L) return false;
  a.in_bytes[0] = (unsigned int)in_bytes; a.in_bytes[1] = 0;
  a.w_bytes = (unsigned int)w_bytes;
  a.N = N; a.H = 1; a.W = 1; a.P = 1; a.Q = 1; a.R = 1; a.S = 1;
  a.stride = 1; a.stride_shift = 0; a.pad_h = 0; a.pad_w = 0; a.transposed = 0;
  a.wp = (const unsigned short*)wp; a.out = ws; a.coutp = coutp; a.bias = nullptr; a.stats = nullptr; a.stats_acc = 0;
  a.scale = nullptr; a.alpha = nullptr; a.residual = nullptr; a.res_first = 0; a.bias9 = 0;
  a.bnb = BnBwdFuse{};
  a.M = N;
  a.parity = 0;
  a.ksplits = ksplits;
  a.split_stride = (long)N * coutp;
  launch_fast<float, 128, 128, 2, 2, 2>(a, st);
  return true;
}
