// Fast path of the weight-gradient kernel for bf16 (same math, ABI and slab format as
// conv_wgrad.hip, which stays the exact-f32 parity path).
//
//   * pixel stages of 64, double-buffered; U / V tiles go HBM/L2 -> LDS by buffer LDS-DMA
//     (16 B per lane, 32-bit offsets, out-of-range offset = zero fill: borders, tails and
//     channel padding cost a v_cndmask),
//   * DMA slot mapping: wave w owns pixel group w (16 pixels) of the stage and issues one
//     instruction per 32-channel group; lane l covers pixel l/4, 16-B chunk l%4.  Every thread
//     therefore decodes ONE pixel per stage (n, py, px -> shifted tap position) and its loads
//     differ only by immediates,
//   * the LDS image is blocked [pixel group][channel group][16 px][64 B]: the four pixel rows of
//     a transposing ds_read_b64_tr_b16 block are 64 B apart inside one 256-B window -> conflict
//     free without padding or swizzle, and a fragment address is one VGPR + immediates.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) void* lptr_t;

#define OOB_OFFSET 0x7ffffff0u

#define WF_MAXGROUP 8
struct WgradFastArgs {
  const unsigned short* u[WF_MAXGROUP]; int up; unsigned int u_bytes;     // one (dY, X, dW) per grouped layer
  const unsigned short* v[WF_MAXGROUP]; int vp; unsigned int v_bytes;
  int zper;                   // splits per layer: grid.z = layers x zper
  int N, H, W, P, Q, R, S, stride, pad_h, pad_w;
  float* ws;
  int arows;
  long Mpix;
  int chunk;                  // pixels per split (multiple of 64)
  float rcp_pq, rcp_q;
  // splits == 1 (many output tiles, short K: the PartialFC / fc weight gradients): no slab, the
  // tile goes straight into dW[a][boff + b][tap] (saves writing and re-reading |dW| floats)
  float* dw[WF_MAXGROUP];
  int direct;
  int A, Breal, Btot, boff, accumulate;
};

__device__ __forceinline__ void divmodf(int m, int d, float rcp, int& q, int& r) {
  q = (int)((float)m * rcp);
  r = m - q * d;
  if (r < 0) { r += d; q--; }
  if (r >= d) { r -= d; q++; }
}

// NTW > 1 (narrow V, vp <= 64): the BB columns of the tile are NTW consecutive taps x 64
// channels, so one U (dY) tile and its fragments serve NTW taps -- the 64-channel layers
// otherwise run 4 MFMAs per wave per 16 KB stage and are barrier / traffic bound.
template <int BA, int BB, int NTW>
__global__ void __launch_bounds__(256) k_wgrad_fast(const WgradFastArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int GA = BA / 32, GB = BB / 32;             // 32-channel groups per operand
  constexpr int GPT = GB / NTW;                         // V groups per tap
  constexpr int TM = BA / 64, TN = BB / 64;             // 32x32 MFMA tiles per wave (2x2 waves)
  constexpr int UBYTES = 64 * BA * 2, STAGE = 64 * (BA + BB) * 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  MSML_LDS_REGION(smem, 2 * STAGE);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int btiles = NTW > 1 ? (p.vp + 63) / 64 : (p.vp + BB - 1) / BB;   // NTW > 1: 64-channel chunks
  const int a0 = blockIdx.x * BA;
  const int tap0 = (blockIdx.y / btiles) * NTW, b0 = (blockIdx.y % btiles) * (NTW > 1 ? 64 : BB);
  const int taps = p.R * p.S;
  const int layer = __builtin_amdgcn_readfirstlane(blockIdx.z / p.zper);    // several same-shape layers per launch
  const int split = blockIdx.z - layer * p.zper;
  const long k_begin = (long)split * p.chunk;
  long k_end = k_begin + p.chunk;
  if (k_end > p.Mpix) k_end = p.Mpix;
  const int nstages = k_begin < k_end ? (int)((k_end - k_begin + 63) / 64) : 0;
  const int PQ = p.P * p.Q;

  __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)p.u[layer], 0, (int)p.u_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)p.v[layer], 0, (int)p.v_bytes, 0x00020000);

  // DMA slot of this lane: pixel (16 * wave + lane / 4) of the stage, 16-B chunk lane % 4 of
  // every 32-channel group
  const int lp = lane >> 2, lc = lane & 3;
  const int pix_in_stage = 16 * wave + lp;
  // per-group channel validity (tiles may overhang the tensor's channel count)
  unsigned int uoffc[GA], voffc[GB];
#pragma unroll
  for (int g = 0; g < GA; g++) {
    int c = a0 + g * 32 + lc * 8;
    uoffc[g] = c < p.up ? (unsigned int)c * 2u : OOB_OFFSET;
  }
#pragma unroll
  for (int g = 0; g < GB; g++) {
    int c = b0 + (g % GPT) * 32 + lc * 8;
    voffc[g] = (c < p.vp && tap0 + g / GPT < taps) ? (unsigned int)c * 2u : OOB_OFFSET;
  }

  auto gissue = [&](int stage, int buf) {
    const long m = k_begin + (long)stage * 64 + pix_in_stage;
    const bool in = m < k_end;
    int n, rem, py, px;
    divmodf(in ? (int)m : 0, PQ, p.rcp_pq, n, rem);
    divmodf(rem, p.Q, p.rcp_q, py, px);
    const int iy0 = py * p.stride - p.pad_h, ix0 = px * p.stride - p.pad_w;
    const unsigned int ubase = in ? (unsigned int)m * (unsigned int)(p.up * 2) : OOB_OFFSET;
    unsigned int vbase[NTW];
#pragma unroll
    for (int tw = 0; tw < NTW; tw++) {
      const int tp = tap0 + tw;
      const int r = tp / p.S, s = tp - r * p.S;         // wave-uniform
      const int iy = iy0 + r, ix = ix0 + s;
      const bool vok = in & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      vbase[tw] = vok ? (unsigned int)((n * p.H + iy) * p.W + ix) * (unsigned int)(p.vp * 2) : OOB_OFFSET;
    }
    char* ub = smem + buf * STAGE + wave * (GA * 1024);          // [pixel group][chan group][1 KB]
    char* vb = smem + buf * STAGE + UBYTES + wave * (GB * 1024);
    unsigned int ou[GA], ov[GB];
#pragma unroll
    for (int g = 0; g < GA; g++) ou[g] = (ubase | uoffc[g]) >= OOB_OFFSET ? OOB_OFFSET : ubase + uoffc[g];
#pragma unroll
    for (int g = 0; g < GB; g++)
      ov[g] = (vbase[g / GPT] | voffc[g]) >= OOB_OFFSET ? OOB_OFFSET : vbase[g / GPT] + voffc[g];
#pragma unroll
    for (int g = 0; g < GA; g++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lptr_t)(ub + g * 1024), 16, ou[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GB; g++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(vb + g * 1024), 16, ov[g], 0, 0, 0);
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
  // transposing fragment read: block row q (pixel), 4 channels pp of a 16-channel half g&1;
  // lane's pixel inside the 16-pixel group = 8 * (g >> 1) + q; the second read is 4 pixels on
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int frag = (8 * (g4 >> 1) + q4) * 64 + (2 * (g4 & 1) + (pp >> 1)) * 16 + (pp & 1) * 8;
  const int aofs = (arow0 / 32) * 1024 + frag, bofs = (bcol0 / 32) * 1024 + frag;
  typedef __attribute__((address_space(3))) s16x4* tr_ptr;
  auto trf = [&](const char* base) -> s16x8 {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(base));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(base + 256));
    s16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
  };

  if (nstages > 0) gissue(0, 0);
  __syncthreads();
  int cur = 0;
  for (int st = 0; st < nstages; st++) {
    if (st + 1 < nstages) gissue(st + 1, cur ^ 1);
    const char* ub = smem + cur * STAGE + aofs;
    const char* vb = smem + cur * STAGE + UBYTES + bofs;
    // register double buffer of the fragments of one pixel group (16 k-values); the fences keep
    // hipcc from sinking the transposing reads next to their MFMAs
    s16x8 a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; i++) a[0][i] = trf(ub + i * 1024);
#pragma unroll
    for (int j = 0; j < TN; j++) b[0][j] = trf(vb + j * 1024);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {                    // pixel group kk = 16 k-values
      const int cb = kk & 1, nb = cb ^ 1;
      if (kk + 1 < 4) {
#pragma unroll
        for (int i = 0; i < TM; i++) a[nb][i] = trf(ub + (kk + 1) * (GA * 1024) + i * 1024);
#pragma unroll
        for (int j = 0; j < TN; j++) b[nb][j] = trf(vb + (kk + 1) * (GB * 1024) + j * 1024);
      }
#ifndef MSML_NO_SCHED_FENCE
      __builtin_amdgcn_sched_barrier(0);
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
    __syncthreads();
    cur ^= 1;
  }

  const int h = lane >> 5, c32 = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col = bcol0 + 32 * j + c32;             // column inside the BB-wide tile
      const int tap = tap0 + col / (GPT * 32);
      const int b = b0 + col % (GPT * 32);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int a = a0 + arow0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (p.direct) {
          if (a < p.A && b < p.Breal && tap < taps) {
            float* dwl = p.dw[layer];
            const long o = ((long)a * p.Btot + p.boff + b) * taps + tap;
            dwl[o] = p.accumulate ? dwl[o] + acc[i][j][e] : acc[i][j][e];
          }
        } else if (a < p.arows && b < p.vp && tap < taps) {
          p.ws[(((long)blockIdx.z * p.arows + a) * taps + tap) * p.vp + b] = acc[i][j][e];
        }
      }
    }
#endif
}



// Called from msml_conv_wgrad[_group] (conv_wgrad.hip) for bf16; chunk is a multiple of 64.  group layers of one shape
// share the launch (grid.z = group x splits; slabs at ws[layer * splits + split], or, with splits == 1, every layer's
// tile straight into its dW).  Returns false when the tensors are too large for 32-bit buffer offsets.
bool msml_wgrad_fast_launch_group(const void* const* u, int up, const void* const* v, int vp, float* ws, int N, int H,
                                  int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w, int ba, int bb,
                                  int ntw, int group, int splits, int chunk, hipStream_t st, float* const* dw_direct,
                                  int A, int Breal, int Btot, int boff, int accumulate) {
  const long ub = (long)N * P * Q * up * 2, vb = (long)N * H * W * vp * 2;
  if (ub >= 0x7fffff00L || vb >= 0x7fffff00L || group < 1 || group > WF_MAXGROUP) return false;
  WgradFastArgs a;
  for (int i = 0; i < WF_MAXGROUP; i++) {
    a.u[i] = (const unsigned short*)u[i < group ? i : 0];
    a.v[i] = (const unsigned short*)v[i < group ? i : 0];
    a.dw[i] = dw_direct ? dw_direct[i < group ? i : 0] : nullptr;
  }
  a.up = up; a.u_bytes = (unsigned int)ub;
  a.vp = vp; a.v_bytes = (unsigned int)vb;
  a.zper = splits;
  a.N = N; a.H = H; a.W = W; a.P = P; a.Q = Q; a.R = R; a.S = S;
  a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.ws = ws; a.arows = up;
  a.Mpix = (long)N * P * Q;
  a.chunk = chunk;
  a.rcp_pq = 1.0f / (float)(P * Q);
  a.rcp_q = 1.0f / (float)Q;
  a.direct = (splits == 1 && dw_direct) ? 1 : 0;
  a.A = A; a.Breal = Breal; a.Btot = Btot; a.boff = boff; a.accumulate = accumulate;
  const int gz = group * splits;
#define WF(BA_, BB_, NTW_) k_wgrad_fast<BA_, BB_, NTW_><<<grid, dim3(256), 2 * 64 * (BA_ + BB_) * 2, st>>>(a)
  if (ntw == 3) {                                       // narrow V: 3 taps per workgroup
    dim3 grid(cdiv(up, ba), cdiv(R * S, 3) * cdiv(vp, 64), gz);
    if (ba == 128) WF(128, 192, 3);
    else WF(64, 192, 3);
    return true;
  }
  dim3 grid(cdiv(up, ba), cdiv(vp, bb) * R * S, gz);
  if (ba == 128 && bb == 128) WF(128, 128, 1);
  else if (ba == 128) WF(128, 64, 1);
  else if (bb == 128) WF(64, 128, 1);
  else WF(64, 64, 1);
#undef WF
  return true;
}

bool msml_wgrad_fast_launch(const void* u, int up, const void* v, int vp, float* ws, int N, int H, int W,
                            int P, int Q, int R, int S, int stride, int pad_h, int pad_w, int ba, int bb,
                            int ntw, int splits, int chunk, hipStream_t st, float* dw_direct, int A, int Breal,
                            int Btot, int boff, int accumulate) {
  return msml_wgrad_fast_launch_group(&u, up, &v, vp, ws, N, H, W, P, Q, R, S, stride, pad_h, pad_w, ba, bb, ntw, 1,
                                      splits, chunk, st, dw_direct ? &dw_direct : nullptr, A, Breal, Btot, boff,
                                      accumulate);
}
