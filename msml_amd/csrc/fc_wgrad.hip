// Weight gradient of flatten + Linear(C*H*W -> E) on a small NHWC map (backbones/frb/iresnet.py:230-232, the
// 25 088 -> 512 fc of every IResNet): the "window" case of msml_conv_wgrad (R = H, S = W, one output pixel),
//     dW[e][c][tap] = sum_n dY[n][e] * X[n][tap][c]          (dW stored (E, C, H, W) like the parameter).
//
// The generic kernel treats the 49 taps as 49 separate [E x C] GEMMs whose results land 49 floats apart in dW:
// 4-byte stores at a 196-byte stride, 179 us for 6.6 GFLOP (37 TFLOP/s).  Here a workgroup owns 32 e x 32 c and
// ALL taps (8 waves x up to 7 taps x one 32x32 accumulator tile), so that its piece of every dW row is one
// contiguous run of 32 * taps floats:
//   * K = the batch: per k-step of 16 images the dY block [16][32 e] and the tap blocks X[16][tap][32 c] (1 KB each)
//     go to LDS by LDS-DMA (three-stage ring, counted vmcnt + raw barrier as in wgrad_n32.hip) and are read with
//     the transposing fragment read of wgrad_fast.hip;
//   * no split-K: 16 x 16 tiles = 256 workgroups for 512 x 512, a fixed summation order over the batch;
//   * epilogue: accumulators -> LDS as [16 e][32 c * taps] f32 (two halves), then 16-byte row stores (read-modify-
//     write when the caller accumulates into the flat gradient arena).
// X is re-read once per e-tile (16 x 12.8 MB from L2): the kernel is bound by the L2 -> LDS fill, ~10x the MFMA time.
#include <stdlib.h>

#include <mutex>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) void* lptr_t;

#define FCW_OOB 0x78000000u
#define FCW_NISS 7                                   // DMA wave-instructions per wave and stage (8 x 7 >= taps + 1)
#define FCW_MAX_TAPS 51                              // 3 stages x (taps + 2) KB <= 160 KB

struct FcWgradArgs {
  const unsigned short* u; unsigned int u_bytes;     // dY [N][up]
  const unsigned short* v; unsigned int v_bytes;     // X  [N][T][vp]
  int N, T, up, vp;
  float* dw;                                         // [up][Btot * T], this launch fills columns (boff + c) * T + tap
  long row_stride;                                   // Btot * T
  int boff, accumulate;
};

__global__ void __launch_bounds__(512) k_fc_wgrad(const FcWgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int c0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
  const int T = p.T, stage_bytes = (T + 2) * 1024;   // dY block, T tap blocks, 1 KB that absorbs surplus DMAs
  const int nsteps = (p.N + 15) >> 4;

  __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, (int)p.u_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)p.v, 0, (int)p.v_bytes, 0x00020000);

  const int lp = lane >> 2, lc = lane & 3;           // DMA slot: image lane / 4 of the k-step, 16-B chunk lane % 4
  auto issue = [&](int step, int buf) {
    char* sb = smem + buf * stage_bytes;
    const int n = step * 16 + lp;
    const bool nok = n < p.N;
#pragma unroll
    for (int i = 0; i < FCW_NISS; i++) {
      const int blk = wave + 8 * i;
      if (blk > T) {                                 // surplus: zeros into the spare KB (keeps vmcnt counts uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lptr_t)(sb + (T + 1) * 1024), 16, FCW_OOB, 0, 0, 0);
      } else if (blk == 0) {
        const unsigned int off = nok ? (unsigned int)(n * p.up + e0) * 2u + (unsigned int)lc * 16u : FCW_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (lptr_t)sb, 16, off, 0, 0, 0);
      } else {
        const unsigned int off = nok ? (unsigned int)((n * T + blk - 1) * p.vp + c0) * 2u + (unsigned int)lc * 16u : FCW_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(sb + blk * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  f32x16 acc[FCW_NISS];
#pragma unroll
  for (int k = 0; k < FCW_NISS; k++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[k][e] = 0.f;

  // transposing fragment read (wgrad_fast.hip / wgrad_n32.hip) on [16 images][64 B] blocks
  const int g4 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int px = 8 * (g4 >> 1) + q4;
  const int chan = (2 * (g4 & 1) + (pp >> 1)) * 16 + (pp & 1) * 8;
  const int aofs = px * 64 + chan;
  typedef __attribute__((address_space(3))) s16x4* tr_ptr;
  auto tr2 = [&](const char* lo) -> s16x8 {
    s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(lo));
    s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)(lo + 256));
    s16x8 o;
    o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = l[3];
    o[4] = h[0]; o[5] = h[1]; o[6] = h[2]; o[7] = h[3];
    return o;
  };
  int ntaps = 0;                                     // taps wave, wave + 8, ... < T
#pragma unroll
  for (int k = 0; k < FCW_NISS; k++) ntaps += (wave + 8 * k < T) ? 1 : 0;

  issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  int cur = 0, nxt = 2;
  for (int step = 0; step < nsteps; step++) {
    if (step + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FCW_NISS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (step + 2 < nsteps) issue(step + 2, nxt);
    const char* sb = smem + cur * stage_bytes + aofs;
    const s16x8 fa = tr2(sb);
#pragma unroll
    for (int k = 0; k < FCW_NISS; k++)
      if (k < ntaps) {
        const s16x8 fb = tr2(sb + (1 + wave + 8 * k) * 1024);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                         acc[k], 0, 0, 0);
      }
    cur = cur == 2 ? 0 : cur + 1;
    nxt = nxt == 2 ? 0 : nxt + 1;
  }
  __syncthreads();

  // epilogue: two halves of 16 e-rows through LDS [16][32 * T] f32, then contiguous 16-byte row pieces
  float* tile = reinterpret_cast<float*>(smem);
  MSML_LDS_REGION(smem, 3 * stage_bytes);
  MSML_LDS_REGION(tile, 16 * 32 * T * 4);
  const int RL = 32 * T, RL4 = RL >> 2;
  const int h = lane >> 5, b = lane & 31;
  for (int hf = 0; hf < 2; hf++) {
#pragma unroll
    for (int k = 0; k < FCW_NISS; k++) {
      if (k >= ntaps) continue;
      const int tp = wave + 8 * k;
#pragma unroll
      for (int e8 = 0; e8 < 8; e8++) {
        const int arow = (e8 & 3) + 8 * (e8 >> 2) + 4 * h;          // accumulator row (e8 + 8 hf) minus 16 hf
        tile[arow * RL + b * T + tp] = hf ? acc[k][8 + e8] : acc[k][e8];
      }
    }
    __syncthreads();
    for (int idx = t; idx < 16 * RL4; idx += 512) {
      const int row = idx / RL4, c4 = idx - row * RL4;
      float* g = p.dw + (long)(e0 + 16 * hf + row) * p.row_stride + (long)(p.boff + c0) * T + c4 * 4;
      f32x4 o = *reinterpret_cast<const f32x4*>(tile + row * RL + c4 * 4);
      if (p.accumulate) {
        const f32x4 old = *reinterpret_cast<const f32x4*>(g);
        o[0] += old[0]; o[1] += old[1]; o[2] += old[2]; o[3] += old[3];
      }
      *reinterpret_cast<f32x4*>(g) = o;
    }
    __syncthreads();
  }
#endif
}

// 1 if the launch was taken (the caller falls back to the generic path otherwise)
int msml_fc_wgrad_launch(const void* u, int up, const void* v, int vp, float* dw, int A, int Breal, int Btot, int boff,
                         int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h, int pad_w,
                         int accumulate, hipStream_t st) {
  static const bool off = getenv("MSML_NO_FC_WGRAD") != nullptr;
  const int T = R * S;
  if (off || P != 1 || Q != 1 || R != H || S != W || pad_h != 0 || pad_w != 0 || stride != 1) return 0;
  if (T < 2 || T > FCW_MAX_TAPS || up % 32 || vp % 32 || A != up || Breal != vp || N < 16) return 0;
  if (boff % 4 || ((long)Btot * T) % 4 || ((size_t)dw % 16)) return 0;
  if ((long)N * T * vp * 2 >= 0x70000000L || (long)N * up * 2 >= 0x70000000L) return 0;
  FcWgradArgs a;
  a.u = (const unsigned short*)u; a.u_bytes = (unsigned int)((long)N * up * 2);
  a.v = (const unsigned short*)v; a.v_bytes = (unsigned int)((long)N * T * vp * 2);
  a.N = N; a.T = T; a.up = up; a.vp = vp;
  a.dw = dw; a.row_stride = (long)Btot * T; a.boff = boff; a.accumulate = accumulate;
  const int lds = 3 * (T + 2) * 1024;                // >= the 16 x 32 T x 4 B epilogue tile for T >= 2
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fc_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize,
                              3 * (FCW_MAX_TAPS + 2) * 1024);
  });
  k_fc_wgrad<<<dim3(vp / 32, up / 32), 512, lds, st>>>(a);
  return 1;
}
