// Box calibration probe for bench.py: a register-resident bf16 MFMA loop on random operands.
// MI355X devices hold different clocks under the same MFMA load (MI355X_MICROARCH.md, DVFS give-back item 5: 12 %
// between devices on a loop with no memory traffic), so a round's kernel gain cannot be read off two bench lines of
// two boxes; the probe's TFLOP/s measured in the SAME bench run is the MFMA-side yardstick the line is normalised by
// (the HBM-side yardstick is a device copy issued by bench.py itself).  Not part of the hot path.
#include "common.h"

// Every wave keeps 4 A and 4 B fragments (random bf16 from `seed`, 8 KB) and 16 independent 16 x 16 accumulators:
// `iters` rounds of 16 v_mfma_f32_16x16x32_bf16 -- the MFMA shape of the dominant conv kernels (conv_halo.hip, M16).
// 256 threads = one wave per SIMD; the sum of the accumulators leaves through `out` so nothing is dead code.
__global__ void __launch_bounds__(256) k_probe_mfma(const unsigned short* seed, float* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int t = threadIdx.x;
  u32x4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[i] = *reinterpret_cast<const u32x4*>(seed + ((t * 4 + i) & 511) * 8);
    b[i] = *reinterpret_cast<const u32x4*>(seed + ((t * 4 + i + 257) & 511) * 8);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (inline asm with VGPR accumulators: hipcc's own allocation of this loop shuttles a quarter of the accumulators
  // through v_accvgpr moves every round; each accumulator is reused 16 MFMAs later, far beyond the dependent-issue distance)
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) s += acc[i][j];
  // (operands are O(1) random values and the sums stay finite in f32 for any iters bench.py uses; the value is unused)
  out[(long)blockIdx.x * 256 + t] = s[0] + s[1] + s[2] + s[3];
#endif
}

// The same with every operand re-read from LDS, in the mix of the halo-tile conv's main loop (conv_halo.hip, 16x16x32
// build): 512 threads = two waves per SIMD; per round 7 image fragments + 2 weight fragments (ds_read_b128, conflict-free
// rows) feed 14 MFMAs, the next round's fragments requested before the current MFMAs.  64 KB of random bf16 in LDS.
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_probe_mfma_lds(const unsigned short* seed, float* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 4096; i += 512)                  // 4096 x 16 B; the 8 KB seed repeats with a per-copy rotation
    *reinterpret_cast<u32x4*>(lds + i * 16) = *reinterpret_cast<const u32x4*>(seed + (((i * 8) + (i >> 9) * 24) & 4095 & ~7));
  __syncthreads();
  // lane reads row (lane & 15) + 16 j of its wave's 8 KB window, chunk (lane >> 4) ^ (row & 7): the conv's fragment map
  const char* base = lds + wave * 8192;
  const int l16 = lane & 15, q16 = lane >> 4;
  int off[2];
#pragma unroll
  for (int w = 0; w < 2; w++) off[w] = l16 * 128 + (((4 * w + q16) ^ (l16 & 7)) << 4);
  f32x4 acc[7][2];
#pragma unroll
  for (int i = 0; i < 7; i++)
#pragma unroll
    for (int g = 0; g < 2; g++) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[2][7], b[2][2];
  auto frags = [&](int it, u32x4 (&a_)[7], u32x4 (&b_)[2]) {
    const int w = it & 1, sh = ((it >> 1) & 3) * 2048;   // walk four 2 KB sub-windows
#pragma unroll
    for (int j = 0; j < 3; j++) a_[j] = *reinterpret_cast<const u32x4*>(base + ((sh + j * 2048 + off[w]) & 8191));
#pragma unroll
    for (int j = 3; j < 7; j++) a_[j] = *reinterpret_cast<const u32x4*>(lds + ((wave * 8192 + 8192 + sh + (j - 3) * 2048 + off[w]) & 65535));
#pragma unroll
    for (int g = 0; g < 2; g++) b_[g] = *reinterpret_cast<const u32x4*>(lds + ((wave * 8192 + 32768 + g * 2048 + sh + off[w]) & 65535));
  };
  frags(0, a[0], b[0]);
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      frags(it + h + 1, a[h ^ 1], b[h ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 7; j++)
#pragma unroll
        for (int g = 0; g < 2; g++)
          acc[j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[h][g]), __builtin_bit_cast(bf16x8, a[h][j]),
                                                              acc[j][g], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 7; j++)
#pragma unroll
    for (int g = 0; g < 2; g++) s += acc[j][g];
  out[(long)blockIdx.x * 512 + t] = s[0] + s[1] + s[2] + s[3];
#endif
}

extern "C" int msml_probe_mfma_lds(const void* seed, float* out, int wgs, int iters, void* stream) {
  MSML_CHECK(seed && out && wgs > 0 && wgs <= 65536 && iters > 0 && iters % 2 == 0, MSML_ERR_SHAPE,
             "msml_probe_mfma_lds: bad arguments");
  k_probe_mfma_lds<<<dim3(wgs), dim3(512), 0, (hipStream_t)stream>>>((const unsigned short*)seed, out, iters);
  MSML_LAUNCH_OK("msml_probe_mfma_lds");
  return MSML_OK;
}

extern "C" int msml_probe_mfma(const void* seed, float* out, int wgs, int iters, void* stream) {
  MSML_CHECK(seed && out && wgs > 0 && wgs <= 65536 && iters > 0, MSML_ERR_SHAPE, "msml_probe_mfma: bad arguments");
  k_probe_mfma<<<dim3(wgs), dim3(256), 0, (hipStream_t)stream>>>((const unsigned short*)seed, out, iters);
  MSML_LAUNCH_OK("msml_probe_mfma");
  return MSML_OK;
}
