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

extern "C" int msml_probe_mfma(const void* seed, float* out, int wgs, int iters, void* stream) {
  MSML_CHECK(seed && out && wgs > 0 && wgs <= 65536 && iters > 0, MSML_ERR_SHAPE, "msml_probe_mfma: bad arguments");
  k_probe_mfma<<<dim3(wgs), dim3(256), 0, (hipStream_t)stream>>>((const unsigned short*)seed, out, iters);
  MSML_LAUNCH_OK("msml_probe_mfma");
  return MSML_OK;
}
