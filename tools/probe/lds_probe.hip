// LDS bank-conflict probe for ds_read_b128 on gfx950: which lane -> address maps are conflict-free?
// One wave reads 16 B per lane from LDS with a host-supplied byte offset per lane; cycles per read (s_memtime) for
// each pattern.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_probe.hip -o build/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ void probe(const int* offs, int npat, int iters, long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = i;
  __syncthreads();
  for (int pat = 0; pat < npat; pat++) {
    const int off = offs[pat * 64 + (threadIdx.x & 63)];
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
      u32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<volatile u32x4*>(smem + ((off + (u & 7) * 8192 + (u >> 3) * 4096) & 65535));
#pragma unroll
      for (int u = 0; u < 16; u++) acc += v[u];
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[pat] = t1 - t0;
    if (acc[0] == 0x12345678u) out[npat + pat] = acc[1];     // keep the loads alive
    __syncthreads();
  }
}

int main() {
  struct Pat { std::string name; int off[64]; };
  std::vector<Pat> pats;
  auto add = [&](const char* name, auto f) { Pat p; p.name = name; for (int l = 0; l < 64; l++) p.off[l] = f(l); pats.push_back(p); };
  add("linear lane*16 (conflict-free reference)", [](int l) { return l * 16; });
  add("stride 128 B no swizzle (worst)", [](int l) { return l * 128; });
  // fragment maps of the MFMA shapes: 32x32x16: row = l & 31, chunk = c0 + (l >> 5); 16x16x32: row = l & 15, chunk = l >> 4;
  // tap shift s moves the rows; key = swizzle XORed onto the 16-B chunk index
  struct Key { const char* name; int (*f)(int); };
  Key keys[] = {
      {"(row>>1)&7", [](int r) { return (r >> 1) & 7; }},
      {"r1<<2|r2<<1", [](int r) { return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1); }},
      {"((row>>1)&3)<<1", [](int r) { return ((r >> 1) & 3) << 1; }},
      {"row&7", [](int r) { return r & 7; }},
      {"(row>>1)&3 | r3<<2", [](int r) { return ((r >> 1) & 3) | (((r >> 3) & 1) << 2); }},
      {"none", [](int) { return 0; }},
  };
  for (auto& k : keys)
    for (int shape = 0; shape < 2; shape++)
      for (int sft = 0; sft < 3; sft++) {
        Pat p; char nm[96]; snprintf(nm, 96, "%s map, tap shift %d, key %s", shape ? "16x16x32" : "32x32x16", sft, k.name); p.name = nm;
        for (int l = 0; l < 64; l++) {
          const int row = (shape ? (l & 15) : (l & 31)) + sft, c = shape ? (l >> 4) : 2 + (l >> 5);
          p.off[l] = row * 128 + ((c ^ k.f(row)) << 4);
        }
        pats.push_back(p);
      }
  int np = (int)pats.size();
  std::vector<int> h(np * 64);
  for (int i = 0; i < np; i++) for (int l = 0; l < 64; l++) h[i * 64 + l] = pats[i].off[l];
  int* d; long long* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 2 * np * 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 2000;
  for (int rep = 0; rep < 2; rep++) probe<<<1, 1024, 65536>>>(d, np, iters, o);
  hipDeviceSynchronize();
  std::vector<long long> r(2 * np);
  hipMemcpy(r.data(), o, 2 * np * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < np; i++) printf("%-58s %.3f milli-ticks per wave-level ds_read_b128 (16 waves x 16 reads in flight)\n", pats[i].name.c_str(), (double)r[i] / (iters * 16.0 * 16.0) * 1000.0);
  return 0;
}
