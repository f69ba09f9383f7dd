// Line convolutions of the OSB's Global-Convolution modules (backbones/osb/unet.py:16-38: 7x1 and 1x7, stride 1,
// biased, 64 -> 18 and 18 -> 18 channels, stored 64 / 32) and their backward-data convs, for bf16 NHWC tensors.
//
// These layers are HBM-bound (gcm5 conv_l1: 103 MB in, 51 MB out for 13 GFLOP), but the im2col kernel gathers the
// input once PER TAP from L2 (7 passes) and a 256-pixel tile of 32 output channels leaves one wave in four busy:
// 55-86 us per launch against ~25 us of streaming time.  Here a tile is a full LINE along the conv axis times B
// positions across it (56 x 4, 28 x 8, 14 x 16, 7 x 32 = 224 pixels): the tile's input region is the tile itself plus
// three zero pixels at either end of every line -- NO halo is re-read -- and goes to LDS once (LDS-DMA, double
// buffered, persistent workgroups keep the 14-28 KB of weights resident).  Pixels are stored line-major, p = a * B + b,
// so a tap is the constant shift (t - 3) * B of the pixel index for every lane; D = W_frag x X_frag (output channels in
// the accumulator rows) lets a lane pair store 32 contiguous channels of a pixel with v_permlane32_swap.  7 waves
// carry one 32-pixel block each (28 MFMAs per tile at 64 input channels: ~5 % of the fill time).
#include <stdlib.h>

#include <mutex>

#include "common.h"

typedef __attribute__((address_space(3))) void* lptr_t;

#define CL_OOB 0x78000000u

struct ConvLineArgs {
  const unsigned short* in; unsigned int in_bytes;      // [N][L][L][CIN]
  const unsigned short* wp; unsigned int w_bytes;       // packed [COUT rows][7 * CIN]
  unsigned short* out;                                  // [N][L][L][COUT]
  const float* bias;                                    // [COUT] or null
  const float* scale;                                   // [COUT] or null: out = acc * scale + bias
  const unsigned short* residual;                       // [N][L][L][COUT] or null: added to the ROUNDED output, as
                                                        // msml_conv2d_fused's general kernel does (conv_fast.hip)
  int N, L, B, tiles_per_img, ntiles;
  int vertical;                                         // 1: taps along y (7x1), 0: along x (1x7)
  int flip;                                             // 1: transposed gather (backward-data): tap t reads a + 3 - t
};

template <int CIN, int COUT>
__global__ void __launch_bounds__(512) k_conv_line(const ConvLineArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CINB = CIN * 2, CH = CIN / 8;           // bytes / 16-B chunks per pixel
  constexpr int PPB = 1024 / CINB;                      // pixels per 1-KB DMA block (8 or 16)
  constexpr int XPIX = 288;                             // region pixels per stage: (L + 6) * B = 248 (L 56) / 272 (L 28), padded
  constexpr int XB = XPIX * CINB, WB = 7 * COUT * CINB;
  constexpr int NCO = COUT / 32, KK = CIN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem;                                      // [tap][co][CIN], chunks swizzled by the row
  char* Xs = smem + WB;                                 // [2][XPIX][CIN], chunks swizzled by the pixel
  MSML_LDS_REGION(Ws, WB);
  MSML_LDS_REGION(Xs, 2 * XB);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int L = p.L, B = p.B, npix = L * B, nreg = (L + 6) * B;

  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.w_bytes, 0x00020000);

  // swizzle key of a row / pixel (the 16 rows a ds_read_b128 lane group touches must land on 16 different
  // 16-B bank slots): 128-B rows (CIN 64) alternate two half-banks, so the key counts row pairs (conv_fast.hip's
  // swz128); 64-B rows (CIN 32) cycle through four quarter-banks, so the key counts groups of four rows
  auto key = [&](int row) -> int { return CIN == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  // weights: 7 * COUT rows of CINB bytes; a DMA block covers PPB rows
  const int lrow = lane / CH, lslot = lane % CH;
  for (int blk = wave; blk < 7 * COUT / PPB; blk += 8) {
    const int row = blk * PPB + lrow;                   // = tap * COUT + co
    const int tap = row / COUT, co = row - tap * COUT;
    const unsigned int off = (unsigned int)((co * 7 + tap) * CIN + ((lslot ^ key(row)) * 8)) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(Ws + blk * 1024), 16, off, 0, 0, 0);
  }

  // region pixel q (0 .. nreg): line position a = q / B - 3 (zero outside [0, L)), cross position b = q % B
  auto issue = [&](int tile, int buf) {
    const int n = tile / p.tiles_per_img, b0 = (tile - n * p.tiles_per_img) * B;
    char* xb = Xs + buf * XB;
    for (int blk = wave; blk * PPB < nreg; blk += 8) {
      const int q = blk * PPB + lrow;
      const int a = q / B - 3, b = b0 + (q - (q / B) * B);
      const bool ok = (q < nreg) & ((unsigned)a < (unsigned)L) & (b < L);
      const int y = p.vertical ? a : b, x = p.vertical ? b : a;
      const unsigned int off = ok ? (unsigned int)(((n * L + y) * L + x) * CIN + ((lslot ^ key(q)) * 8)) * 2u : CL_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(xb + blk * 1024), 16, off, 0, 0, 0);
    }
  };

  const int r32 = lane & 31, h = lane >> 5;
  const int nblk = npix / 32;                           // 7 pixel blocks per tile
  float bv[NCO][4][4], sv[NCO][4][4];
#pragma unroll
  for (int i = 0; i < NCO; i++)
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        bv[i][g][j] = p.bias ? p.bias[i * 32 + 8 * g + 4 * h + j] : 0.f;
        sv[i][g][j] = p.scale ? p.scale[i * 32 + 8 * g + 4 * h + j] : 1.f;
      }

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue(tile, 0);
  __syncthreads();                                      // weights + first region landed (the fence drains vmcnt)
  int cur = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    if (nxt < p.ntiles) issue(nxt, cur ^ 1);
    if (wave < nblk) {
      const char* xb = Xs + cur * XB;
      const int pq = wave * 32 + r32;                   // this lane's output pixel (line-major)
      f32x16 acc[NCO];
#pragma unroll
      for (int i = 0; i < NCO; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
#pragma unroll
      for (int tp = 0; tp < 7; tp++) {
        const int q = pq + (p.flip ? (6 - tp) : tp) * B;                // region pixel of tap tp (a + tp - 3, + 3 rows of padding)
        const char* xq = xb + q * CINB;
        const int kq = key(q);
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
          const u32x4 bfrag = *reinterpret_cast<const u32x4*>(xq + (((kk * 2 + h) ^ kq) << 4));
#pragma unroll
          for (int i = 0; i < NCO; i++) {
            const int row = tp * COUT + i * 32 + r32;
            const u32x4 afrag = *reinterpret_cast<const u32x4*>(Ws + row * CINB + (((kk * 2 + h) ^ key(row)) << 4));
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afrag),
                                                             __builtin_bit_cast(bf16x8, bfrag), acc[i], 0, 0, 0);
          }
        }
      }
      // epilogue: lane (pixel r32, half h) holds channels 8 g + 4 h + j of each 32-channel block; after the
      // swaps h = 0 holds channels 0-7 and 16-23, h = 1 holds 8-15 and 24-31 -> two 16-B stores per lane
      const int n = tile / p.tiles_per_img, b0 = (tile - n * p.tiles_per_img) * B;
      const int a = pq / B, b = b0 + (pq - (pq / B) * B);
      const bool valid = b < L;
      const int y = p.vertical ? a : b, x = p.vertical ? b : a;
      unsigned short* o = p.out + ((long)(n * L + y) * L + x) * COUT + 8 * h;
#pragma unroll
      for (int i = 0; i < NCO; i++) {
        u32x2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = p.scale ? acc[i][g * 4 + j] * sv[i][g][j] + bv[i][g][j] : acc[i][g * 4 + j] + bv[i][g][j];
          pk[g][0] = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
          pk[g][1] = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
        }
        u32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 2; e++) {
          auto r01 = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
          auto r23 = __builtin_amdgcn_permlane32_swap(pk[2][e], pk[3][e], false, false);
          lo[e] = r01[0]; lo[2 + e] = r01[1];
          hi[e] = r23[0]; hi[2 + e] = r23[1];
        }
        if (valid && p.residual) {                      // (uniform branch; lane: channels 8 h .. and 16 + 8 h .. of block i)
          const unsigned short* rp = p.residual + (o - p.out) + i * 32;
          const u32x4 r0 = *reinterpret_cast<const u32x4*>(rp), r1 = *reinterpret_cast<const u32x4*>(rp + 16);
          Vec8 a0 = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&lo));
          Vec8 a1 = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&hi));
          const Vec8 b0 = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&r0));
          const Vec8 b1 = load8<unsigned short>(reinterpret_cast<const unsigned short*>(&r1));
#pragma unroll
          for (int q = 0; q < 8; q++) {
            a0.v[q] += b0.v[q];
            a1.v[q] += b1.v[q];
          }
          store8<unsigned short>(reinterpret_cast<unsigned short*>(&lo), a0);
          store8<unsigned short>(reinterpret_cast<unsigned short*>(&hi), a1);
        }
        if (valid) {
          *reinterpret_cast<u32x4*>(o + i * 32) = lo;
          *reinterpret_cast<u32x4*>(o + i * 32 + 16) = hi;
        }
      }
    }
    __syncthreads();                                    // next region landed, everyone done with this one
    cur ^= 1;
  }
#endif
}

bool msml_conv_line_applies(int c0p, int coutp, int N, int H, int W, int P, int Q, int R, int S, int stride, int pad_h,
                            int pad_w) {
  static const bool off = getenv("MSML_NO_LINE_CONV") != nullptr;
  if (off) return false;
  if (!((R == 7 && S == 1 && pad_h == 3 && pad_w == 0) || (R == 1 && S == 7 && pad_h == 0 && pad_w == 3))) return false;
  if (stride != 1 || H != W || P != H || Q != W) return false;
  if (!(H == 56 || H == 28)) return false;            // 224 / H lines per tile; smaller maps stay on the generic kernel
  if (!((c0p == 64 && coutp == 32) || (c0p == 32 && coutp == 32) || (c0p == 32 && coutp == 64))) return false;
  if ((long)N * H * W * c0p * 2 >= 0x70000000L) return false;
  return true;
}

template <int CIN, int COUT>
static void cl_launch(const ConvLineArgs& a, hipStream_t st) {
  const size_t lds = (size_t)7 * COUT * CIN * 2 + 2 * (size_t)288 * CIN * 2;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_line<CIN, COUT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int per_cu = lds <= 76 * 1024 ? 2 : 1;
  int grid = cus * per_cu;
  if (grid > a.ntiles) grid = a.ntiles;
  k_conv_line<CIN, COUT><<<dim3(grid), dim3(512), lds, st>>>(a);
}

bool msml_conv_line_dispatch(const void* in0, int c0p, const void* wp, int kop, const float* bias, void* out, int coutp,
                             int N, int H, int W, int R, int S, int transposed, hipStream_t st, const float* scale,
                             const void* residual) {
  ConvLineArgs a;
  a.scale = scale; a.residual = (const unsigned short*)residual;
  a.in = (const unsigned short*)in0; a.in_bytes = (unsigned int)((long)N * H * W * c0p * 2);
  a.wp = (const unsigned short*)wp; a.w_bytes = (unsigned int)((long)kop * 7 * c0p * 2);
  a.out = (unsigned short*)out; a.bias = bias;
  a.N = N; a.L = H; a.B = 224 / H;
  a.tiles_per_img = cdiv(H, a.B);
  a.ntiles = N * a.tiles_per_img;
  a.vertical = R == 7 ? 1 : 0;
  a.flip = transposed ? 1 : 0;
  if (c0p == 64 && coutp == 32) cl_launch<64, 32>(a, st);
  else if (c0p == 32 && coutp == 32) cl_launch<32, 32>(a, st);
  else if (c0p == 32 && coutp == 64) cl_launch<32, 64>(a, st);
  else return false;
  return true;
}
