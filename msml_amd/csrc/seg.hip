// OSB tail and segmentation loss.
//
//  * DAP (backbones/osb/unet.py:158-161,223): PixelShuffle(3) -> AvgPool2d(3) == mean over
//    each group of 9 consecutive channels.  Fused with the NHWC->NCHW f32
//    conversion of final_seg and with the occlusion-mask index (argmax over the 2 classes,
//    ties -> class 0: train.py:357, eval/qeval_mxnet.py:347).
//  * StructureConsensuLossFunction(alpha, beta, 'idx', 'idx') with blobs == target == msk
//    (tricks/consensus_loss.py:65-167, called at train.py:258): forward value and d loss / d logit
//    in three small kernels (per-image blob sums -> scalar loss + coefficients -> per-pixel grad).
#include "common.h"

template <typename T>
__global__ void __launch_bounds__(256) k_dap_fwd(const T* __restrict__ x, float* __restrict__ seg,
                                                 unsigned char* __restrict__ mask, long npix, int HW,
                                                 int Cp) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix;
       i += (long)gridDim.x * blockDim.x) {
    const T* p = x + i * Cp;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      s0 += load1<T>(p + j);
      s1 += load1<T>(p + 9 + j);
    }
    s0 = s0 / 9.f;
    s1 = s1 / 9.f;
    long n = i / HW, hw = i % HW;
    seg[(n * 2 + 0) * HW + hw] = s0;
    seg[(n * 2 + 1) * HW + hw] = s1;
    if (mask) mask[i] = s1 > s0 ? 1 : 0;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_dap_bwd(const float* __restrict__ dseg, T* __restrict__ dx,
                                                 long npix, int HW, int Cp) {
  const int C8 = Cp / 8;
  long total = npix * C8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    long pix = i / C8;
    int c0 = (int)(i % C8) * 8;
    long n = pix / HW, hw = pix % HW;
    float g0 = dseg[(n * 2 + 0) * HW + hw] / 9.f;
    float g1 = dseg[(n * 2 + 1) * HW + hw] / 9.f;
    Vec8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int c = c0 + j;
      v.v[j] = c < 9 ? g0 : (c < 18 ? g1 : 0.f);
    }
    store8<T>(dx + pix * Cp + c0, v);
  }
}

extern "C" int msml_dap_fwd(const void* x, float* seg, unsigned char* mask, int N, int H, int W,
                            int Cp, int dtype, void* stream) {
  MSML_CHECK(x && seg && N > 0 && H > 0 && W > 0 && Cp >= 18 && Cp % 8 == 0, MSML_ERR_SHAPE,
             "dap_fwd: bad shape");
  long npix = (long)N * H * W;
  int grid = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
  MSML_DISPATCH_DTYPE(dtype, "dap_fwd",
                      k_dap_fwd<DT><<<grid, 256, 0, (hipStream_t)stream>>>((const DT*)x, seg, mask, npix,
                                                                          H * W, Cp);)
  MSML_LAUNCH_OK("dap_fwd");
  return MSML_OK;
}

extern "C" int msml_dap_bwd(const float* dseg, void* dx, int N, int H, int W, int Cp, int dtype,
                            void* stream) {
  MSML_CHECK(dseg && dx && N > 0 && H > 0 && W > 0 && Cp >= 18 && Cp % 8 == 0, MSML_ERR_SHAPE,
             "dap_bwd: bad shape");
  long npix = (long)N * H * W;
  long total = npix * (Cp / 8);
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  MSML_DISPATCH_DTYPE(dtype, "dap_bwd",
                      k_dap_bwd<DT><<<grid, 256, 0, (hipStream_t)stream>>>(dseg, (DT*)dx, npix, H * W, Cp);)
  MSML_LAUNCH_OK("dap_bwd");
  return MSML_OK;
}

// ------------------------------------------------------------------ consensus seg loss -------
// Per image n and blob s in {0,1} (s == mask value): 5 sums over pixels of the blob
//   sup, P0 = sum p0, P1 = sum p1, L0 = sum log p0, L1 = sum log p1
// sums[n][s][5] in f32 (each image: one workgroup, wave + LDS reduction, fixed order).
__global__ void __launch_bounds__(256) k_seg_sums(const float* __restrict__ logit,
                                                  const long* __restrict__ msk, int HW,
                                                  float* __restrict__ sums) {
  const int n = blockIdx.x, t = threadIdx.x;
  const float* l0 = logit + (long)n * 2 * HW;
  const float* l1 = l0 + HW;
  float q[2][5];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int a = 0; a < 5; a++) q[s][a] = 0.f;
  for (int i = t; i < HW; i += 256) {
    float a = l0[i], b = l1[i];
    float mx = fmaxf(a, b);
    float lse = mx + logf(expf(a - mx) + expf(b - mx));
    float lp0 = a - lse, lp1 = b - lse;
    float p0 = expf(lp0), p1 = expf(lp1);
    int s = msk[(long)n * HW + i] != 0;
    float w0 = s == 0 ? 1.f : 0.f, w1 = 1.f - w0;
    q[0][0] += w0; q[0][1] += w0 * p0; q[0][2] += w0 * p1; q[0][3] += w0 * lp0; q[0][4] += w0 * lp1;
    q[1][0] += w1; q[1][1] += w1 * p0; q[1][2] += w1 * p1; q[1][3] += w1 * lp0; q[1][4] += w1 * lp1;
  }
  __shared__ float red[4][10];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int a = 0; a < 5; a++) {
      float v = wave_sum(q[s][a]);
      if ((t & 63) == 0) red[t >> 6][s * 5 + a] = v;
    }
  __syncthreads();
  if (t < 10) sums[n * 10 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// One workgroup: scalar loss and per-(n, s) gradient coefficients.
//   coef[n][s] = {valid, gA0, gA1, gB0, gB1}: dL/dp_c(x) = gA_c - gB_c / p_c(x)  for x in blob s
// mean_all: the blob mean divides by H * W instead of the blob's pixel count (reduce_pixel = 'all',
// consensus_loss.py:131-133); kl_all: the consensus term is averaged over N * H * W instead of the non-zero entries
// (reduce_pixel_kl = 'all', :159-160).
__global__ void k_seg_loss(const float* __restrict__ sums, int N, float alpha, float beta,
                           float* __restrict__ loss, float* __restrict__ coef, int HW, int mean_all, int kl_all) {
  __shared__ double sh_sup[2], sh_nll[2], sh_kl[2];
  __shared__ double red[3][2][256];
  const int t = threadIdx.x;
  // every thread sums its images (f64), then a fixed-order tree over the 256 threads
  double acc[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
  for (int n = t; n < N; n += blockDim.x)
    for (int s = 0; s < 2; s++) {
      const float* q = sums + (n * 2 + s) * 5;
      double sup = q[0];
      if (sup > 0.0) {
        const double D = mean_all ? (double)HW : sup;
        double pb0 = q[1] / D, pb1 = q[2] / D;
        double pbs = s == 0 ? pb0 : pb1;
        acc[0][s] += sup;
        acc[1][s] += -log(pbs);
        acc[2][s] += sup * pb0 * log(pb0) - pb0 * q[3] + sup * pb1 * log(pb1) - pb1 * q[4];
      }
    }
  for (int a = 0; a < 3; a++)
    for (int s = 0; s < 2; s++) red[a][s][t] = acc[a][s];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w)
      for (int a = 0; a < 3; a++)
        for (int s = 0; s < 2; s++) red[a][s][t] += red[a][s][t + w];
    __syncthreads();
  }
  if (t < 2) {
    sh_sup[t] = red[0][t][0];
    sh_nll[t] = red[1][t][0];
    sh_kl[t] = red[2][t][0];
  }
  __syncthreads();
  const int nblobs = (sh_sup[0] > 0.0) + (sh_sup[1] > 0.0);
  if (t == 0) {
    double total = 0.0;
    for (int s = 0; s < 2; s++)
      if (sh_sup[s] > 0.0)
        total += alpha * sh_nll[s] / N + beta * sh_kl[s] / (kl_all ? (double)N * HW : 2.0 * sh_sup[s]);
    loss[0] = (float)(total / (nblobs > 0 ? nblobs : 1));
  }
  for (int i = t; i < N * 2; i += blockDim.x) {
    const int s = i & 1;
    const float* q = sums + i * 5;
    float* c = coef + i * 5;
    double sup = q[0];
    if (sup > 0.0 && nblobs > 0) {
      const double D = mean_all ? (double)HW : sup;                // pbar_k = (sum_x p_k) / D
      double pb[2] = {q[1] / D, q[2] / D};
      double Z = kl_all ? (double)N * HW : 2.0 * sh_sup[s];
      double inv_nb = 1.0 / nblobs;
      for (int k = 0; k < 2; k++) {
        double A = sup * (log(pb[k]) + 1.0) - q[3 + k];            // d(kl_n)/d(pbar_k)
        double gA = beta * (A / D) / Z;
        if (k == s) gA += -alpha / (N * pb[k] * D);
        c[1 + k] = (float)(gA * inv_nb);
        c[3 + k] = (float)(beta * pb[k] / Z * inv_nb);
      }
      c[0] = 1.f;
    } else {
      c[0] = c[1] = c[2] = c[3] = c[4] = 0.f;
    }
  }
}

// dlogit_c = p_c * (G_c - sum_k p_k G_k),  G_c = gA_c - gB_c / p_c
__global__ void __launch_bounds__(256) k_seg_grad(const float* __restrict__ logit, const long* __restrict__ msk,
                                                  const float* __restrict__ coef, int HW, long total,
                                                  float* __restrict__ dlogit) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    long n = i / HW;
    int hw = (int)(i % HW);
    float a = logit[(n * 2) * HW + hw], b = logit[(n * 2 + 1) * HW + hw];
    float mx = fmaxf(a, b);
    float e0 = expf(a - mx), e1 = expf(b - mx);
    float inv = 1.f / (e0 + e1);
    float p0 = e0 * inv, p1 = e1 * inv;
    int s = msk[i] != 0;
    const float* c = coef + (n * 2 + s) * 5;
    // p_c * G_c = gA_c * p_c - gB_c
    float t0 = c[1] * p0 - c[3], t1 = c[2] * p1 - c[4];
    float dot = t0 + t1;
    dlogit[(n * 2) * HW + hw] = t0 - p0 * dot;
    dlogit[(n * 2 + 1) * HW + hw] = t1 - p1 * dot;
  }
}

extern "C" int msml_seg_consensus_loss_r(const float* logit, const long* msk, int N, int H, int W, float alpha,
                                         float beta, int reduce_pixel_all, int reduce_pixel_kl_all, float* loss,
                                         float* dlogit, float* workspace, long ws_floats, void* stream);

extern "C" int msml_seg_consensus_loss(const float* logit, const long* msk, int N, int H, int W,
                                       float alpha, float beta, float* loss, float* dlogit,
                                       float* workspace, long ws_floats, void* stream) {
  return msml_seg_consensus_loss_r(logit, msk, N, H, W, alpha, beta, 0, 0, loss, dlogit, workspace, ws_floats, stream);
}

extern "C" int msml_seg_consensus_loss_r(const float* logit, const long* msk, int N, int H, int W, float alpha,
                                         float beta, int reduce_pixel_all, int reduce_pixel_kl_all, float* loss,
                                         float* dlogit, float* workspace, long ws_floats, void* stream) {
  MSML_CHECK(logit && msk && loss && workspace && N > 0 && H > 0 && W > 0, MSML_ERR_SHAPE,
             "seg_consensus_loss: bad args");
  MSML_CHECK(ws_floats >= (long)N * 20, MSML_ERR_WORKSPACE, "seg_consensus_loss: workspace < %ld floats",
             (long)N * 20);
  hipStream_t st = (hipStream_t)stream;
  float* sums = workspace;
  float* coef = workspace + (long)N * 10;
  const int HW = H * W;
  k_seg_sums<<<N, 256, 0, st>>>(logit, msk, HW, sums);
  MSML_LAUNCH_OK("seg_sums");
  k_seg_loss<<<1, 256, 0, st>>>(sums, N, alpha, beta, loss, coef, HW, reduce_pixel_all, reduce_pixel_kl_all);
  MSML_LAUNCH_OK("seg_loss");
  if (dlogit) {
    long total = (long)N * HW;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    k_seg_grad<<<grid, 256, 0, st>>>(logit, msk, coef, HW, total, dlogit);
    MSML_LAUNCH_OK("seg_grad");
  }
  return MSML_OK;
}
