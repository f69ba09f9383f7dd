// Classification-head kernels: row L2-normalisation, ArcFace/CosFace margins, the PartialFC
// distributed softmax cross-entropy passes, fused SGD.
//
// Reference: headers/margin_losses.py:356-418 (AMArcFace), :241-305 (AMCosFace);
// headers/partial_fc.py:106-177 (prepare / forward_backward), train.py:188-191 (opt_pfc SGD).
// The GEMMs in between run on the implicit-GEMM kernels (conv_igemm.hip as 1x1 windows,
// conv_wgrad.hip for sub_weight.grad).
#include "common.h"

enum { HEAD_ARC = 0, HEAD_COS = 1 };

// ------------------------------------------------------------------ row normalise ------------
// F.normalize(w, dim=1, eps=1e-12): y = w / max(||w||, eps).  One wave per row; writes the
// normalised rows in storage dtype into dst (row stride `ld`, rows >= R zero-filled up to Rp)
// and optionally a f32 copy + the inverse norms for the backward.
template <typename T>
__global__ void __launch_bounds__(256) k_rownorm_fwd(const float* __restrict__ w, int R, int Rp, int E,
                                                     T* __restrict__ dst, int ld,
                                                     float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Rp) return;
  if (row >= R) {
    for (int i = lane; i < ld; i += 64) store1<T>(dst + (long)row * ld + i, 0.f);
    return;
  }
  const float* p = w + (long)row * E;
  float ss = 0.f;
  for (int i = lane; i < E; i += 64) ss += p[i] * p[i];
  ss = wave_sum(ss);
  float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
  for (int i = lane; i < ld; i += 64) store1<T>(dst + (long)row * ld + i, i < E ? p[i] * inv : 0.f);
}

// E % 512 == 0, E <= 1024, ld == E (the 512-wide ArcFace / PartialFC weight): one wave per row, a row read once as
// 32-B pieces that stay in registers between the norm and the scaled store (the scalar kernel above reads each row
// twice with 4-B loads: 101 us for the 85 742 x 512 head, 2.6 TB/s).
template <typename T>
__global__ void __launch_bounds__(256) k_rownorm_fwd_v8(const float* __restrict__ w, int R, int Rp, int E,
                                                        T* __restrict__ dst, float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Rp) return;
  T* d = dst + (long)row * E;
  Vec8 v[2];
  if (row >= R) {
#pragma unroll
    for (int j = 0; j < 8; j++) v[0].v[j] = 0.f;
    for (int i = lane * 8; i < E; i += 512) store8<T>(d + i, v[0]);
    return;
  }
  const float* p = w + (long)row * E;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < 2; c++)
    if (c * 512 + lane * 8 < E) {
      v[c] = load8<float>(p + c * 512 + lane * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) ss += v[c].v[j] * v[c].v[j];
    }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
#pragma unroll
  for (int c = 0; c < 2; c++)
    if (c * 512 + lane * 8 < E) {
#pragma unroll
      for (int j = 0; j < 8; j++) v[c].v[j] *= inv;
      store8<T>(d + c * 512 + lane * 8, v[c]);
    }
}

extern "C" int msml_rownorm_fwd(const float* w, int R, int Rp, int E, void* dst, int ld,
                                float* inv_norm, int dtype, void* stream) {
  MSML_CHECK(w && dst && R > 0 && Rp >= R && E > 0 && ld >= E, MSML_ERR_SHAPE, "rownorm_fwd: bad args");
  if (E % 512 == 0 && E <= 1024 && ld == E) {
    MSML_DISPATCH_DTYPE(dtype, "rownorm_fwd",
                        k_rownorm_fwd_v8<DT><<<cdiv(Rp, 4), 256, 0, (hipStream_t)stream>>>(w, R, Rp, E, (DT*)dst, inv_norm);)
    MSML_LAUNCH_OK("rownorm_fwd");
    return MSML_OK;
  }
  MSML_DISPATCH_DTYPE(dtype, "rownorm_fwd",
                      k_rownorm_fwd<DT><<<cdiv(Rp, 4), 256, 0, (hipStream_t)stream>>>(w, R, Rp, E, (DT*)dst,
                                                                                    ld, inv_norm);)
  MSML_LAUNCH_OK("rownorm_fwd");
  return MSML_OK;
}

// backward of y = w * inv:  dw = (dy - y * <y, dy>) * inv, with y recomputed from w and inv.
// dy: f32 [R][ldy].  accumulate: dw += (for gradient accumulation).
__global__ void __launch_bounds__(256) k_rownorm_bwd(const float* __restrict__ w, const float* __restrict__ inv_norm,
                                                     const float* __restrict__ dy, int ldy, int R, int E,
                                                     float* __restrict__ dw, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float inv = inv_norm[row];
  const float* p = w + (long)row * E;
  const float* g = dy + (long)row * ldy;
  float dot = 0.f;
  for (int i = lane; i < E; i += 64) dot += p[i] * inv * g[i];
  dot = wave_sum(dot);
  for (int i = lane; i < E; i += 64) {
    float v = (g[i] - p[i] * inv * dot) * inv;
    long o = (long)row * E + i;
    dw[o] = accumulate ? dw[o] + v : v;
  }
}

// E % 512 == 0, E <= 1024, ldy % 4 == 0: rows of w and dy held in registers between the dot product and the store
__global__ void __launch_bounds__(256) k_rownorm_bwd_v8(const float* __restrict__ w, const float* __restrict__ inv_norm,
                                                        const float* __restrict__ dy, int ldy, int R, int E,
                                                        float* __restrict__ dw, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float inv = inv_norm[row];
  const float* p = w + (long)row * E;
  const float* g = dy + (long)row * ldy;
  float* o = dw + (long)row * E;
  Vec8 pv[2], gv[2];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < 2; c++)
    if (c * 512 + lane * 8 < E) {
      pv[c] = load8<float>(p + c * 512 + lane * 8);
      gv[c] = load8<float>(g + c * 512 + lane * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) dot += pv[c].v[j] * inv * gv[c].v[j];
    }
  dot = wave_sum(dot);
#pragma unroll
  for (int c = 0; c < 2; c++)
    if (c * 512 + lane * 8 < E) {
      Vec8 r;
      if (accumulate) r = load8<float>(o + c * 512 + lane * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float v = (gv[c].v[j] - pv[c].v[j] * inv * dot) * inv;
        r.v[j] = accumulate ? r.v[j] + v : v;
      }
      store8<float>(o + c * 512 + lane * 8, r);
    }
}

extern "C" int msml_rownorm_bwd(const float* w, const float* inv_norm, const float* dy, int ldy, int R,
                                int E, float* dw, int accumulate, void* stream) {
  MSML_CHECK(w && inv_norm && dy && dw && R > 0 && E > 0 && ldy >= E, MSML_ERR_SHAPE, "rownorm_bwd: bad args");
  if (E % 512 == 0 && E <= 1024 && ldy % 4 == 0 && ((size_t)dy % 16) == 0 && ((size_t)dw % 16) == 0 && ((size_t)w % 16) == 0) {
    k_rownorm_bwd_v8<<<cdiv(R, 4), 256, 0, (hipStream_t)stream>>>(w, inv_norm, dy, ldy, R, E, dw, accumulate);
    MSML_LAUNCH_OK("rownorm_bwd");
    return MSML_OK;
  }
  k_rownorm_bwd<<<cdiv(R, 4), 256, 0, (hipStream_t)stream>>>(w, inv_norm, dy, ldy, R, E, dw, accumulate);
  MSML_LAUNCH_OK("rownorm_bwd");
  return MSML_OK;
}

// ------------------------------------------------------------------ margins -------------------
// target logit and its derivative wrt the cosine c (theta = acos c):
//   Arc: s cos(theta + m - k (theta - a));   Cos: s (c - m + k (theta - a))
__device__ __forceinline__ void margin_target(float c, int kind, float s, float m, float a, float k,
                                              float& out, float& dout_dc) {
  float theta = acosf(c);
  float dth = -1.f / sqrtf(fmaxf(1.f - c * c, 1e-30f));
  if (kind == HEAD_ARC) {
    float phi = theta + m - k * (theta - a);
    out = s * cosf(phi);
    dout_dc = -s * sinf(phi) * (1.f - k) * dth;
  } else {
    out = s * (c - m + k * (theta - a));
    dout_dc = s * (1.f + k * dth);
  }
}

// In place on cos [N][ld] f32 (C valid columns): logits = s * cos, target entries get the margin.
// label[i] == -1: no target in this row (PartialFC rows owned by another rank).
__global__ void __launch_bounds__(256) k_margin_fwd(float* __restrict__ cosm, const long* __restrict__ label,
                                                    int N, int C, int ld, int kind, float s, float m,
                                                    float a, float k) {
  const int row = blockIdx.y;
  const long y = label[row];
  float* p = cosm + (long)row * ld;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < C; j += gridDim.x * blockDim.x) {
    float c = p[j];
    float o;
    if (j == y) {
      float d;
      margin_target(c, kind, s, m, a, k, o, d);
    } else {
      o = s * c;
    }
    p[j] = o;
  }
}

extern "C" int msml_margin_fwd(float* cosm, const long* label, int N, int C, int ld, int kind, float s,
                               float m, float a, float k, void* stream) {
  MSML_CHECK(cosm && label && N > 0 && C > 0 && ld >= C && (kind == 0 || kind == 1), MSML_ERR_SHAPE,
             "margin_fwd: bad args");
  dim3 grid(cdiv(C, 256) < 64 ? cdiv(C, 256) : 64, N);
  k_margin_fwd<<<grid, 256, 0, (hipStream_t)stream>>>(cosm, label, N, C, ld, kind, s, m, a, k);
  MSML_LAUNCH_OK("margin_fwd");
  return MSML_OK;
}

// dcos[i][j] = dlogit[i][j] * d logit / d cos, written in storage dtype with row stride ldo
// (columns C..ldo-1 zero) so it can feed the backward GEMMs directly.  cos_t[i] = the target
// cosine saved by the forward (needed because margin_fwd overwrote it).
template <typename T>
__global__ void __launch_bounds__(256) k_margin_bwd(const float* __restrict__ dlogit, int ldg,
                                                    const long* __restrict__ label,
                                                    const float* __restrict__ cos_t, int N, int C,
                                                    T* __restrict__ dcos, int ldo, int kind, float s,
                                                    float m, float a, float k) {
  const int row = blockIdx.y;
  const long y = label[row];
  const float* g = dlogit + (long)row * ldg;
  T* o = dcos + (long)row * ldo;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ldo; j += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (j < C) {
      if (j == y) {
        float out, d;
        margin_target(cos_t[row], kind, s, m, a, k, out, d);
        v = g[j] * d;
      } else {
        v = g[j] * s;
      }
    }
    store1<T>(o + j, v);
  }
}

extern "C" int msml_margin_bwd(const float* dlogit, int ldg, const long* label, const float* cos_t, int N,
                               int C, void* dcos, int ldo, int kind, float s, float m, float a, float k,
                               int dtype, void* stream) {
  MSML_CHECK(dlogit && label && cos_t && dcos && N > 0 && C > 0 && ldg >= C && ldo >= C, MSML_ERR_SHAPE,
             "margin_bwd: bad args");
  dim3 grid(cdiv(ldo, 256) < 64 ? cdiv(ldo, 256) : 64, N);
  MSML_DISPATCH_DTYPE(dtype, "margin_bwd",
                      k_margin_bwd<DT><<<grid, 256, 0, (hipStream_t)stream>>>(dlogit, ldg, label, cos_t, N, C,
                                                                             (DT*)dcos, ldo, kind, s, m, a, k);)
  MSML_LAUNCH_OK("margin_bwd");
  return MSML_OK;
}

// gather the target cosine before margin_fwd overwrites it (0 for label == -1)
__global__ void k_gather_target(const float* __restrict__ cosm, int ld, const long* __restrict__ label,
                                int N, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  long y = label[i];
  out[i] = y >= 0 ? cosm[(long)i * ld + y] : 0.f;
}

extern "C" int msml_gather_target(const float* cosm, int ld, const long* label, int N, float* out,
                                  void* stream) {
  MSML_CHECK(cosm && label && out && N > 0, MSML_ERR_SHAPE, "gather_target: bad args");
  k_gather_target<<<cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(cosm, ld, label, N, out);
  MSML_LAUNCH_OK("gather_target");
  return MSML_OK;
}

// ------------------------------------------------------------------ PartialFC softmax-CE ------
// Pass A (headers/partial_fc.py:132-141): per row of the LOCAL cosine block [N][ld] (C valid
// columns), apply scale/margin on the fly and produce the row max and sum exp(logit - rowmax)
// in one online pass (no intermediate logits written).  One workgroup per row.
__global__ void __launch_bounds__(256) k_pfc_rowstats(const float* __restrict__ cosm, int ld, int C,
                                                      const long* __restrict__ label, int kind, float s,
                                                      float m, float a, float k,
                                                      float* __restrict__ rowmax, float* __restrict__ rowsum) {
  const int row = blockIdx.x, t = threadIdx.x;
  const float* p = cosm + (long)row * ld;
  const long y = label[row];
  float mx = -INFINITY, sm = 0.f;
  for (int j = t; j < C; j += 256) {
    float c = p[j], l;
    if (j == y) {
      float d;
      margin_target(c, kind, s, m, a, k, l, d);
    } else {
      l = s * c;
    }
    if (l > mx) {
      sm = sm * __expf(mx - l) + 1.f;
      mx = l;
    } else {
      sm += __expf(l - mx);
    }
  }
  __shared__ float smx[4], ssm[4];
  float wm = wave_max(mx);
  float ws = wave_sum(mx == -INFINITY ? 0.f : sm * __expf(mx - wm));
  if ((t & 63) == 0) { smx[t >> 6] = wm; ssm[t >> 6] = ws; }
  __syncthreads();
  if (t == 0) {
    float M = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    float S = 0.f;
    for (int w = 0; w < 4; w++) S += smx[w] == -INFINITY ? 0.f : ssm[w] * __expf(smx[w] - M);
    rowmax[row] = M;
    rowsum[row] = S;
  }
}

// ld % 4 == 0: 1024 threads per row, 16-B loads, the running (max, sum) rescaled once per four logits (the kernel
// above walks a row with 256 scalar-load lanes and one dependent exp per logit: 105 us for 256 x 85 742).
__global__ void __launch_bounds__(1024) k_pfc_rowstats_v4(const float* __restrict__ cosm, int ld, int C,
                                                          const long* __restrict__ label, int kind, float s,
                                                          float m, float a, float k,
                                                          float* __restrict__ rowmax, float* __restrict__ rowsum) {
  const int row = blockIdx.x, t = threadIdx.x;
  const float* p = cosm + (long)row * ld;
  const long y = label[row];
  float mx = -INFINITY, sm = 0.f;
  for (int j = t * 4; j < C; j += 4096) {
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(p + j);
    float l[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      l[e] = s * c4[e];
      if (j + e == y) {
        float d;
        margin_target(c4[e], kind, s, m, a, k, l[e], d);
      }
      if (j + e >= C) l[e] = -INFINITY;
    }
    const float nm = fmaxf(fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3])), mx);     // finite: l[0] is a valid column
    sm = sm * __expf(mx - nm) + ((__expf(l[0] - nm) + __expf(l[1] - nm)) + (__expf(l[2] - nm) + __expf(l[3] - nm)));
    mx = nm;
  }
  __shared__ float smx[16], ssm[16];
  const float wm = wave_max(mx);
  const float ws = wave_sum(mx == -INFINITY ? 0.f : sm * __expf(mx - wm));
  if ((t & 63) == 0) { smx[t >> 6] = wm; ssm[t >> 6] = ws; }
  __syncthreads();
  if (t == 0) {
    float M = smx[0];
    for (int w = 1; w < 16; w++) M = fmaxf(M, smx[w]);
    float S = 0.f;
    for (int w = 0; w < 16; w++) S += smx[w] == -INFINITY ? 0.f : ssm[w] * __expf(smx[w] - M);
    rowmax[row] = M;
    rowsum[row] = S;
  }
}

extern "C" int msml_pfc_rowstats(const float* cosm, int ld, int N, int C, const long* label, int kind,
                                 float s, float m, float a, float k, float* rowmax, float* rowsum,
                                 void* stream) {
  MSML_CHECK(cosm && label && rowmax && rowsum && N > 0 && C > 0 && ld >= C, MSML_ERR_SHAPE,
             "pfc_rowstats: bad args");
  if (ld % 4 == 0 && ((size_t)cosm % 16) == 0) {
    k_pfc_rowstats_v4<<<N, 1024, 0, (hipStream_t)stream>>>(cosm, ld, C, label, kind, s, m, a, k, rowmax, rowsum);
    MSML_LAUNCH_OK("pfc_rowstats");
    return MSML_OK;
  }
  k_pfc_rowstats<<<N, 256, 0, (hipStream_t)stream>>>(cosm, ld, C, label, kind, s, m, a, k, rowmax, rowsum);
  MSML_LAUNCH_OK("pfc_rowstats");
  return MSML_OK;
}

// Pass B (headers/partial_fc.py:144-167 + autograd through the margin): with the GLOBAL row max
// gmax and denominator gsum (after the all-reduces), p = exp(logit - gmax) / gsum,
//   y = label-smoothed one-hot over the LOCAL classes for rows whose target is local
//       (0.9 at the target, 0.1/(C-1) elsewhere; F9), 0 for other rows,
//   dlogit = (p - y) / Ntot;   dcos = dlogit * dlogit/dcos   -> storage dtype [N][ldo]
//   ptarget[row] = p at the target (0 if not local) for the loss all-reduce.
template <typename T>
__global__ void __launch_bounds__(256) k_pfc_grad(const float* __restrict__ cosm, int ld, int C,
                                                  const long* __restrict__ label, int kind, float s,
                                                  float m, float a, float k, const float* __restrict__ gmax,
                                                  const float* __restrict__ gsum, float eps_ls, float inv_n,
                                                  T* __restrict__ dcos, int ldo, float* __restrict__ ptarget) {
  const int row = blockIdx.y;
  const float* p = cosm + (long)row * ld;
  T* o = dcos + (long)row * ldo;
  const long y = label[row];
  const float M = gmax[row], inv_s = 1.f / gsum[row];
  const float off = y >= 0 ? eps_ls / (float)(C - 1) : 0.f;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ldo; j += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (j < C) {
      float c = p[j], l, d;
      if (j == y) {
        margin_target(c, kind, s, m, a, k, l, d);
      } else {
        l = s * c;
        d = s;
      }
      float prob = __expf(l - M) * inv_s;
      float tgt = j == y ? 1.f - eps_ls : off;
      if (j == y) ptarget[row] = prob;
      v = (prob - tgt) * inv_n * d;
    }
    store1<T>(o + j, v);
  }
  if (y < 0 && blockIdx.x == 0 && threadIdx.x == 0) ptarget[row] = 0.f;
}

extern "C" int msml_pfc_grad(const float* cosm, int ld, int N, int C, const long* label, int kind, float s,
                             float m, float a, float k, const float* gmax, const float* gsum, float eps_ls,
                             float inv_n, void* dcos, int ldo, float* ptarget, int dtype, void* stream) {
  MSML_CHECK(cosm && label && gmax && gsum && dcos && ptarget && N > 0 && C > 1 && ld >= C && ldo >= C,
             MSML_ERR_SHAPE, "pfc_grad: bad args");
  dim3 grid(cdiv(ldo, 256) < 128 ? cdiv(ldo, 256) : 128, N);
  MSML_DISPATCH_DTYPE(dtype, "pfc_grad",
                      k_pfc_grad<DT><<<grid, 256, 0, (hipStream_t)stream>>>(cosm, ld, C, label, kind, s, m, a, k,
                                                                           gmax, gsum, eps_ls, inv_n, (DT*)dcos,
                                                                           ldo, ptarget);)
  MSML_LAUNCH_OK("pfc_grad");
  return MSML_OK;
}

// ------------------------------------------------------------------ fused SGD + clip ---------
// torch.optim.SGD(momentum, weight_decay, dampening 0, nesterov False) (train.py:179-191):
//   g = grad * coef + wd * w;  buf = first ? g : mu * buf + g;  w -= lr * buf
// coef (device scalar, optional) is the clip_grad_norm_ factor (train.py:270,275) so that
// clipping costs no extra pass over the gradients and no host synchronisation.
// lr_ptr != nullptr: the learning rate is read from device memory (a captured step then follows LambdaLR without
// a re-capture: the host updates the scalar between replays).
__global__ void __launch_bounds__(256) k_sgd(float* __restrict__ w, const float* __restrict__ grad,
                                             float* __restrict__ mom, long n4, long n, float lr, float mu,
                                             float wd, int first, const float* __restrict__ coef_ptr,
                                             const float* __restrict__ lr_ptr) {
  const float coef = coef_ptr ? coef_ptr[0] : 1.f;
  if (lr_ptr) lr = lr_ptr[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 ww = reinterpret_cast<f32x4*>(w)[i];
    f32x4 gg = reinterpret_cast<const f32x4*>(grad)[i];
    f32x4 bb = {0.f, 0.f, 0.f, 0.f};
    if (!first) bb = reinterpret_cast<f32x4*>(mom)[i];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float g = gg[j] * coef + wd * ww[j];
      float b = first ? g : mu * bb[j] + g;
      bb[j] = b;
      ww[j] -= lr * b;
    }
    reinterpret_cast<f32x4*>(mom)[i] = bb;
    reinterpret_cast<f32x4*>(w)[i] = ww;
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long i = n4 * 4 + threadIdx.x;
    float g = grad[i] * coef + wd * w[i];
    float b = first ? g : mu * mom[i] + g;
    mom[i] = b;
    w[i] -= lr * b;
  }
}

extern "C" int msml_sgd_momentum(float* w, const float* grad, float* mom, long n, float lr, float mu,
                                 float wd, int first_step, const float* clip_coef, void* stream) {
  MSML_CHECK(w && grad && mom && n > 0, MSML_ERR_SHAPE, "sgd_momentum: bad args");
  MSML_CHECK(((uintptr_t)w & 15) == 0 && ((uintptr_t)grad & 15) == 0 && ((uintptr_t)mom & 15) == 0,
             MSML_ERR_SHAPE, "sgd_momentum: buffers must be 16-byte aligned");
  long n4 = n / 4;
  long b = (n4 + 255) / 256;
  if (b < 1) b = 1;
  k_sgd<<<(int)(b < 4096 ? b : 4096), 256, 0, (hipStream_t)stream>>>(w, grad, mom, n4, n, lr, mu, wd, first_step,
                                                                    clip_coef, nullptr);
  MSML_LAUNCH_OK("sgd_momentum");
  return MSML_OK;
}

extern "C" int msml_sgd_momentum_dev(float* w, const float* grad, float* mom, long n, const float* lr, float mu,
                                     float wd, const float* coef, void* stream) {
  MSML_CHECK(w && grad && mom && lr && n > 0, MSML_ERR_SHAPE, "sgd_momentum_dev: bad args");
  MSML_CHECK(((uintptr_t)w & 15) == 0 && ((uintptr_t)grad & 15) == 0 && ((uintptr_t)mom & 15) == 0,
             MSML_ERR_SHAPE, "sgd_momentum_dev: buffers must be 16-byte aligned");
  long n4 = n / 4;
  long b = (n4 + 255) / 256;
  if (b < 1) b = 1;
  k_sgd<<<(int)(b < 4096 ? b : 4096), 256, 0, (hipStream_t)stream>>>(w, grad, mom, n4, n, 0.f, mu, wd, 0, coef, lr);
  MSML_LAUNCH_OK("sgd_momentum_dev");
  return MSML_OK;
}

// Global L2 norm of a flat gradient buffer and the clip factor min(1, max_norm / (norm + 1e-6))
// (torch.nn.utils.clip_grad_norm_, train.py:270,275).  Two-level deterministic reduction.
// 16 B per lane, four loads in flight per thread (236 MB of gradients: HBM-bound); f64 block sums
__global__ void __launch_bounds__(256) k_sumsq(const float* __restrict__ x, long n, float* __restrict__ partial) {
  const long n4 = n >> 2, stride = (long)gridDim.x * blockDim.x;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    s0 += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    s1 += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    s2 += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    s3 += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  }
  for (; i < n4; i += stride) {
    const f32x4 a = x4[i];
    s0 += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {    // tail (n not a multiple of 4)
    const float v = x[n4 * 4 + threadIdx.x];
    s0 += v * v;
  }
  float s = (s0 + s1) + (s2 + s3);
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// one workgroup: thread t sums rows t, t + 256, ... in f64, then a fixed-order tree
// scale: the gradient buffer holds scale^-1 times the gradient (the SUM over W ranks of a data-parallel job, scale =
// 1 / W): out[0] = norm of the scaled gradient, out[1] = scale * clip factor -- the one coefficient k_sgd applies.
__global__ void __launch_bounds__(256) k_norm_finalize(const float* __restrict__ partial, int rows, float max_norm,
                                                       float scale, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < rows; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float norm = (float)sqrt(red[0]) * scale;
    out[0] = norm;
    float c = max_norm / (norm + 1e-6f);
    out[1] = (c < 1.f ? c : 1.f) * scale;
  }
}

static int grad_norm_clip_impl(const float* grad, long n, float max_norm, float scale, float* out2, float* workspace,
                               long ws_floats, void* stream) {
  MSML_CHECK(grad && out2 && workspace && n > 0 && ((uintptr_t)grad & 15) == 0, MSML_ERR_SHAPE,
             "grad_norm_clip: bad args (grad must be 16-byte aligned)");
  MSML_CHECK(scale > 0.f, MSML_ERR_SHAPE, "grad_norm_clip: scale %f", scale);
  long b = (n + 255) / 256;
  int rows = (int)(b < 1024 ? b : 1024);
  MSML_CHECK(ws_floats >= rows, MSML_ERR_WORKSPACE, "grad_norm_clip: workspace too small");
  k_sumsq<<<rows, 256, 0, (hipStream_t)stream>>>(grad, n, workspace);
  MSML_LAUNCH_OK("grad_norm_clip");
  k_norm_finalize<<<1, 256, 0, (hipStream_t)stream>>>(workspace, rows, max_norm, scale, out2);
  MSML_LAUNCH_OK("grad_norm_finalize");
  return MSML_OK;
}

extern "C" int msml_grad_norm_clip(const float* grad, long n, float max_norm, float* out2,
                                   float* workspace, long ws_floats, void* stream) {
  return grad_norm_clip_impl(grad, n, max_norm, 1.f, out2, workspace, ws_floats, stream);
}

extern "C" int msml_grad_norm_clip_scaled(const float* grad, long n, float max_norm, float scale, float* out2,
                                          float* workspace, long ws_floats, void* stream) {
  return grad_norm_clip_impl(grad, n, max_norm, scale, out2, workspace, ws_floats, stream);
}
