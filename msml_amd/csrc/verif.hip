// Device side of the verification metrics (eval/verification.py:54-199 of the reference: 10-fold
// accuracy over a threshold grid, TAR @ FAR): per-pair squared distance of the L2-normalised
// embeddings in f64, and a per-(fold, same/different) histogram over the threshold grid from which
// every tp / fp / tn / fn count of every threshold and fold follows by prefix sums -- the reference
// evaluates nrof_thresholds x nrof_folds boolean passes over all pairs in numpy.  Integer counts:
// bit-exact against the CPU restatement.
#include "common.h"

// emb: [2 * n_pairs][E] f32 (rows 2i, 2i+1 form pair i, verification.py:184-185); dist[i] =
// sum_j (a_j/|a| - b_j/|b|)^2 in f64 (sklearn.preprocessing.normalize + np.subtract/square/sum, :298-301,78-79).
// One wave per pair.
// T = float: f32 embeddings; T = double: the f64 sum of the orig + flip passes (verification.py:283,299-300 keeps both
// passes in float64 arrays and sums / normalises them there).
template <typename T>
__global__ void __launch_bounds__(256) k_pair_sqdist(const T* __restrict__ emb, int n_pairs, int E,
                                                     double* __restrict__ dist) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n_pairs) return;
  const T* a = emb + (long)(2 * pair) * E;
  const T* b = a + E;
  double sa = 0.0, sb = 0.0;
  for (int j = lane; j < E; j += 64) {
    const double x = a[j], y = b[j];
    sa += x * x;
    sb += y * y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sa += __shfl_xor(sa, o, 64);
    sb += __shfl_xor(sb, o, 64);
  }
  double na = sqrt(sa), nb = sqrt(sb);
  if (na == 0.0) na = 1.0;                 // sklearn: zero rows are left as they are
  if (nb == 0.0) nb = 1.0;
  double d = 0.0;
  for (int j = lane; j < E; j += 64) {
    const double t = (double)a[j] / na - (double)b[j] / nb;
    d += t * t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if (lane == 0) dist[pair] = d;
}

// hist[fold][same][bin], bin = number of thresholds <= dist = the first threshold index k with
// dist < thr[k] (np.less(dist, threshold), :111,167); fold = the KFold(n_splits, shuffle=False) test fold of
// the pair (contiguous blocks, the first n % k folds one longer).  thr ascending.
__global__ void __launch_bounds__(256) k_pair_hist(const double* __restrict__ dist, const unsigned char* __restrict__ same,
                                                   int n_pairs, const double* __restrict__ thr, int nthr, int nfolds,
                                                   int* __restrict__ hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const double d = dist[i];
  int lo = 0, hi = nthr;                   // first k with thr[k] > d
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (d < thr[mid]) hi = mid;
    else lo = mid + 1;
  }
  const int base = n_pairs / nfolds, rem = n_pairs % nfolds;
  int fold;
  if (i < rem * (base + 1)) fold = i / (base + 1);
  else fold = rem + (i - rem * (base + 1)) / (base > 0 ? base : 1);
  atomicAdd(&hist[((long)fold * 2 + (same[i] ? 1 : 0)) * (nthr + 1) + lo], 1);
}

extern "C" int msml_pair_sqdist(const float* emb, int n_pairs, int E, double* dist, void* stream) {
  MSML_CHECK(emb && dist && n_pairs > 0 && E > 0, MSML_ERR_SHAPE, "pair_sqdist: bad shape n_pairs=%d E=%d", n_pairs, E);
  k_pair_sqdist<float><<<cdiv(n_pairs, 4), 256, 0, (hipStream_t)stream>>>(emb, n_pairs, E, dist);
  MSML_LAUNCH_OK("pair_sqdist");
  return MSML_OK;
}

extern "C" int msml_pair_sqdist_f64(const double* emb, int n_pairs, int E, double* dist, void* stream) {
  MSML_CHECK(emb && dist && n_pairs > 0 && E > 0, MSML_ERR_SHAPE, "pair_sqdist_f64: bad shape n_pairs=%d E=%d", n_pairs, E);
  k_pair_sqdist<double><<<cdiv(n_pairs, 4), 256, 0, (hipStream_t)stream>>>(emb, n_pairs, E, dist);
  MSML_LAUNCH_OK("pair_sqdist_f64");
  return MSML_OK;
}

extern "C" int msml_pair_hist(const double* dist, const unsigned char* same, int n_pairs, const double* thr, int nthr,
                              int nfolds, int* hist, void* stream) {
  MSML_CHECK(dist && same && thr && hist && n_pairs > 0 && nthr > 0 && nfolds > 0 && nfolds <= n_pairs, MSML_ERR_SHAPE,
             "pair_hist: bad shape n_pairs=%d nthr=%d nfolds=%d", n_pairs, nthr, nfolds);
  hipError_t e = hipMemsetAsync(hist, 0, sizeof(int) * (size_t)nfolds * 2 * (nthr + 1), (hipStream_t)stream);
  MSML_CHECK(e == hipSuccess, MSML_ERR_LAUNCH, "pair_hist: memset failed");
  k_pair_hist<<<cdiv(n_pairs, 256), 256, 0, (hipStream_t)stream>>>(dist, same, n_pairs, thr, nthr, nfolds, hist);
  MSML_LAUNCH_OK("pair_hist");
  return MSML_OK;
}
