"""Verification on the HIP path: the embedding protocol and the pair metrics of the reference
(eval/verification.py:239-306 `test`, :54-199 `calculate_roc` / `calculate_val` / `evaluate`;
eval/qeval_mxnet.py:326-390 uses the same orig + flip sum).

* `extract_embeddings(model, x)`: embeddings of the batch and of its horizontal flip, summed (the
  reference normalises after the sum, verification.py:299-301).
* `evaluate(embeddings, issame)`: 10-fold accuracy over the 0..4 / 0.01 threshold grid and TAR @ FAR 1e-3
  over the 0..4 / 0.001 grid.  The reference makes nrof_thresholds x nrof_folds boolean passes over all
  pairs in numpy; here ONE kernel computes the f64 pair distances and ONE builds a [fold][same][threshold]
  histogram (integer atomics, deterministic), from which every confusion-matrix entry of every threshold
  and fold is a prefix sum -- the remaining arithmetic is on a table of a few thousand integers.
"""
import numpy as np
import torch

from ._lib import call

def slinear_first_order(x, y, xq):
    """interp1d(x, y, kind='slinear')(xq) as scipy 1.5.4 (the reference's pin, requirements.txt:100)
    evaluates it when x holds duplicates -- far_train is a step function, so it always does: interp1d
    sorts x with a stable argsort and builds the degree-1 B-spline on the knots [x0, x..., xn] without
    the strictly-increasing check newer scipy applies (>= 1.10 raises 'Expect x to not have
    duplicates').  Empty knot intervals are skipped, so the value at xq interpolates between the LAST
    sample of the run x[j] <= xq and the first sample of the next run."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    order = np.argsort(x, kind="mergesort")
    x, y = x[order], y[order]
    j = int(np.searchsorted(x, xq, side="right")) - 1
    j = min(max(j, 0), len(x) - 2)
    while j > 0 and x[j + 1] == x[j]:
        j -= 1
    if x[j + 1] == x[j]:
        return float(y[j])
    return float(y[j] + (y[j + 1] - y[j]) * (xq - x[j]) / (x[j + 1] - x[j]))


@torch.no_grad()
def extract_embeddings(model, x, normalize=False):
    """(B, E) f64: model(x) + model(flip(x)) summed in float64 like the reference, which stores both passes in
    float64 arrays (verification.py:283,299); normalize=True returns the f32 L2-normalised rows instead (:300, on
    device) for callers that only need cosines."""
    f1 = model(x)[0]
    f2 = model(x.flip(3))[0]
    out = f1.double() + f2.double()
    if normalize:
        from . import functional as Fh
        out = Fh.normalize(out.float())
    return out


@torch.no_grad()
def pair_cosine(emb):
    """Cosine of every pair (rows 2i, 2i+1) of summed embeddings: 1 - dist / 2 with the f64 distance of
    the normalised rows."""
    return 1.0 - pair_sqdist(emb) / 2.0


@torch.no_grad()
def pair_sqdist(emb):
    """emb: (2 * n_pairs, E) f32 or f64 (extract_embeddings) on the GPU, NOT normalised -> (n_pairs,) f64 squared
    distances of the L2-normalised rows (sklearn.preprocessing.normalize + np.sum(np.square(diff), 1))."""
    f64 = emb.dtype == torch.float64
    emb = emb.contiguous() if f64 else emb.float().contiguous()
    n2, e = emb.shape
    assert n2 % 2 == 0 and emb.is_cuda
    dist = torch.empty(n2 // 2, dtype=torch.float64, device=emb.device)
    call("msml_pair_sqdist_f64" if f64 else "msml_pair_sqdist", emb, n2 // 2, e, dist)
    return dist


def _fold_hist(dist, issame, thresholds, nfolds):
    n = dist.numel()
    thr = torch.as_tensor(np.asarray(thresholds, np.float64), device=dist.device)
    same = torch.as_tensor(np.asarray(issame).astype(np.uint8), device=dist.device)
    hist = torch.empty(nfolds, 2, thr.numel() + 1, dtype=torch.int32, device=dist.device)
    call("msml_pair_hist", dist, same, n, thr, thr.numel(), nfolds, hist)
    return hist.cpu().numpy().astype(np.int64)


def _counts(hist):
    """accept[f][s][k] = pairs of test fold f with issame == s and dist < thr[k]; tot[f][s]."""
    acc = np.cumsum(hist, axis=2)[:, :, :-1]
    tot = hist.sum(axis=2)
    return acc, tot


def evaluate(emb, issame, nrof_folds=10, far_target=1e-3):
    """emb: (2 * n_pairs, E) summed (orig + flip) embeddings on the GPU.  Returns the reference's
    `evaluate` tuple (tpr, fpr, accuracy, val, val_std, far) -- verification.py:181-199."""
    issame = np.asarray(issame).astype(bool)
    dist = pair_sqdist(emb)
    n = dist.numel()
    assert len(issame) == n
    # ---- calculate_roc (:54-107): thresholds 0..4 step 0.01 ----
    thr = np.arange(0, 4, 0.01)
    acc, tot = _counts(_fold_hist(dist, issame, thr, nrof_folds))          # test-fold counts
    tp, fp = acc[:, 1], acc[:, 0]                                          # [fold][thr]
    n_same, n_diff = tot[:, 1:2], tot[:, 0:1]
    all_tp, all_fp = tp.sum(0, keepdims=True), fp.sum(0, keepdims=True)
    tr_tp, tr_fp = all_tp - tp, all_fp - fp                                # train = all folds but f
    tr_same, tr_diff = n_same.sum() - n_same, n_diff.sum() - n_diff
    acc_train = (tr_tp + (tr_diff - tr_fp)) / (tr_same + tr_diff)
    best = np.argmax(acc_train, axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        tprs = np.where(n_same > 0, tp / np.maximum(n_same, 1), 0.0)
        fprs = np.where(n_diff > 0, fp / np.maximum(n_diff, 1), 0.0)
    size = (n_same + n_diff)[:, 0]
    f = np.arange(nrof_folds)
    accuracy = (tp[f, best] + (n_diff[:, 0] - fp[f, best])) / size
    tpr, fpr = tprs.mean(0), fprs.mean(0)
    # ---- calculate_val (:125-163): thresholds 0..4 step 0.001, FAR target ----
    thr2 = np.arange(0, 4, 0.001)
    acc2, tot2 = _counts(_fold_hist(dist, issame, thr2, nrof_folds))
    tp2, fp2 = acc2[:, 1], acc2[:, 0]
    s2, d2 = tot2[:, 1:2], tot2[:, 0:1]
    far_train = (fp2.sum(0, keepdims=True) - fp2) / (d2.sum() - d2)
    val = np.zeros(nrof_folds)
    far = np.zeros(nrof_folds)
    d_host = None
    for k in range(nrof_folds):
        if np.max(far_train[k]) >= far_target:
            t = slinear_first_order(far_train[k], thr2, far_target)
        else:
            t = 0.0
        # the interpolated threshold is not on the grid: count the test fold directly (a few hundred pairs)
        if d_host is None:
            d_host = dist.cpu().numpy()
            base, rem = n // nrof_folds, n % nrof_folds
            starts = np.cumsum([0] + [base + (1 if i < rem else 0) for i in range(nrof_folds)])
        sl = slice(starts[k], starts[k + 1])
        pred = np.less(d_host[sl], t)
        val[k] = float(np.sum(pred & issame[sl])) / float(np.sum(issame[sl]))
        far[k] = float(np.sum(pred & ~issame[sl])) / float(np.sum(~issame[sl]))
    return tpr, fpr, accuracy, float(np.mean(val)), float(np.std(val)), float(np.mean(far))
