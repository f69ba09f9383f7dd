"""CPU restatement of the reference's verification protocol and metrics (test oracle; not used by the
product).

eval/verification.py of the reference cannot be imported in the build container (it imports mxnet at
:31, absent), so these functions are written from its text: `embed_protocol` restates test() :283-301
(orig + h-flip embeddings summed, L2-normalised), `calculate_roc` :54-107, `calculate_accuracy`
:110-122, `calculate_val` :125-163, `calculate_val_far` :166-178, `evaluate` :181-199.  The metric
part is not pinned by running the reference (no golden could be produced); it is pinned by analytic
known-answer cases in tests/test_verification.py (separable pairs -> accuracy 1, val 1, far 0;
hand-counted confusion matrices).  sklearn's KFold is the reference's own dependency and is called exactly as it calls it; scipy's
interp1d(kind='slinear') is restated (slinear_first_order) because current scipy rejects the duplicate
abscissae the reference feeds it.
"""
import numpy as np
from sklearn.model_selection import KFold
import sklearn.preprocessing

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


def embed_protocol(model, x):
    """verification.py:259-301: embeddings of the batch and of its horizontal flip, summed, normalised."""
    import torch
    with torch.no_grad():
        f1 = model(x)[0]
        f2 = model(x.flip(3))[0]
    return sklearn.preprocessing.normalize((f1 + f2).double().numpy())


def calculate_accuracy(threshold, dist, actual_issame):
    predict_issame = np.less(dist, threshold)
    tp = np.sum(np.logical_and(predict_issame, actual_issame))
    fp = np.sum(np.logical_and(predict_issame, np.logical_not(actual_issame)))
    tn = np.sum(np.logical_and(np.logical_not(predict_issame), np.logical_not(actual_issame)))
    fn = np.sum(np.logical_and(np.logical_not(predict_issame), actual_issame))
    tpr = 0 if (tp + fn == 0) else float(tp) / float(tp + fn)
    fpr = 0 if (fp + tn == 0) else float(fp) / float(fp + tn)
    acc = float(tp + tn) / dist.size
    return tpr, fpr, acc


def calculate_roc(thresholds, embeddings1, embeddings2, actual_issame, nrof_folds=10):
    nrof_pairs = min(len(actual_issame), embeddings1.shape[0])
    nthr = len(thresholds)
    tprs = np.zeros((nrof_folds, nthr))
    fprs = np.zeros((nrof_folds, nthr))
    accuracy = np.zeros((nrof_folds))
    indices = np.arange(nrof_pairs)
    dist = np.sum(np.square(np.subtract(embeddings1, embeddings2)), 1)
    for fold_idx, (train_set, test_set) in enumerate(KFold(n_splits=nrof_folds, shuffle=False).split(indices)):
        acc_train = np.zeros((nthr))
        for k, threshold in enumerate(thresholds):
            _, _, acc_train[k] = calculate_accuracy(threshold, dist[train_set], actual_issame[train_set])
        best = np.argmax(acc_train)
        for k, threshold in enumerate(thresholds):
            tprs[fold_idx, k], fprs[fold_idx, k], _ = calculate_accuracy(threshold, dist[test_set],
                                                                         actual_issame[test_set])
        _, _, accuracy[fold_idx] = calculate_accuracy(thresholds[best], dist[test_set], actual_issame[test_set])
    return np.mean(tprs, 0), np.mean(fprs, 0), accuracy


def calculate_val_far(threshold, dist, actual_issame):
    predict_issame = np.less(dist, threshold)
    true_accept = np.sum(np.logical_and(predict_issame, actual_issame))
    false_accept = np.sum(np.logical_and(predict_issame, np.logical_not(actual_issame)))
    n_same = np.sum(actual_issame)
    n_diff = np.sum(np.logical_not(actual_issame))
    return float(true_accept) / float(n_same), float(false_accept) / float(n_diff)


def calculate_val(thresholds, embeddings1, embeddings2, actual_issame, far_target, nrof_folds=10):
    nrof_pairs = min(len(actual_issame), embeddings1.shape[0])
    nthr = len(thresholds)
    val = np.zeros(nrof_folds)
    far = np.zeros(nrof_folds)
    dist = np.sum(np.square(np.subtract(embeddings1, embeddings2)), 1)
    indices = np.arange(nrof_pairs)
    for fold_idx, (train_set, test_set) in enumerate(KFold(n_splits=nrof_folds, shuffle=False).split(indices)):
        far_train = np.zeros(nthr)
        for k, threshold in enumerate(thresholds):
            _, far_train[k] = calculate_val_far(threshold, dist[train_set], actual_issame[train_set])
        if np.max(far_train) >= far_target:
            threshold = slinear_first_order(far_train, thresholds, far_target)     # verification.py:152-153
        else:
            threshold = 0.0
        val[fold_idx], far[fold_idx] = calculate_val_far(threshold, dist[test_set], actual_issame[test_set])
    return np.mean(val), np.std(val), np.mean(far)


def evaluate(embeddings, actual_issame, nrof_folds=10):
    """embeddings: [2 * n_pairs][E] already L2-normalised (rows 2i, 2i+1 = pair i)."""
    e1, e2 = embeddings[0::2], embeddings[1::2]
    issame = np.asarray(actual_issame)
    tpr, fpr, accuracy = calculate_roc(np.arange(0, 4, 0.01), e1, e2, issame, nrof_folds)
    val, val_std, far = calculate_val(np.arange(0, 4, 0.001), e1, e2, issame, 1e-3, nrof_folds)
    return tpr, fpr, accuracy, val, val_std, far
