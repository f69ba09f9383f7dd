"""Verification protocol and metrics: the CPU restatement against analytic known-answer cases (CPU),
the HIP path against the restatement (GPU)."""
import numpy as np
import pytest
import torch

from oracle import verification as ov


def _pairs(n_pairs, e, seed, sep):
    """Synthetic pairs: `same` pairs are a unit vector and a small perturbation of it, `different`
    pairs two independent vectors; sep scales the perturbation."""
    rng = np.random.RandomState(seed)
    a = rng.randn(n_pairs, e)
    issame = rng.rand(n_pairs) < 0.5
    other = rng.randn(n_pairs, e) if sep < 0.5 else 0.6 * a + rng.randn(n_pairs, e)   # sep >= 0.5: overlapping classes
    b = np.where(issame[:, None], a + sep * rng.randn(n_pairs, e), other)
    emb = np.empty((2 * n_pairs, e), np.float32)
    emb[0::2], emb[1::2] = a, b
    return emb, issame


def test_oracle_known_answers():
    """Separable pairs: same-pair distance ~0.02, different-pair distance ~2 -> every fold picks a
    threshold in between: accuracy 1, TAR 1 at FAR 0.  Hand-counted confusion matrix for one threshold."""
    emb, issame = _pairs(200, 64, 0, 0.01)
    import sklearn.preprocessing
    en = sklearn.preprocessing.normalize(emb.astype(np.float64))
    tpr, fpr, acc, val, val_std, far = ov.evaluate(en, issame, 10)
    # (FAR 1e-3 is below the resolution 1/90 of a train split with ~90 different pairs: the interpolated
    # threshold sits just above the closest different pair of the train split, far stays ~1 pair)
    assert np.all(acc == 1.0) and val == 1.0 and val_std == 0.0 and far <= 0.02
    # the duplicate-abscissa interpolation: last sample of the run <= xq, first sample of the next run
    assert ov.slinear_first_order([0, 0, 0, 0.5, 0.5, 1], [0, 1, 2, 3, 4, 5], 0.25) == 2.5
    assert ov.slinear_first_order([0, 0, 0, 0.5, 0.5, 1], [0, 1, 2, 3, 4, 5], 0.75) == 4.5
    assert tpr[0] == 0.0 and fpr[0] == 0.0 and tpr[-1] == 1.0 and fpr[-1] == 1.0     # thresholds 0 and 3.99
    dist = np.array([0.1, 0.5, 0.2, 0.9, 0.3, 0.05])
    same = np.array([True, True, False, False, True, False])
    t, f, a = ov.calculate_accuracy(0.25, dist, same)          # predicted same: 0.1, 0.2, 0.05
    assert (t, f, a) == (1 / 3, 2 / 3, 2 / 6)
    v, fa = ov.calculate_val_far(0.25, dist, same)
    assert (v, fa) == (1 / 3, 2 / 3)


def test_oracle_against_an_independent_vectorised_restatement():
    """Second, independently written check of the (execution-unpinned) metric oracle: the same protocol computed a
    different way -- distances sorted once, confusion counts from np.searchsorted on the sorted same / different
    distances of every fold (no per-threshold loop, no KFold object, no boolean masks), threshold selection by
    argmax / interpolation on those counts -- must give the same accuracies, TPR / FPR curves, val and far."""
    emb, issame = _pairs(1503, 32, 4, 1.2)
    import sklearn.preprocessing
    en = sklearn.preprocessing.normalize(emb.astype(np.float64))
    tpr, fpr, acc, val, val_std, far = ov.evaluate(en, issame, 10)
    d = ((en[0::2] - en[1::2]) ** 2).sum(1)
    n, k = len(d), 10
    edges = np.cumsum([0] + [n // k + (1 if i < n % k else 0) for i in range(k)])

    def counts(idx, thr):
        ds, dd = np.sort(d[idx][issame[idx]]), np.sort(d[idx][~issame[idx]])
        return np.searchsorted(ds, thr, "left"), np.searchsorted(dd, thr, "left"), len(ds), len(dd)
    thr = np.arange(0, 4, 0.01)
    thr_v = np.arange(0, 4, 0.001)
    acc2, tprs, fprs, vals, fars = [], [], [], [], []
    for f in range(k):
        test = np.arange(edges[f], edges[f + 1])
        train = np.concatenate((np.arange(0, edges[f]), np.arange(edges[f + 1], n)))
        tp, fp, ns, nd = counts(train, thr)
        best = np.argmax((tp + (nd - fp)) / float(len(train)))
        tp, fp, ns, nd = counts(test, thr)
        tprs.append(tp / ns)
        fprs.append(fp / nd)
        acc2.append((tp[best] + nd - fp[best]) / float(len(test)))
        _, fpv, _, ndv = counts(train, thr_v)
        far_train = fpv / ndv
        t = ov.slinear_first_order(far_train, thr_v, 1e-3) if far_train.max() >= 1e-3 else 0.0
        tpt, fpt, nst, ndt = counts(test, np.array([t]))
        vals.append(tpt[0] / nst)
        fars.append(fpt[0] / ndt)
    assert np.allclose(acc, acc2, rtol=0, atol=1e-15)
    assert np.allclose(tpr, np.mean(tprs, 0), atol=1e-15) and np.allclose(fpr, np.mean(fprs, 0), atol=1e-15)
    assert abs(val - np.mean(vals)) < 1e-15 and abs(far - np.mean(fars)) < 1e-15 and abs(val_std - np.std(vals)) < 1e-15
    assert 0.6 < np.mean(acc) < 1.0          # a non-trivial operating point


def test_oracle_fold_sizes_match_device_rule():
    """KFold(shuffle=False) test folds are contiguous with the first n % k folds one longer -- the rule
    k_pair_hist uses to assign a pair to its fold."""
    from sklearn.model_selection import KFold
    for n, k in ((103, 10), (6000, 10), (17, 4)):
        base, rem = n // k, n % k
        starts = np.cumsum([0] + [base + (1 if i < rem else 0) for i in range(k)])
        for f, (_, test) in enumerate(KFold(n_splits=k, shuffle=False).split(np.arange(n))):
            assert test[0] == starts[f] and test[-1] == starts[f + 1] - 1


@pytest.mark.gpu
@pytest.mark.parametrize("n_pairs,sep", [(600, 1.0), (1003, 1.5), (6000, 1.2)])
def test_device_metrics_match_oracle(n_pairs, sep):
    """msml_pair_sqdist + msml_pair_hist + the prefix-sum arithmetic == the restated calculate_roc /
    calculate_val on overlapping (non-separable) pairs: accuracies, val and far are ratios of
    integer counts and must match exactly; tpr / fpr curves exactly."""
    import sklearn.preprocessing
    from msml_amd import verification as hv
    emb, issame = _pairs(n_pairs, 64, n_pairs, sep)
    en = sklearn.preprocessing.normalize(emb.astype(np.float64))
    ref = ov.evaluate(en, issame, 10)
    got = hv.evaluate(torch.from_numpy(emb).cuda(), issame, 10)
    d = hv.pair_sqdist(torch.from_numpy(emb).cuda()).cpu().numpy()
    dref = np.sum(np.square(en[0::2] - en[1::2]), 1)
    assert np.abs(d - dref).max() < 1e-13
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[2], ref[2])
    assert got[3] == ref[3] and abs(got[4] - ref[4]) < 1e-15 and got[5] == ref[5]
    assert 0.5 < ref[2].mean() < 1.0          # the case is not degenerate


@pytest.mark.gpu
def test_embedding_protocol_config5():
    """BASELINE config 5 protocol at bs 8 against the CPU oracle: orig + flip sum, L2-normalise, cosine
    of occluded pairs (fp16=True in its default split-bf16 precision: cosines within 1e-4)."""
    from msml_amd import synthetic
    from msml_amd import verification as hv
    from msml_amd.backbones import MSML
    from oracle import model as om
    from oracle.fill import fill_module
    peer = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
    kw = dict(fm_params=(3, 2, "sigmoid", "mul"), header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0))
    torch.manual_seed(0)
    m = fill_module(MSML("iresnet18", "unet", (1, 1, 1, 1), 8, fp16=True, peer_params=peer, **kw)).cuda().eval()
    o = fill_module(om.MSML("iresnet18", "unet", (1, 1, 1, 1), 8, **kw)).eval()
    a, b, _ = synthetic.occluded_pairs(4, seed=7)
    x = torch.stack((a, b), 1).reshape(8, 3, 112, 112)          # rows 2i, 2i+1 = pair i
    eo = ov.embed_protocol(o, x)
    cos_o = np.sum(eo[0::2] * eo[1::2], 1)
    emb = hv.extract_embeddings(m, x.cuda())
    cos_h = hv.pair_cosine(emb).cpu().numpy()
    en = torch.nn.functional.normalize(emb.double()).cpu().numpy()
    assert np.abs(en - eo).max() < 1e-4
    assert np.abs(cos_h - cos_o).max() < 1e-4
    assert np.all(cos_o < 0.9999) and np.all(cos_o > -1)


@pytest.mark.gpu
def test_embedding_protocol_bs1024_properties():
    """Config 5 at its full batch (ires50, bs 1024 = 512 occluded pairs), split-bf16: the result of an
    image does not depend on the batch (chunked execution is exact), flipping twice is the identity of
    the protocol (sum is symmetric), cosines are in [-1, 1] and clean/occluded pairs correlate."""
    from msml_amd import synthetic
    from msml_amd import verification as hv
    from msml_amd.backbones import MSML
    peer = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
    torch.manual_seed(0)
    m = MSML("iresnet50", "unet", (1, 1, 1, 1), 8, fp16=True, fm_params=(3, 2, "sigmoid", "mul"),
             header_type="AMArcFace", peer_params=peer).cuda().eval()
    a, b, _ = synthetic.occluded_pairs(512, seed=3)
    x = torch.stack((a, b), 1).reshape(1024, 3, 112, 112).cuda()
    emb = hv.extract_embeddings(m, x)
    assert emb.shape == (1024, 512) and torch.isfinite(emb).all()
    sub = hv.extract_embeddings(m, x[500:508].contiguous())
    assert torch.equal(emb[500:508], sub)
    embf = hv.extract_embeddings(m, x.flip(3))
    assert torch.equal(emb, embf)                     # (f(x) + f(flip x)) is flip-symmetric, bit for bit
    cos = hv.pair_cosine(emb).cpu().numpy()
    assert cos.min() >= -1 - 1e-12 and cos.max() <= 1 + 1e-12
