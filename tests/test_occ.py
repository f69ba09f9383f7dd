"""Device input pipeline: the CPU restatement against msml_amd/synthetic.py's geometry and hand-checked
cases (CPU); the HIP kernels against the restatement (GPU): descriptors and masks bit-exact, images to
f32 rounding of exp()."""
import numpy as np
import pytest
import torch

from oracle import occ as oo


def test_oracle_rect_geometry_is_synthetic_py():
    """Given the draws synthetic.rect_occlusion makes (numpy RandomState), the oracle's paste / mask equal
    synthetic.py's: same rectangle, same mask, same colours (u8 domain -> [-1, 1])."""
    from msml_amd import synthetic
    b, h, w = 6, 112, 112
    rng = np.random.RandomState(3)
    src = np.random.RandomState(0).randint(0, 256, (b, h, w, 3)).astype(np.uint8)
    x = torch.from_numpy(((src.astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5))
                         .transpose(0, 3, 1, 2).copy())
    xs, msk_s = synthetic.rect_occlusion(x, seed=3, lo=5, hi=36)
    desc = np.zeros((b, oo.DESC_WORDS), np.int32)
    for i in range(b):                       # the draw order of synthetic.rect_occlusion
        ratio = rng.randint(5, 36) * 0.01
        area = int(w * h * ratio)
        ow = rng.randint(int(w * ratio) + 1, w + 1)
        oh = int(area / ow)
        ox = rng.randint(0, w - ow + 1)
        oy = rng.randint(0, h - oh + 1)
        cols = [rng.randint(0, 256) for _ in range(3)]
        desc[i, :8] = [oo.OCC_RECT, ox, oy, ow, oh] + cols
    img, msk, ori = oo.apply(src, desc, light=False)
    assert np.array_equal(msk, msk_s.numpy())
    assert np.abs(img - xs.numpy()).max() < 1e-6
    assert np.abs(ori - x.numpy()).max() == 0


def test_oracle_hand_cases():
    d = np.zeros(oo.DESC_WORDS, np.int32)
    d[:5] = [oo.OCC_BLOCK, 10, 20, 70, 70]
    occ = oo.inside(d, 112, 112)
    assert occ.sum() == 4900 and occ[20, 10] and occ[89, 79] and not occ[90, 79] and not occ[20, 80]
    d[:5] = [oo.OCC_ELLIPSE, 56, 56, 30, 20]
    e = oo.inside(d, 112, 112)
    assert e[56, 56] and e[56, 86] and not e[56, 87] and e[76, 56] and not e[77, 56]
    assert abs(e.sum() - np.pi * 30 * 20) < 0.03 * np.pi * 30 * 20
    # RandomBlock(40, 41): side int((0.40 * 112^2)^0.5) = 70 (BASELINE config 5)
    desc = oo.draw(5, 0, 8, 112, 112, 2, 40, 41, flip=False)
    assert (desc[:, 0] == oo.OCC_BLOCK).all() and (desc[:, 3] == 70).all()
    assert (desc[:, 1] >= 0).all() and (desc[:, 1] <= 42).all()
    # training mix: every kind appears, rectangles obey RandomRect's constraints
    desc = oo.draw(7, 100, 400, 112, 112, 0)
    kinds = set(desc[:, 0].tolist())
    assert kinds == {0, 1, 2, 4}
    r = desc[desc[:, 0] == oo.OCC_RECT]
    assert (r[:, 1] + r[:, 3] <= 112).all() and (r[:, 2] + r[:, 4] <= 112).all()
    assert (r[:, 3] * r[:, 4] <= 0.36 * 112 * 112).all()
    assert 0.4 < desc[:, 8].mean() < 0.8                     # P{flip} = 6/10 (randint(1, 11) >= 5)
    # RandomConnectedPolygon: a hand-made square, and the drawn stars (rand_occ.py:262-322: 4..10 big-circle points
    # of radius 22..28 around a centre in the middle three fifths, a small-circle point after each with P = 1/2)
    d = np.zeros(oo.DESC_WORDS, np.int32)
    d[0], d[12] = oo.OCC_POLY, 4
    d[16:24] = [10, 10, 50, 10, 50, 40, 10, 40]
    sq = oo.inside(d, 112, 112)
    assert sq[10, 10] and sq[39, 49] and not sq[40, 20] and not sq[20, 50] and sq.sum() == 40 * 30
    s, c = oo.sincos(np.float32(2.5))
    assert abs(float(s) - np.sin(2.5)) < 2e-6 and abs(float(c) - np.cos(2.5)) < 2e-6
    assert all(abs(float(oo.sincos(np.float32(a))[0]) - np.sin(np.float32(a))) < 2e-6 for a in np.linspace(0, 9, 200))
    p = desc[desc[:, 0] == oo.OCC_POLY]
    assert len(p) > 50 and (p[:, 12] >= 5).all() and (p[:, 12] <= 21).all()
    for dd in p[:40]:
        v = dd[16:16 + 2 * dd[12]].reshape(-1, 2)
        r = np.hypot(v[:, 0] - v[0, 0], v[:, 1] - v[0, 1])                   # distances from the first vertex (angle 0)
        assert r.max() <= 2 * 28 + 2                                          # every vertex within the big circle's diameter
        area = oo.inside(dd, 112, 112).sum()
        assert 150 < area < np.pi * 29 * 29                                   # a star inside the big circle


@pytest.mark.gpu
@pytest.mark.parametrize("mode,lo,hi", [("train", 0, 36), ("rect", 0, 36), ("block", 40, 41), ("none", 0, 36),
                                        ("polygon", 0, 36)])
def test_device_pipeline_matches_oracle(mode, lo, hi):
    from msml_amd import data
    n = 64
    src = np.random.RandomState(11).randint(0, 256, (n, 112, 112, 3)).astype(np.uint8)
    dsrc = torch.from_numpy(src).cuda()
    for light in (False, True):
        img, msk, ori, desc = data.augment(dsrc, 1234, 5000, mode, lo, hi, True, light, True)
        ref_desc = oo.draw(1234, 5000, n, 112, 112, data.MODES[mode], lo, hi, True)
        assert np.array_equal(desc.cpu().numpy(), ref_desc)                      # integer draws: exact
        rimg, rmsk, rori = oo.apply(src, ref_desc, light)
        assert np.array_equal(msk.cpu().numpy(), rmsk)                           # masks: exact
        assert np.array_equal(ori.cpu().numpy(), rori)
        err = np.abs(img.cpu().numpy() - rimg).max()
        assert err < (2e-3 if light else 1e-7), err                              # exp() + float16 map
        assert img.min() >= -1 - 1e-6 and img.max() <= 1 + 1e-6


@pytest.mark.gpu
def test_device_loader_semantics():
    """DeviceLoaderX: batches arrive in order, reproducible from (seed, batch index), independent of the
    prefetch timing; labels travel with their images."""
    from msml_amd import data
    src = data.SynthFaceSource(32, 1000, steps=5, pool=3, seed=9)
    seen = []
    for img, msk, ori, lab in data.DeviceLoaderX(src, 0, seed=77, mode="train"):
        seen.append((img.clone(), msk.clone(), lab.clone()))
        torch.cuda.synchronize()
    assert len(seen) == 5
    for k, (img, msk, lab) in enumerate(seen):
        faces, labels = src.pool[k % 3]
        assert torch.equal(lab.cpu(), labels)
        ref_desc = oo.draw(77, k * 32, 32, 112, 112, 0)
        rimg, rmsk, _ = oo.apply(faces.numpy(), ref_desc, True, False)
        assert np.array_equal(msk.cpu().numpy(), rmsk)
        assert np.abs(img.cpu().numpy() - rimg).max() < 2e-3
