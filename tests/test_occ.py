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


# ------------------------------------------------------------------------ texture occluders (glasses / scarf / object)
def synthetic_sets(seed=3):
    """Stand-ins for the reference's occluder folders (no asset travels with this repo): random RGBA entries with the
    shapes the reference's constructors preload (rand_occ.py:345-366, 441-462, 531-556) -- two glasses folders (80 x 40),
    scarves (90 x 90), objects (55 x 55) -- with transparent, faint (alpha <= 10), translucent and opaque regions."""
    rng = np.random.default_rng(seed)

    def entries(num, h, w):
        a = rng.integers(0, 256, (num, h, w, 4), dtype=np.uint8)
        u = rng.random((num, h, w))
        a[..., 3] = np.where(u < 0.35, 0, np.where(u < 0.45, rng.integers(1, 11, (num, h, w)), a[..., 3]))
        a[:, h // 4:h // 2, w // 4:w // 2, 3] = 255
        return a
    return [("glasses", entries(5, 40, 80)), ("glasses", entries(3, 40, 80)), ("scarf", entries(4, 90, 90)),
            ("object", entries(6, 55, 55))]


def oracle_sets(sets):
    return [(oo.KIND_OF[k], a) for k, a in sets]


def test_oracle_resize_is_pil_resize():
    """The oracle's Image.resize restatement against PIL itself -- the library rand_occ.py:375,466,562 calls -- bit for
    bit, over the size ranges the three occluder classes draw (incl. the unchanged-size copy and a one-axis resample)."""
    from PIL import Image
    rng = np.random.default_rng(0)
    cases = [(40, 80, (33, 37, 41), (67, 75, 80, 82)), (90, 90, (81, 86, 90), (81, 89, 90)),
             (55, 55, (55, 56, 83, 110), (55, 77, 109))]
    for h0, w0, hs, ws in cases:
        rgba = rng.integers(0, 256, (h0, w0, 4), dtype=np.uint8)
        rgba[..., 3] = np.where(rng.random((h0, w0)) < 0.4, 0, rgba[..., 3])
        rgba[5:15, 5:30, 3] = 255
        for h in hs:
            for w in ws:
                ref = np.array(Image.fromarray(rgba, mode="RGBA").resize((w, h)))
                assert np.array_equal(oo.resize_rgba(rgba, w, h), ref), (h0, w0, h, w)


def test_resample_tables_of_the_product_equal_the_oracle():
    from msml_amd import data
    for insz, outsz in [(80, 67), (80, 82), (40, 33), (40, 41), (90, 81), (55, 110), (55, 56)]:
        tab = data.resample_table(insz, outsz)
        for xx, (xmin, cnt, k) in enumerate(oo.resize_coeffs(insz, outsz)):
            assert tab[xx, 0] == xmin and tab[xx, 1] == cnt and list(tab[xx, 2:2 + cnt]) == k


def test_oracle_texture_paste_rules():
    """Known answers of the three paste rules (rand_occ.py:390,400-401 / 495,504-505 / 591,600-601) on an
    unscaled 55 x 55 entry: glasses replace where alpha > 10, scarf / object where alpha != 0; the mask marks
    alpha != 0 for all; the paste is cropped at the border."""
    entry = np.zeros((1, 55, 55, 4), np.uint8)
    entry[0, :, :, :3] = 200
    entry[0, :, 0:10, 3] = 0          # transparent
    entry[0, :, 10:20, 3] = 5         # faint: mask yes, glasses pixel no
    entry[0, :, 20:55, 3] = 255
    face = np.full((1, 112, 112, 3), 50, np.uint8)
    for kind in (oo.OCC_GLASSES, oo.OCC_SCARF, oo.OCC_OBJECT):
        d = np.zeros((1, oo.DESC_WORDS), np.int32)
        d[0, :5] = [kind, 80, 30, 55, 55]          # 23 columns past the right border are cropped
        img, msk, _ = oo.apply(face, d, light=False, sets=[(kind, entry)])
        px = np.rint((img[0, 0] * 0.5 + 0.5) * 255).astype(int)
        assert (msk[0, 30:85, 80:90] == 1).all() and (msk[0, 30:85, 90:112] == 0).all()
        assert (msk[0, :30] == 1).all() and (msk[0, 85:] == 1).all() and (msk[0, :, :80] == 1).all()
        assert (px[30:85, 100:112] == 200).all() and (px[30:85, 80:90] == 50).all()
        assert (px[30:85, 90:100] == (50 if kind == oo.OCC_GLASSES else 200)).all()


def test_oracle_texture_draws_follow_the_placement_rules():
    sets = oracle_sets(synthetic_sets())
    d = oo.draw(11, 0, 700, 112, 112, 5, sets=sets)
    assert set(d[:, 0].tolist()) == {0, 1, 2, 4, 5, 6, 7}            # the seven-way ms1m mix
    g, s, o = d[d[:, 0] == oo.OCC_GLASSES], d[d[:, 0] == oo.OCC_SCARF], d[d[:, 0] == oo.OCC_OBJECT]
    assert min(len(g), len(s), len(o)) > 60
    assert set(g[:, 13].tolist()) == {0, 1} and (s[:, 13] == 2).all() and (o[:, 13] == 3).all()
    assert set(g[:, 1].tolist()) <= {int((0.12 + k * 0.02) * 112) for k in range(-5, 6)}
    assert set(g[:, 2].tolist()) <= {int((0.3 + k * 0.01) * 112) + e for k in range(-5, 6) for e in (-1, 0)}
    assert (g[:, 3] >= 66).all() and (g[:, 3] <= 83).all() and (g[:, 4] >= 32).all() and (g[:, 4] <= 42).all()
    assert (s[:, 3] >= 80).all() and (s[:, 3] <= 90).all() and (s[:, 2] >= 60).all() and (s[:, 2] <= 73).all()
    assert (o[:, 3] >= 55).all() and (o[:, 3] <= 110).all() and (o[:, 1] >= 16).all() and (o[:, 1] <= 56).all()
    c = oo.draw(11, 0, 800, 112, 112, 6, sets=sets)                  # casia: half of the images stay clean
    assert 0.42 < (c[:, 0] == 0).mean() < 0.58


def test_geometric_rasters_against_pil_imagedraw():
    """Second, independently written check of the unpinned geometric occluders: PIL.ImageDraw -- a different rasteriser
    (scan conversion that also paints the outline) -- against the oracle's analytic ellipse and even-odd lattice test.
    Ellipses: the oracle's region is a subset of PIL's and PIL's lies within one pixel of it.  Polygons (the drawn stars
    have thin spikes that a lattice test leaves empty and an outline painter draws): every pixel the two disagree on
    lies within one pixel of a polygon edge."""
    from PIL import Image, ImageDraw
    from scipy import ndimage
    desc = oo.draw(5, 0, 300, 112, 112, 0)
    ys, xs = np.mgrid[0:112, 0:112]
    checked = {oo.OCC_ELLIPSE: 0, oo.OCC_POLY: 0}
    for d in desc:
        if d[0] not in checked:
            continue
        mine = oo.inside(d, 112, 112)
        im = Image.new("L", (112, 112), 0)
        dr = ImageDraw.Draw(im)
        if d[0] == oo.OCC_ELLIPSE:
            dr.ellipse([d[1] - d[3], d[2] - d[4], d[1] + d[3], d[2] + d[4]], fill=255)
            theirs = np.array(im) > 0
            assert not (mine & ~theirs).any()
            assert not (theirs & ~ndimage.binary_dilation(mine, iterations=1)).any()
            assert (mine & theirs).sum() >= 0.9 * (mine | theirs).sum()
        else:
            v = d[16:16 + 2 * d[12]].reshape(-1, 2).astype(np.float64)
            dr.polygon([tuple(p) for p in v.tolist()], fill=255)
            theirs = np.array(im) > 0
            diff = mine ^ theirs
            if (mine & theirs).sum() < 0.6 * (mine | theirs).sum():
                continue                               # a star that overlaps itself: the fill rules legitimately differ
            px, py = xs[diff].astype(np.float64), ys[diff].astype(np.float64)
            dist = np.full(px.shape, 1e9)
            for i in range(len(v)):                    # distance of every disputed pixel to the nearest edge
                a, b = v[i], v[(i + 1) % len(v)]
                ab = b - a
                t = np.clip(((px - a[0]) * ab[0] + (py - a[1]) * ab[1]) / max(ab @ ab, 1e-12), 0, 1)
                dist = np.minimum(dist, np.hypot(px - (a[0] + t * ab[0]), py - (a[1] + t * ab[1])))
            assert (dist <= 1.0 + 1e-9).all(), (d[:5], dist.max())
        checked[d[0]] += 1
    assert checked[oo.OCC_ELLIPSE] > 50 and checked[oo.OCC_POLY] > 50, checked


def test_load_occluder_sets_from_a_reference_checkout():
    """Container-only: the loader on the reference's own folders (read at run time from the user's checkout, never
    copied) -- shapes of the preloaded sets, and the oracle's resize against PIL on a real entry."""
    import os
    from PIL import Image
    from msml_amd import data
    root = "/root/reference/datasets/augment/occluder"
    if not os.path.isdir(root):
        pytest.skip("no reference checkout on this machine")
    sets = data.load_occluder_sets(root)
    assert [k for k, _ in sets] == ["glasses", "glasses", "scarf", "object"]
    assert sets[0][1].shape[1:] == (40, 80, 4) and sets[2][1].shape[1:] == (90, 90, 4) and sets[3][1].shape[1:] == (55, 55, 4)
    for k, a in sets:
        assert a.dtype == np.uint8 and a.shape[0] > 5 and (a[..., 3] == 0).any() and (a[..., 3] == 255).any()
    e = sets[3][1][0]
    assert np.array_equal(oo.resize_rgba(e, 93, 71), np.array(Image.fromarray(e, mode="RGBA").resize((93, 71))))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ms1m", "casia", "glasses", "scarf", "object"])
def test_texture_occluders_hip_vs_oracle(mode):
    """The texture occluders on the device against the oracle (itself pinned to PIL's resize): descriptors, masks and
    the unlit images bit-exact (the resampled RGBA patch enters both), lit images to f32 rounding of exp()."""
    from msml_amd import data
    sets = synthetic_sets()
    atlas = data.OccluderAtlas(sets, 112, "cuda")
    osets = oracle_sets(sets)
    n = 96
    src = np.random.RandomState(13).randint(0, 256, (n, 112, 112, 3)).astype(np.uint8)
    dsrc = torch.from_numpy(src).cuda()
    for light in (False, True):
        img, msk, ori, desc = data.augment(dsrc, 4321, 777, mode, 0, 36, True, light, True, atlas)
        ref_desc = oo.draw(4321, 777, n, 112, 112, data.MODES[mode], 0, 36, True, sets=osets)
        assert np.array_equal(desc.cpu().numpy(), ref_desc)
        rimg, rmsk, rori = oo.apply(src, ref_desc, light, sets=osets)
        assert np.array_equal(msk.cpu().numpy(), rmsk)
        assert np.array_equal(ori.cpu().numpy(), rori)
        err = np.abs(img.cpu().numpy() - rimg).max()
        assert err < (2e-3 if light else 1e-7), err
    if mode in ("glasses", "scarf", "object"):
        assert (ref_desc[:, 0] == oo.KIND_OF[mode]).all() and (rmsk == 0).any()


@pytest.mark.gpu
def test_texture_resize_kernel_is_pil_resize():
    """msml_occ_resize alone: every (w', h') of a set's range, against PIL.Image.resize on the same entry."""
    from PIL import Image
    from msml_amd import data
    from msml_amd._lib import call
    sets = synthetic_sets(5)
    atlas = data.OccluderAtlas(sets, 112, "cuda")
    meta = atlas.meta.cpu().numpy()
    for si in (0, 2, 3):
        h0, w0, kind, wmin, wmax, hmin, hmax = (int(meta[si, j]) for j in (2, 3, 4, 5, 6, 7, 8))
        sizes = [(w, h) for w in range(wmin, wmax + 1, 3) for h in range(hmin, hmax + 1, 4)] + [(w0, h0), (w0, hmin), (wmax, h0)]
        sizes = [(w, h) for w, h in sizes if wmin <= w <= wmax and hmin <= h <= hmax]
        desc = np.zeros((len(sizes), oo.DESC_WORDS), np.int32)
        for i, (w, h) in enumerate(sizes):
            desc[i, 0], desc[i, 3], desc[i, 4], desc[i, 13], desc[i, 14] = kind, w, h, si, i % sets[si][1].shape[0]
        patch = torch.zeros(len(sizes), atlas.patch_bytes, dtype=torch.uint8, device="cuda")
        call("msml_occ_resize", atlas.atlas, atlas.meta, atlas.dir, atlas.rtab, torch.from_numpy(desc).cuda(), patch,
             atlas.patch_bytes, len(sizes), atlas.lds_bytes)
        got = patch.cpu().numpy()
        for i, (w, h) in enumerate(sizes):
            ref = np.array(Image.fromarray(sets[si][1][desc[i, 14]], mode="RGBA").resize((w, h)))
            assert np.array_equal(got[i, :h * w * 4].reshape(h, w, 4), ref), (si, w, h)


@pytest.mark.gpu
def test_device_loader_with_texture_atlas():
    """DeviceLoaderX on the reference's full ms1m mix (load_dataset.py:155-157) with an occluder atlas: every batch is
    reproducible from (seed, batch index) and equals the oracle's -- masks bit-exact, images to exp() rounding."""
    from msml_amd import data
    sets = synthetic_sets(9)
    atlas = data.OccluderAtlas(sets, 112, "cuda")
    osets = oracle_sets(sets)
    src = data.SynthFaceSource(24, 500, steps=3, pool=2, seed=4)
    seen = []
    for img, msk, ori, lab in data.DeviceLoaderX(src, 0, seed=55, mode="ms1m", atlas=atlas):
        seen.append((img.clone(), msk.clone(), ori.clone(), lab.clone()))
        torch.cuda.synchronize()
    assert len(seen) == 3
    kinds = set()
    for k, (img, msk, ori, lab) in enumerate(seen):
        faces, labels = src.pool[k % 2]
        assert torch.equal(lab.cpu(), labels)
        ref_desc = oo.draw(55, k * 24, 24, 112, 112, 5, sets=osets)
        kinds |= set(ref_desc[:, 0].tolist())
        rimg, rmsk, rori = oo.apply(faces.numpy(), ref_desc, True, True, sets=osets)
        assert np.array_equal(msk.cpu().numpy(), rmsk)
        assert np.array_equal(ori.cpu().numpy(), rori)
        assert np.abs(img.cpu().numpy() - rimg).max() < 2e-3
    assert kinds & {5, 6, 7}                       # texture occluders were drawn
    with pytest.raises(ValueError):
        data.draw(4, 1, 0, "ms1m")                 # the texture mixes need an atlas
