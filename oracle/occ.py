"""CPU restatement of the device input pipeline (test oracle; not used by the product).

Restates, for numpy arrays, the reference's per-sample augmentation (datasets/load_dataset.py:101-139):
RandomRect (datasets/augment/rand_occ.py:103-139), RandomEllipse (:148-203; analytic ellipse instead of
cv2.ellipse, which is not installed here), RandomConnectedPolygon (:217-322; even-odd point-in-polygon test on the
integer lattice instead of cv2.fillPoly), NoneOcc (:80-90), RandomBlock (:43-72), the random flip
(load_dataset.py:119-123), _add_gauss_to_face (:183-201) with _get_gauss (:282-339) and ToTensor +
Normalize(0.5, 0.5).  The reference's dataset module cannot be imported in the build container
(mxnet, cv2, torchvision absent), so this part is parity-unpinned by execution: the geometry is pinned
by the draws of msml_amd/synthetic.py (same formulas, tests/test_occ.py) and by hand-checked cases.

The random draws are the counter-based generator of csrc/occ.hip (splitmix64), restated with numpy
uint64 arithmetic so that the CPU regenerates exactly the batch the GPU drew.

Texture occluders (RandomGlasses / RandomGlassesList rand_occ.py:337-428, RandomScarf :431-517, RandomRealObject
:520-600): `sets` = [(kind, rgba[num, h0, w0, 4] uint8), ...] as the reference's constructors preload them.  Their
per-sample `Image.resize` is restated from Pillow's resampling code (RGBA -> premultiplied RGBa, separable bicubic with
22-bit fixed-point coefficients, back to straight alpha): resize_rgba below, which tests/test_occ.py checks BIT FOR BIT
against PIL.Image.resize itself -- the library the reference calls -- so this part of the oracle is pinned to the
reference's own arithmetic even though rand_occ.py cannot be imported (cv2).
"""
import numpy as np

OCC_NONE, OCC_RECT, OCC_ELLIPSE, OCC_BLOCK, OCC_POLY, OCC_GLASSES, OCC_SCARF, OCC_OBJECT = 0, 1, 2, 3, 4, 5, 6, 7
KIND_OF = {"glasses": OCC_GLASSES, "scarf": OCC_SCARF, "object": OCC_OBJECT}
DESC_WORDS = 64
M64 = (1 << 64) - 1
f32 = np.float32


def _mix(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def u32(seed, img, k):
    return (_mix((_mix(seed & M64) + img * 64 + k) & M64) >> 32) & 0xFFFFFFFF


def randint(u, a, b):
    return a + ((u * (b - a)) >> 32)


def unif(u):
    return f32(u >> 8) * f32(1.0 / 16777216.0)


def sincos(a):
    """occ_sincos of csrc/occ.hip, operation by operation in f32 (quadrant reduction + Taylor polynomials)."""
    a = f32(a)
    q = int(f32(f32(a * f32(0.63661977236758134)) + f32(0.5)))
    r = f32(f32(a - f32(f32(q) * f32(1.5707963705062866))) - f32(f32(q) * f32(-4.371138828673793e-08)))
    r2 = f32(r * r)
    sp = f32(f32(-1.9841270114e-04) + f32(r2 * f32(2.7557314297e-06)))
    sp = f32(f32(8.3333337680e-03) + f32(r2 * sp))
    sp = f32(f32(-1.6666667163e-01) + f32(r2 * sp))
    sp = f32(r + f32(r * f32(r2 * sp)))
    cp = f32(f32(-1.3888889225e-03) + f32(r2 * f32(f32(2.4801587642e-05) + f32(r2 * f32(-2.7557314297e-07)))))
    cp = f32(f32(4.1666667908e-02) + f32(r2 * cp))
    cp = f32(f32(-0.5) + f32(r2 * cp))
    cp = f32(f32(1.0) + f32(r2 * cp))
    m = q & 3
    s = sp if m == 0 else (cp if m == 1 else (f32(-sp) if m == 2 else f32(-cp)))
    c = cp if m == 0 else (f32(-sp) if m == 1 else (f32(-cp) if m == 2 else sp))
    return s, c


# ------------------------------------------------------------------ PIL Image.resize, restated
def _bicubic(x):
    """Pillow Resample.c bicubic_filter (a = -0.5), in double."""
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resize_coeffs(insz, outsz):
    """precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the bicubic filter (support 2):
    [(first tap, taps, [22-bit fixed-point coefficients])] per output coordinate."""
    scale = insz / outsz
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    rows = []
    for xx in range(outsz):
        center = (xx + 0.5) * scale
        ss = 1.0 / fscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), insz) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        rows.append((xmin, xmax, [int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in k]))
    return rows


def _resample_axis(a, axis, outsz):
    a = np.moveaxis(a, axis, 0)
    if a.shape[0] == outsz:                        # ImagingResample: need_horizontal / need_vertical false
        return np.moveaxis(a, 0, axis)
    out = np.empty((outsz,) + a.shape[1:], np.uint8)
    for xx, (xmin, cnt, k) in enumerate(resize_coeffs(a.shape[0], outsz)):
        ss = np.full(a.shape[1:], 1 << 21, np.int64)
        for j in range(cnt):
            ss += a[xmin + j].astype(np.int64) * k[j]
        out[xx] = np.clip(ss >> 22, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_rgba(rgba, w, h):
    """PIL.Image.fromarray(rgba, 'RGBA').resize((w, h)) (default resampling = bicubic): unchanged size -> copy;
    else RGBA -> RGBa (MULDIV255), horizontal pass, vertical pass, RGBa -> RGBA (255 * c / a)."""
    if rgba.shape[0] == h and rgba.shape[1] == w:
        return rgba.copy()
    a = rgba[..., 3:4].astype(np.uint32)
    t = rgba[..., :3].astype(np.uint32) * a + 128
    p = rgba.copy()
    p[..., :3] = (((t >> 8) + t) >> 8).astype(np.uint8)
    p = _resample_axis(p, 1, w)
    p = _resample_axis(p, 0, h)
    al = p[..., 3].astype(np.int64)
    m = (al != 255) & (al != 0)
    o = p.copy()
    for c in range(3):
        v = p[..., c].astype(np.int64)
        o[..., c] = np.clip(np.where(m, (255 * v) // np.maximum(al, 1), v), 0, 255).astype(np.uint8)
    return o


def size_range(kind, h0, w0, h, w):
    """(wmin, wmax, hmin, hmax): resampled sizes a set can draw (one pixel of slack each way)."""
    if kind == OCC_GLASSES:
        bw, bh = w * (w0 / 120.0), h * (h0 / 120.0)
        lo, hi = 1 / 1.1, 1.1
    elif kind == OCC_SCARF:
        bw, bh, lo, hi = float(w0), float(h0), 1 / 1.1, 1.0
    else:
        bw, bh, lo, hi = float(w0), float(h0), 1.0, 2.0
    return (max(int(bw * lo) - 1, 1), int(bw * hi) + 1, max(int(bh * lo) - 1, 1), int(bh * hi) + 1)


def _pick_set(sets, kind, u):
    idx = [i for i, (k, _) in enumerate(sets) if k == kind]
    if not idx:
        return -1
    return idx[randint(u, 0, len(idx))]


def draw(seed, offset, n, h, w, mode, lo=0, hi=36, flip=True, sets=()):
    """desc[n][64] int32, the same words as msml_occ_draw / msml_occ_draw_tex (sets: [(kind, rgba)])."""
    out = np.zeros((n, DESC_WORDS), np.int32)
    for i in range(n):
        img = offset + i
        d = out[i]
        kind = OCC_NONE
        if mode == 0:
            pick = randint(u32(seed, img, 0), 0, 4)
            kind = (OCC_RECT, OCC_ELLIPSE, OCC_POLY, OCC_NONE)[pick]
        elif mode == 1:
            kind = OCC_RECT
        elif mode == 2:
            kind = OCC_BLOCK
        elif mode == 4:
            kind = OCC_POLY
        elif mode in (5, 6):
            if mode == 5:
                six = randint(u32(seed, img, 0), 0, 7)
            else:
                six = randint(u32(seed, img, 13), 0, 6) if randint(u32(seed, img, 0), 0, 8) >= 4 else 6
            kind = (OCC_RECT, OCC_ELLIPSE, OCC_POLY, OCC_GLASSES, OCC_SCARF, OCC_OBJECT, OCC_NONE)[six]
        elif mode in (7, 8, 9):
            kind = OCC_GLASSES + (mode - 7)
        if kind >= OCC_GLASSES:
            si = _pick_set(sets, kind, u32(seed, img, 1))
            if si < 0:
                kind = OCC_NONE
            else:
                rgba = sets[si][1]
                h0, w0 = f32(rgba.shape[1]), f32(rgba.shape[2])
                u3, u4 = unif(u32(seed, img, 3)), unif(u32(seed, img, 4))
                one = f32(1.0)
                if kind == OCC_GLASSES:                # rand_occ.py:371-387
                    bw = f32(f32(w) * f32(w0 / f32(120.0)))
                    bh = f32(f32(h) * f32(h0 / f32(120.0)))
                    lo_s = f32(one / f32(1.1))
                    ow = int(f32(bw * f32(lo_s + f32(f32(f32(1.1) - lo_s) * u3))))
                    oh = int(f32(bh * f32(lo_s + f32(f32(f32(1.1) - lo_s) * u4))))
                    x0 = int(f32(f32(f32(0.12) + f32(f32(randint(u32(seed, img, 5), -5, 6)) * f32(0.02))) * f32(w)))
                    y0 = int(f32(f32(f32(0.3) + f32(f32(randint(u32(seed, img, 6), -5, 6)) * f32(0.01))) * f32(h)))
                elif kind == OCC_SCARF:                # :466-479
                    lo_s = f32(one / f32(1.1))
                    ow = int(f32(w0 * f32(lo_s + f32(f32(one - lo_s) * u3))))
                    oh = int(f32(h0 * f32(lo_s + f32(f32(one - lo_s) * u4))))
                    x0 = int(f32(f32(f32(0.1) + f32(f32(randint(u32(seed, img, 5), -5, 5)) * f32(0.01))) * f32(w)))
                    y0 = int(f32(f32(f32(0.6) + f32(f32(randint(u32(seed, img, 6), -5, 5)) * f32(0.01))) * f32(w)))
                else:                                  # :563-575
                    ow = int(f32(w0 * f32(one + f32(f32(f32(2.0) - one) * u3))))
                    oh = int(f32(h0 * f32(one + f32(f32(f32(2.0) - one) * u4))))
                    x0 = int(f32(f32(f32(randint(u32(seed, img, 5), 15, 51)) * f32(0.01)) * f32(w)))
                    y0 = int(f32(f32(f32(randint(u32(seed, img, 6), 15, 51)) * f32(0.01)) * f32(h)))
                wmin, wmax, hmin, hmax = size_range(kind, rgba.shape[1], rgba.shape[2], h, w)
                ow, oh = min(max(ow, wmin), wmax), min(max(oh, hmin), hmax)
                d[1], d[2], d[3], d[4] = x0, y0, ow, oh
                d[13] = si
                d[14] = randint(u32(seed, img, 2), 0, rgba.shape[0])
        if kind == OCC_RECT:
            pct = randint(u32(seed, img, 1), lo, hi)
            ratio = f32(pct) * f32(0.01)
            area = int(f32(w * h) * ratio)
            ow = randint(u32(seed, img, 2), int(f32(w) * ratio) + 1, w + 1)
            oh = area // ow
            d[1] = randint(u32(seed, img, 3), 0, w - ow + 1)
            d[2] = randint(u32(seed, img, 4), 0, h - oh + 1)
            d[3], d[4] = ow, oh
            for c in range(3):
                d[5 + c] = randint(u32(seed, img, 5 + c), 0, 256)
            if oh == 0:
                kind = OCC_NONE
        elif kind == OCC_ELLIPSE:
            ch = randint(u32(seed, img, 1), h // 5, 4 * h // 5)
            cw = randint(u32(seed, img, 2), w // 5, 4 * w // 5)
            mh = min(ch, h - ch)
            ah = randint(u32(seed, img, 3), 20, mh if mh > 20 else 21)
            ratio = f32(0.2) + (f32(0.4) - f32(0.2)) * unif(u32(seed, img, 4))
            aw = int(f32(h * w) * ratio / (f32(3.14) * f32(ah)))
            d[1], d[2], d[3], d[4] = cw, ch, aw, ah
            for c in range(3):
                d[5 + c] = randint(u32(seed, img, 5 + c), 1, 256)
        elif kind == OCC_POLY:                     # rand_occ.py:262-322
            cnt = randint(u32(seed, img, 1), 4, 11)
            cx = randint(u32(seed, img, 2), h // 5, 4 * h // 5)
            cy = randint(u32(seed, img, 3), w // 5, 4 * w // 5)
            big = randint(u32(seed, img, 4), h // 5, int(f32(1.3) * f32(h)) // 5)
            small = f32(f32(big) / f32(f32(1.3) + f32(f32(f32(2.6) - f32(1.3)) * unif(u32(seed, img, 12)))))
            step = f32(f32(6.2831854820251465) / f32(cnt))
            ab, as_ = f32(0), f32(0)
            verts = [(int(f32(f32(cx) + f32(big))), cy)]
            for i in range(cnt):
                ab = f32(ab + f32(step * f32(f32(0.7) + f32(f32(f32(1.3) - f32(0.7)) * unif(u32(seed, img, 16 + 3 * i))))))
                sn, cs = sincos(ab)
                verts.append((int(f32(f32(cx) + f32(f32(big) * cs))), int(f32(f32(cy) + f32(f32(big) * sn)))))
                if unif(u32(seed, img, 17 + 3 * i)) > f32(0.5):
                    as_ = f32(as_ + f32(step * f32(f32(0.6) + f32(f32(f32(1.4) - f32(0.6)) * unif(u32(seed, img, 18 + 3 * i))))))
                    sn, cs = sincos(as_)
                    verts.append((int(f32(f32(cx) + f32(small * cs))), int(f32(f32(cy) + f32(small * sn)))))
            d[12] = len(verts)
            for v, (vx, vy) in enumerate(verts):
                d[16 + 2 * v], d[17 + 2 * v] = vx, vy
            for c in range(3):
                d[5 + c] = randint(u32(seed, img, 5 + c), 1, 256)
        elif kind == OCC_BLOCK:
            pct = randint(u32(seed, img, 1), lo, hi)
            ratio = f32(pct) * f32(0.01)
            bw = int(np.sqrt(ratio * f32(w) * f32(w), dtype=f32))
            if pct == 0 or bw == 0:
                kind = OCC_NONE
            else:
                d[1] = randint(u32(seed, img, 2), 0, w - bw + 1)
                d[2] = randint(u32(seed, img, 3), 0, w - bw + 1)
                d[3] = d[4] = bw
        d[0] = kind
        d[8] = int(flip and randint(u32(seed, img, 8), 1, 11) >= 5)
        fl = np.array([f32(w) * unif(u32(seed, img, 9)), f32(h) * unif(u32(seed, img, 10)),
                       f32(0.7) + (f32(1.4) - f32(0.7)) * unif(u32(seed, img, 11))], f32)
        d[9:12] = fl.view(np.int32)
    return out


def inside(d, h, w):
    """Boolean (h, w) occlusion region in SOURCE coordinates."""
    ys, xs = np.mgrid[0:h, 0:w]
    if d[0] in (OCC_RECT, OCC_BLOCK):
        return (xs >= d[1]) & (xs < d[1] + d[3]) & (ys >= d[2]) & (ys < d[2] + d[4])
    if d[0] == OCC_ELLIPSE:
        dx, dy = (xs - d[1]).astype(f32), (ys - d[2]).astype(f32)
        aw, ah = f32(d[3]), f32(d[4])
        return dx * dx * ah * ah + dy * dy * aw * aw <= aw * aw * ah * ah
    if d[0] == OCC_POLY:                           # even-odd rule, exact integer arithmetic (csrc/occ.hip occ_inside)
        nv = int(d[12])
        ins = np.zeros((h, w), bool)
        xs, ys = xs.astype(np.int64), ys.astype(np.int64)
        j = nv - 1
        for i in range(nv):
            xi, yi, xj, yj = int(d[16 + 2 * i]), int(d[17 + 2 * i]), int(d[16 + 2 * j]), int(d[17 + 2 * j])
            cross = (yi > ys) != (yj > ys)
            dyv = yj - yi
            lhs, rhs = (xs - xi) * dyv, (xj - xi) * (ys - yi)
            ins ^= cross & ((lhs < rhs) if dyv > 0 else (lhs > rhs))
            j = i
        return ins
    return np.zeros((h, w), bool)


def light_map(d, h, w):
    """_get_gauss (Euclidean, radius 128, int16-truncated offsets) * scale, as a float16 map."""
    lc = d[9:12].view(f32)
    ix = (np.arange(w, dtype=f32) - lc[0]).astype(np.int16).astype(np.int32)
    iy = (np.arange(h, dtype=f32) - lc[1]).astype(np.int16).astype(np.int32)
    dist = np.sqrt((ix[None, :] ** 2 + iy[:, None] ** 2).astype(f32))
    g = np.exp(f32(-0.5) * (dist * dist) / f32(16384.0)).astype(np.float16)
    return (g * np.float16(lc[2])).astype(np.float16).astype(f32)


def patch_of(d, sets):
    """The resampled RGBA occluder of a texture descriptor."""
    return resize_rgba(sets[int(d[13])][1][int(d[14])], int(d[3]), int(d[4]))


def paste(pix, d, sets):
    """Texture kinds: paste the resampled occluder into `pix` (h, w, 3) in place, return the occluded region.
    Glasses replace the face where alpha > 10 (rand_occ.py:390), scarf / object where alpha != 0 (:495, :591); the mask
    marks alpha != 0 for all three (:400-401, :504-505, :600-601); cropped at the image border (:486-489, :582-585)."""
    h, w, _ = pix.shape
    pt = patch_of(d, sets)
    x0, y0 = int(d[1]), int(d[2])
    ph, pw = min(pt.shape[0], h - y0), min(pt.shape[1], w - x0)
    occ = np.zeros((h, w), bool)
    if ph <= 0 or pw <= 0:
        return occ
    pt = pt[:ph, :pw]
    alpha = pt[..., 3]
    keep = alpha > 10 if d[0] == OCC_GLASSES else alpha != 0
    region = pix[y0:y0 + ph, x0:x0 + pw]
    region[keep] = pt[..., :3][keep]
    occ[y0:y0 + ph, x0:x0 + pw] = alpha != 0
    return occ


def apply(src, desc, light=True, want_ori=True, sets=()):
    """src: (n, h, w, 3) uint8 -> img (n, 3, h, w) f32, msk (n, h, w) int64, ori (n, 3, h, w) f32."""
    n, h, w, _ = src.shape
    img = np.empty((n, 3, h, w), f32)
    ori = np.empty((n, 3, h, w), f32) if want_ori else None
    msk = np.empty((n, h, w), np.int64)
    for i in range(n):
        d = desc[i]
        occ = inside(d, h, w)
        pix = src[i].copy()
        if d[0] >= OCC_GLASSES:
            occ = paste(pix, d, sets)
        elif d[0] == OCC_BLOCK:
            pix[occ] = 0
        elif d[0] != OCC_NONE:
            pix[occ] = d[5:8].astype(np.uint8)
        clean = src[i]
        if d[8]:                                   # occlude -> flip
            pix, occ, clean = pix[:, ::-1], occ[:, ::-1], clean[:, ::-1]
        t = pix.astype(f32) / f32(255.0)           # ToTensor
        if light:
            t = t * light_map(d, h, w)[:, :, None]
            t = t / t.max()
        img[i] = ((t - f32(0.5)) / f32(0.5)).transpose(2, 0, 1)
        if want_ori:
            ori[i] = ((clean.astype(f32) / f32(255.0) - f32(0.5)) / f32(0.5)).transpose(2, 0, 1)
        msk[i] = np.where(occ, 0, 1)
    return img, msk, ori
