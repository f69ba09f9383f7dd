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
"""
import numpy as np

OCC_NONE, OCC_RECT, OCC_ELLIPSE, OCC_BLOCK, OCC_POLY = 0, 1, 2, 3, 4
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


def draw(seed, offset, n, h, w, mode, lo=0, hi=36, flip=True):
    """desc[n][64] int32, the same words as msml_occ_draw."""
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


def apply(src, desc, light=True, want_ori=True):
    """src: (n, h, w, 3) uint8 -> img (n, 3, h, w) f32, msk (n, h, w) int64, ori (n, 3, h, w) f32."""
    n, h, w, _ = src.shape
    img = np.empty((n, 3, h, w), f32)
    ori = np.empty((n, 3, h, w), f32) if want_ori else None
    msk = np.empty((n, h, w), np.int64)
    for i in range(n):
        d = desc[i]
        occ = inside(d, h, w)
        pix = src[i].copy()
        if d[0] == OCC_BLOCK:
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
