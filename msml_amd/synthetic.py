"""Synthetic faces, occlusions and labels for measurement and parity tests.

No dataset travels with this repo: images are seeded noise in [-1, 1],
occlusions follow the geometry of the reference's training / evaluation occluders, restated
here for tensors instead of PIL images:

* ``rect_occlusion``  -- datasets/augment/rand_occ.py:103-139 (RandomRect): area ratio
  ``randint(lo, hi) %``, width ``randint(int(W*ratio)+1, W+1)``, height ``area // width``,
  uniform position, one constant colour per channel.  Mask convention 0 = occluded, 1 = clean
  (datasets/load_dataset.py:37).
* ``block_occlusion`` -- datasets/augment/rand_occ.py:43-72 (RandomBlock): a square of side
  ``int((ratio * W * W) ** 0.5)`` at a uniform position, filled black (-1 after normalisation).
"""
import numpy as np
import torch


def images(batch, seed=1, size=112):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, size, size, generator=g).clamp_(-1.0, 1.0)


def labels(batch, num_classes, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, num_classes, (batch,), generator=g)


def rect_occlusion(x, seed=1, lo=0, hi=36):
    """Paste one random rectangle per image; returns (x_occluded, mask int64 (B,H,W) in {0,1})."""
    rng = np.random.RandomState(seed)
    b, _, h, w = x.shape
    x = x.clone()
    msk = torch.ones(b, h, w, dtype=torch.int64)
    for i in range(b):
        ratio = rng.randint(lo, hi) * 0.01
        area = int(w * h * ratio)
        ow = rng.randint(int(w * ratio) + 1, w + 1)
        oh = int(area / ow)
        ox = rng.randint(0, w - ow + 1)
        oy = rng.randint(0, h - oh + 1)
        for c in range(3):
            val = rng.randint(0, 256) / 255.0 * 2.0 - 1.0
            x[i, c, oy:oy + oh, ox:ox + ow] = val
        msk[i, oy:oy + oh, ox:ox + ow] = 0
    return x, msk


def block_occlusion(x, seed=1, lo=40, hi=41):
    """Black square covering ``ratio`` of the image (evaluation protocol); returns (x_occ, mask)."""
    rng = np.random.RandomState(seed)
    b, _, h, w = x.shape
    x = x.clone()
    msk = torch.ones(b, h, w, dtype=torch.int64)
    for i in range(b):
        ratio = rng.randint(lo, hi) * 0.01
        if ratio == 0:
            continue
        bw = int((ratio * w * w) ** 0.5)
        ox = rng.randint(0, w - bw + 1)
        oy = rng.randint(0, w - bw + 1)
        x[i, :, oy:oy + bw, ox:ox + bw] = -1.0
        msk[i, oy:oy + bw, ox:ox + bw] = 0
    return x, msk


def occluded_pairs(num_pairs, seed=1, noise=0.05):
    """Config 5 input: image A clean, image B = A + 40 % black block + small noise."""
    a = images(num_pairs, seed)
    g = torch.Generator().manual_seed(seed + 1000)
    b = (a + noise * torch.randn(a.shape, generator=g)).clamp_(-1.0, 1.0)
    b, msk = block_occlusion(b, seed)
    return a, b, msk
