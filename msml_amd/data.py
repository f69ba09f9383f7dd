"""Device input pipeline: H2D double buffering with DataLoaderX semantics and
occlusion synthesis / flip / Gaussian light / normalisation on the GPU (csrc/occ.hip).

At 7 000 images/s per GPU (56 000 on a node) the reference's per-sample PIL / OpenCV augmentation
(FaceByRandOccMask.__getitem__, datasets/load_dataset.py:101-139; 32 workers, config.py:29) is the next
bottleneck: here the host only hands over decoded uint8 faces (37.6 KB per image, 9.6 MB per batch of
256 -- 0.15 ms of PCIe Gen5) and labels; everything else runs on the device in two kernels per batch.

* `augment(src_u8, seed, offset, mode)`   -- (img, msk, ori) from a uint8 batch already on the device.
* `DeviceLoaderX(source, device, ...)`    -- datasets/dataloaderx.py:40-66 semantics: a background thread
  pulls host batches, the next batch is copied on a side stream from pinned memory and augmented there
  while the current step runs, `__next__` makes the compute stream wait for that stream.
* `SynthFaceSource`                        -- stand-in for the record reader: a pool of random uint8 faces in
  pinned host memory (no dataset travels with this repo).
"""
import queue
import threading

import torch

from ._lib import call

# "train": the asset-free mix {rect, ellipse, polygon, none}; "ms1m" / "casia": the reference's full mixes with the
# texture occluders (load_dataset.py:155-163), which need an OccluderAtlas; "glasses" / "scarf" / "object": one class.
MODES = {"train": 0, "rect": 1, "block": 2, "none": 3, "polygon": 4, "ms1m": 5, "casia": 6, "glasses": 7, "scarf": 8,
         "object": 9}
DESC_WORDS = 64
KINDS = {"glasses": 5, "scarf": 6, "object": 7}
META_WORDS, RT_WORDS = 16, 10


def _bicubic(x):
    x = abs(x)
    if x < 1.0:
        return (1.5 * x - 2.5) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * -0.5
    return 0.0


def resample_table(insz, outsz):
    """Pillow's bicubic resampling coefficients for one axis (Resample.c precompute_coeffs + normalize_coeffs_8bpc,
    what Image.resize -- the call of rand_occ.py:375,466,562 -- uses): int32 [outsz][10] = first tap, taps, up to 8
    coefficients in 22-bit fixed point.  Double precision on the host, exactly as the library computes them."""
    import numpy as np
    scale = insz / outsz
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    tab = np.zeros((outsz, RT_WORDS), np.int32)
    for xx in range(outsz):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        cnt = min(int(center + support + 0.5), insz) - xmin
        if cnt > RT_WORDS - 2:
            raise ValueError("resample_table: %d -> %d needs %d taps (at most %d)" % (insz, outsz, cnt, RT_WORDS - 2))
        k = [_bicubic((x + xmin - center + 0.5) / fscale) for x in range(cnt)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        tab[xx, 0], tab[xx, 1] = xmin, cnt
        tab[xx, 2:2 + cnt] = [int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in k]
    return tab


class OccluderAtlas:
    """Texture occluders on the device.  sets: [(kind, rgba)], kind in {'glasses', 'scarf', 'object'}, rgba a uint8
    array / tensor [num, h0, w0, 4] preloaded as the reference's constructors do (load_occluder_sets reads the
    reference's folders).  Holds the atlas bytes, the set table and the resampling tables for every size a sample
    can draw at image size `size` (rand_occ.py:371-374, 464-465, 560-561)."""

    def __init__(self, sets, size=112, device="cuda"):
        import numpy as np
        if not 1 <= len(sets) <= 16:
            raise ValueError("OccluderAtlas: 1..16 occluder sets")
        self.size = size
        blobs, meta, dirs, tabs = [], np.zeros((len(sets), META_WORDS), np.int32), [], []
        off, rt_off, self.patch_bytes, self.lds_bytes = 0, 0, 0, 0
        cache = {}

        def table_dir(insz, lo, hi):
            nonlocal rt_off
            base = len(dirs)
            for out in range(lo, hi + 1):
                key = (insz, out)
                if key not in cache:
                    cache[key] = rt_off
                    tabs.append(resample_table(insz, out).reshape(-1))
                    rt_off += out * RT_WORDS
                dirs.append(cache[key])
            return base
        for i, (kind, rgba) in enumerate(sets):
            a = np.ascontiguousarray(rgba.cpu().numpy() if torch.is_tensor(rgba) else rgba, dtype=np.uint8)
            if a.ndim != 4 or a.shape[3] != 4 or a.shape[0] < 1:
                raise ValueError("OccluderAtlas: set %d must be [num, h0, w0, 4] uint8" % i)
            num, h0, w0, _ = a.shape
            k = KINDS[kind]
            if k == 5:                                   # the glasses scale with the image (rand_occ.py:371-372)
                bw, bh, lo, hi = size * (w0 / 120.0), size * (h0 / 120.0), 1 / 1.1, 1.1
            elif k == 6:
                bw, bh, lo, hi = float(w0), float(h0), 1 / 1.1, 1.0
            else:
                bw, bh, lo, hi = float(w0), float(h0), 1.0, 2.0
            wmin, wmax = max(int(bw * lo) - 1, 1), int(bw * hi) + 1
            hmin, hmax = max(int(bh * lo) - 1, 1), int(bh * hi) + 1
            meta[i, :11] = [off, num, h0, w0, k, wmin, wmax, hmin, hmax, table_dir(w0, wmin, wmax),
                            table_dir(h0, hmin, hmax)]
            blobs.append(a.reshape(-1))
            off += a.size
            self.patch_bytes = max(self.patch_bytes, hmax * wmax * 4, h0 * w0 * 4)
            self.lds_bytes = max(self.lds_bytes, (h0 * w0 + h0 * wmax) * 4)
        if self.lds_bytes > 160 * 1024:
            raise ValueError("OccluderAtlas: an occluder set needs %d bytes of LDS (160 KB available)" % self.lds_bytes)
        self.nsets = len(sets)
        self.atlas = torch.from_numpy(np.concatenate(blobs)).to(device)
        self.meta = torch.from_numpy(meta).to(device)
        self.dir = torch.tensor(dirs, dtype=torch.int32, device=device)
        self.rtab = torch.from_numpy(np.concatenate(tabs)).to(device)


def load_occluder_sets(root):
    """Preload the reference's occluder folders at run time, from the USER's checkout (nothing is copied into this
    package): `root` = <reference>/datasets/augment/occluder.  Same order and arithmetic as the training mix of
    datasets/load_dataset.py:72-85: RandomGlassesList(glasses_crop, eleglasses_crop) with RandomGlasses.__init__
    (rand_occ.py:345-366: RGBA, resize to 80 x 40), RandomScarf (:441-462: 90 x 90), RandomRealObject (:531-556:
    rescale by max(w / 55, h / 55), centre crop to 55 x 55)."""
    import os
    import numpy as np
    from PIL import Image

    def folder(name, fn):
        d = os.path.join(root, name)
        return np.stack([fn(Image.open(os.path.join(d, f)).convert("RGBA")) for f in os.listdir(d)])

    def fixed(w, h):
        return lambda im: np.array(im.resize((w, h)), dtype=np.uint8)

    def center_crop(im, w=55, h=55):
        fw, fh = im.size
        ratio = max(fw / w, fh / h)
        im = im.resize((int(fw / ratio), int(fh / ratio)))
        # torchvision CenterCrop((55, 55)): pad evenly with zeros when smaller, then crop around the centre
        iw, ih = im.size
        if iw < w or ih < h:
            pl, pt = (w - iw) // 2 if iw < w else 0, (h - ih) // 2 if ih < h else 0
            canvas = Image.new("RGBA", (max(w, iw), max(h, ih)), (0, 0, 0, 0))
            canvas.paste(im, (pl, pt))
            im = canvas
            iw, ih = im.size
        top, left = int(round((ih - h) / 2.0)), int(round((iw - w) / 2.0))
        return np.array(im.crop((left, top, left + w, top + h)), dtype=np.uint8)

    return [("glasses", folder("glasses_crop", fixed(80, 40))), ("glasses", folder("eleglasses_crop", fixed(80, 40))),
            ("scarf", folder("scarf_crop", fixed(90, 90))), ("object", folder("object_train", center_crop))]


def draw(n, seed, offset, mode="train", lo=0, hi=36, flip=True, size=112, device="cuda", atlas=None):
    """Per-image occlusion / flip / light descriptors (msml_occ_draw[_tex]), int32 [n, 64] on the device."""
    desc = torch.empty(n, DESC_WORDS, dtype=torch.int32, device=device)
    if MODES[mode] >= 5:
        if atlas is None:
            raise ValueError("mode %r needs an OccluderAtlas (texture occluders come from the caller's assets)" % mode)
        if atlas.size != size:
            raise ValueError("OccluderAtlas was built for %d-pixel images" % atlas.size)
        call("msml_occ_draw_tex", int(seed), int(offset), n, size, size, MODES[mode], lo, hi, int(flip), atlas.meta,
             atlas.nsets, desc)
    else:
        call("msml_occ_draw", int(seed), int(offset), n, size, size, MODES[mode], lo, hi, int(flip), desc)
    return desc


def apply(src, desc, light=True, want_ori=True, atlas=None):
    """src: (N, H, W, 3) uint8 on the device -> img (N,3,H,W) f32, msk (N,H,W) int64, ori or None.  atlas: the
    OccluderAtlas the descriptors were drawn with (texture kinds are resampled per sample, then pasted)."""
    n, h, w, c = src.shape
    assert c == 3 and src.dtype == torch.uint8 and src.is_cuda and src.is_contiguous()
    img = torch.empty(n, 3, h, w, dtype=torch.float32, device=src.device)
    ori = torch.empty(n, 3, h, w, dtype=torch.float32, device=src.device) if want_ori else None
    msk = torch.empty(n, h, w, dtype=torch.int64, device=src.device)
    if atlas is not None:
        patch = torch.empty(n, atlas.patch_bytes, dtype=torch.uint8, device=src.device)
        call("msml_occ_resize", atlas.atlas, atlas.meta, atlas.dir, atlas.rtab, desc, patch, atlas.patch_bytes, n,
             atlas.lds_bytes)
        call("msml_occ_apply_tex", src, desc, patch, atlas.patch_bytes, img, msk, ori, n, h, w, int(light))
    else:
        call("msml_occ_apply", src, desc, img, msk, ori, n, h, w, int(light))
    return img, msk, ori


def augment(src, seed, offset, mode="train", lo=0, hi=36, flip=True, light=True, want_ori=True, atlas=None):
    desc = draw(src.shape[0], seed, offset, mode, lo, hi, flip, src.shape[1], src.device, atlas)
    return apply(src, desc, light, want_ori, atlas if MODES[mode] >= 5 else None) + (desc,)


class SynthFaceSource:
    """Endless host-side source of (uint8 faces [B,112,112,3] in pinned memory, int64 labels [B]): `pool`
    distinct batches are generated once and cycled, as a record reader with a warm page cache would."""

    def __init__(self, batch, num_classes, steps=None, pool=4, seed=1, size=112):
        self.batch, self.steps = batch, steps
        self.pool = []
        for i in range(pool):
            g = torch.Generator().manual_seed(seed + 977 * i)
            faces = torch.randint(0, 256, (batch, size, size, 3), generator=g, dtype=torch.uint8)
            labels = torch.randint(0, num_classes, (batch,), generator=g)
            if torch.cuda.is_available():
                faces, labels = faces.pin_memory(), labels.pin_memory()
            self.pool.append((faces, labels))

    def __iter__(self):
        i = 0
        while self.steps is None or i < self.steps:
            yield self.pool[i % len(self.pool)]
            i += 1


class _Background(threading.Thread):
    """BackgroundGenerator of datasets/dataloaderx.py:12-37: a daemon thread fills a bounded queue."""

    def __init__(self, generator, device_index, max_prefetch=6):
        super().__init__(daemon=True)
        self.queue = queue.Queue(max_prefetch)
        self.generator = generator
        self.device_index = device_index
        self.start()

    def run(self):
        torch.cuda.set_device(self.device_index)
        for item in self.generator:
            self.queue.put(item)
        self.queue.put(None)

    def __next__(self):
        item = self.queue.get()
        if item is None:
            raise StopIteration
        return item


class DeviceLoaderX:
    """for img, msk, ori, label in DeviceLoaderX(source, local_rank, seed=...): ...

    Batch k+1 is copied (non_blocking, pinned -> HBM) and augmented on `self.stream` while the caller
    trains on batch k; `__next__` first makes the current stream wait for `self.stream`
    (dataloaderx.py:60-66).  The random draws of batch k are keyed by (seed, k * batch + image index):
    reproducible whatever the timing."""

    def __init__(self, source, local_rank=0, seed=1, mode="train", lo=0, hi=36, flip=True, light=True,
                 want_ori=True, max_prefetch=6, atlas=None):
        self.source, self.local_rank = source, local_rank
        self.stream = torch.cuda.Stream(local_rank)
        self.atlas = atlas                   # OccluderAtlas for the "ms1m" / "casia" mixes (texture occluders)
        self.cfg = (seed, mode, lo, hi, flip, light, want_ori)
        self.max_prefetch = max_prefetch
        self.count = 0
        self.batch = None

    def __iter__(self):
        self.iter = _Background(iter(self.source), self.local_rank, self.max_prefetch)
        self.count = 0
        self.preload()
        return self

    def preload(self):
        try:
            host = next(self.iter)
        except StopIteration:
            self.batch = None
            return
        seed, mode, lo, hi, flip, light, want_ori = self.cfg
        faces, label = host[0], host[-1]
        with torch.cuda.stream(self.stream):
            src = faces.to(device=self.local_rank, non_blocking=True)
            lab = label.to(device=self.local_rank, non_blocking=True)
            img, msk, ori, _ = augment(src, seed, self.count * faces.shape[0], mode, lo, hi, flip, light, want_ori,
                                       self.atlas)
        self.count += 1
        self.batch = (img, msk, ori, lab, src)

    def __next__(self):
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.stream)
        batch = self.batch
        if batch is None:
            raise StopIteration
        for t in batch:                      # produced on the side stream, consumed on this one
            if t is not None:
                t.record_stream(cur)
        self.preload()
        img, msk, ori, lab, _ = batch
        return img, msk, ori, lab
