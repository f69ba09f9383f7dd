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

MODES = {"train": 0, "rect": 1, "block": 2, "none": 3, "polygon": 4}
DESC_WORDS = 64


def draw(n, seed, offset, mode="train", lo=0, hi=36, flip=True, size=112, device="cuda"):
    """Per-image occlusion / flip / light descriptors (msml_occ_draw), int32 [n, 64] on the device."""
    desc = torch.empty(n, DESC_WORDS, dtype=torch.int32, device=device)
    call("msml_occ_draw", int(seed), int(offset), n, size, size, MODES[mode], lo, hi, int(flip), desc)
    return desc


def apply(src, desc, light=True, want_ori=True):
    """src: (N, H, W, 3) uint8 on the device -> img (N,3,H,W) f32, msk (N,H,W) int64, ori or None."""
    n, h, w, c = src.shape
    assert c == 3 and src.dtype == torch.uint8 and src.is_cuda and src.is_contiguous()
    img = torch.empty(n, 3, h, w, dtype=torch.float32, device=src.device)
    ori = torch.empty(n, 3, h, w, dtype=torch.float32, device=src.device) if want_ori else None
    msk = torch.empty(n, h, w, dtype=torch.int64, device=src.device)
    call("msml_occ_apply", src, desc, img, msk, ori, n, h, w, int(light))
    return img, msk, ori


def augment(src, seed, offset, mode="train", lo=0, hi=36, flip=True, light=True, want_ori=True):
    desc = draw(src.shape[0], seed, offset, mode, lo, hi, flip, src.shape[1], src.device)
    return apply(src, desc, light, want_ori) + (desc,)


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
                 want_ori=True, max_prefetch=6):
        self.source, self.local_rank = source, local_rank
        self.stream = torch.cuda.Stream(local_rank)
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
            img, msk, ori, _ = augment(src, seed, self.count * faces.shape[0], mode, lo, hi, flip, light, want_ori)
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
