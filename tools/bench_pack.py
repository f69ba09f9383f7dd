#!/usr/bin/env python
"""Time of the per-step batched weight repack (ops.PACKS.refresh) for the bench model."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msml_amd import ops  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

if __name__ == "__main__":
    args = argparse.Namespace(frb="iresnet50", batch=64, classes=85742, dtype="bf16", mode="train")
    tr = bench.Trainer(args, 0, 0, 1)
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    print("entries %d, refresh %.1f us" % (len(ops.PACKS.entries), timeit(ops.PACKS.refresh, 20) * 1e6))
