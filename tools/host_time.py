#!/usr/bin/env python
"""Host issue time of one eager multi-stream training step (no synchronisation inside the loop)
against its GPU wall time: how close the eager path is to being host-bound."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msml_amd import ops  # noqa: E402


def main():
    args = argparse.Namespace(frb="iresnet50", batch=256, classes=85742, dtype="bf16", mode="train", emulate_world=1,
                              data="resident")
    tr = bench.Trainer(args, 0, 0, 1)
    ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.step()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10
    print("host issue %.1f ms/step (min %.1f), pipelined wall %.1f ms/step" % (1e3 * sum(host) / len(host), 1e3 * min(host), 1e3 * wall))


if __name__ == "__main__":
    main()
