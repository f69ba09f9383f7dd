#!/usr/bin/env python
"""cProfile of the host side of one eager multi-stream training step (where the ~24 ms of Python issue time go).
    python tools/host_profile.py [N lines]"""
import argparse
import cProfile
import os
import pstats
import sys

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
    pr = cProfile.Profile()
    for _ in range(5):
        torch.cuda.synchronize()
        pr.enable()
        tr.step()
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
    st.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
