#!/usr/bin/env python
"""Which torch (aten) ops run inside one training step, and from where: finds stray copies /
fills / adds that are not msml_amd kernels.   python tools/trace_ops.py"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = argparse.Namespace(frb="iresnet50", batch=256, classes=85742, dtype="bf16", mode="train", emulate_world=1,
                              data="resident")
    tr = bench.Trainer(args, 0, 0, 1)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        tr.step()
        torch.cuda.synchronize()
    rows = prof.key_averages(group_by_stack_n=8)
    keep = [r for r in rows if r.key.startswith("aten::") and any(k in r.key for k in
            ("copy_", "add", "fill_", "zero_", "zeros", "contiguous", "clone", "mul", "sum", "to", "cat", "index", "slice"))]
    keep.sort(key=lambda r: -r.count)
    for r in keep[:60]:
        st = [s for s in r.stack if "msml_amd" in s or "bench.py" in s][:3]
        print("%-28s n=%4d cpu_us=%8.0f  %s" % (r.key, r.count, r.cpu_time_total, " <- ".join(s.split("/")[-1] for s in st)))


if __name__ == "__main__":
    main()
