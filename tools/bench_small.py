#!/usr/bin/env python
"""Microbenchmark of the small-map / stride-2 conv launches (forward + statistics, backward-data) at batch 256."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

# (cin, cout, H, stride)
SHAPES = [(512, 512, 4, 1), (512, 512, 7, 1), (256, 256, 7, 1), (128, 128, 7, 1), (512, 512, 14, 2), (256, 256, 14, 2),
          (256, 256, 28, 2), (128, 128, 28, 2), (128, 128, 56, 2), (64, 64, 112, 2), (64, 64, 56, 2)]


def main():
    n = 256
    for cin, cout, h, stride in SHAPES:
        p = (h + 2 - 3) // stride + 1
        x = torch.randn(n, h, h, cin, device="cuda").bfloat16()
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        dy = torch.randn(n, p, p, cout, device="cuda").bfloat16()
        wp = ops.pack_weight(w, False, cin, 0, _lib.BF16)
        wpt = ops.pack_weight(w, True, cout, 0, _lib.BF16)
        flops = 2.0 * n * p * p * cout * cin * 9
        tf = timeit(lambda: ops.conv2d(x, None, wp, None, cout, 3, 3, stride, 1, 1, False, want_stats=True), 20)
        td = timeit(lambda: ops.conv2d(dy, None, wpt, None, cin, 3, 3, stride, 1, 1, True, p=h, q=h), 20)
        kf = _lib.value("msml_conv2d_kernel", cin, 0, cout, n, h, h, p, p, 3, 3, stride, 1, 1, 0, 1, 1, 1).decode()
        print("%3d->%3d @%3d s%d  fwd %6.1f us %6.1f TF/s | dgrad %6.1f us %6.1f TF/s  [%s]"
              % (cin, cout, h, stride, tf * 1e6, flops / tf / 1e12, td * 1e6, flops / td / 1e12, kf.split("<")[0]))


if __name__ == "__main__":
    main()
