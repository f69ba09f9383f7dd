#!/usr/bin/env python
"""Per-shape microbenchmark of the conv kernels (fwd / dgrad / wgrad) at the BASELINE batch.
    python tools/bench_conv.py [--batch 256] [--dtype bf16] [--only fwd,dgrad,wgrad]
Prints TFLOP/s per shape class of SURVEY section 2.3 (ires50 multiplicities) and the weighted total."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402

# (cin, cout, H, R, stride, count in ires50-MSML fwd)
SHAPES = [
    (64, 64, 112, 3, 1, 1), (64, 64, 112, 3, 2, 1), (64, 64, 56, 3, 1, 5), (64, 128, 56, 3, 1, 1),
    (128, 128, 56, 3, 2, 1), (128, 128, 28, 3, 1, 6), (128, 256, 28, 3, 1, 1),
    (256, 256, 28, 3, 2, 1), (256, 256, 14, 3, 1, 26), (256, 512, 14, 3, 1, 1),
    (512, 512, 14, 3, 2, 1), (512, 512, 7, 3, 1, 4), (64, 64, 56, 1, 2, 1), (256, 128, 14, 1, 1, 4),
    (128, 128, 14, 3, 1, 2), (128, 256, 14, 1, 1, 2),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--shapes", default="", help="comma-separated indices into SHAPES")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--zeros", action="store_true",
                    help="all-zero operands: the clock the chip holds rises (MI355X_MICROARCH.md DVFS give-back), "
                         "so random-vs-zero tells whether a kernel is clock-limited")
    args = ap.parse_args()
    dt = _lib.BF16 if args.dtype == "bf16" else _lib.F32
    tdt = _lib.TORCH_DTYPE[dt]
    only = args.only.split(",")
    tot = {k: [0.0, 0.0] for k in only}
    print("%-28s %s" % ("shape", "  ".join("%12s" % k for k in only)))
    shapes = SHAPES if not args.shapes else [SHAPES[int(i)] for i in args.shapes.split(",")]
    for cin, cout, h, r, stride, cnt in shapes:
        n = args.batch
        pad = r // 2
        p = (h + 2 * pad - r) // stride + 1
        x = torch.randn(n, h, h, cin, device="cuda").to(tdt)
        w = torch.randn(cout, cin, r, r, device="cuda") * 0.05
        dy = torch.randn(n, p, p, cout, device="cuda").to(tdt)
        if args.zeros:
            x.zero_(); w.zero_(); dy.zero_()
        wp = ops.pack_weight(w, False, cin, 0, dt)
        wpt = ops.pack_weight(w, True, cout, 0, dt)
        dw = torch.empty_like(w)
        flops = 2.0 * n * p * p * cout * cin * r * r
        fns = {
            "fwd": lambda: ops.conv2d(x, None, wp, None, cout, r, r, stride, pad, pad, False, want_stats=True),
            "dgrad": lambda: ops.conv2d(dy, None, wpt, None, cin, r, r, stride, pad, pad, True, p=h, q=h),
            "wgrad": lambda: ops.conv_wgrad(dy, x, dw, cout, cin, cin, 0, r, r, stride, pad, pad),
        }
        res = []
        for k in only:
            t = timeit(fns[k], args.iters)
            res.append("%7.1f TF %5.0fus" % (flops / t / 1e12, t * 1e6))
            tot[k][0] += flops * cnt
            tot[k][1] += t * cnt
        print("%-28s %s" % ("%d->%d @%d k%d s%d x%d" % (cin, cout, h, r, stride, cnt), "  ".join(res)))
    print("weighted: " + "  ".join("%s %.1f TF/s (%.2f ms)" % (k, v[0] / v[1] / 1e12, v[1] * 1e3) for k, v in tot.items()))


if __name__ == "__main__":
    main()
