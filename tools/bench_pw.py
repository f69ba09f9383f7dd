#!/usr/bin/env python
"""Cold-data rate of the byte-bound conv launches of the step (1x1 convs of the FM bottlenecks / downsample paths, the
im2col'd stems, the 64-channel stride-2 3x3 layers): GB/s over the ALGORITHMIC bytes (input once + output once) with
the operands rotated through > 1.5 GB of buffers, so that neither L2 nor the 256 MB last-level cache serves a repeat --
next to a device copy of the same bytes (what the memory system gives a plain stream on this box).
    python tools/bench_pw.py [--batch 256]          (MSML_PW_CONV=all / =0: the pointwise kernel everywhere / nowhere)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402

# (cin, cout, H, R, stride, transposed, launches per step)
SHAPES = [
    (32, 64, 112, 1, 1, False, 1), (32, 64, 56, 1, 1, False, 3), (64, 32, 56, 1, 1, False, 2),
    (64, 128, 28, 1, 1, False, 2), (128, 64, 28, 1, 1, False, 2), (128, 256, 14, 1, 1, False, 2), (256, 128, 14, 1, 1, False, 2),
    (64, 32, 56, 1, 1, True, 2), (32, 64, 56, 1, 1, True, 2), (128, 64, 28, 1, 1, True, 2), (256, 128, 14, 1, 1, True, 2), (128, 256, 14, 1, 1, True, 2),
    (64, 64, 112, 1, 2, False, 1), (64, 64, 112, 3, 2, False, 1), (64, 64, 112, 3, 2, True, 1),
    (32, 32, 56, 3, 1, False, 2), (32, 32, 56, 3, 1, True, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--fused", default="", choices=["", "add", "bnb"],
                    help="1x1 backward-data launches with another gradient added / the BatchNorm backward sums in the epilogue")
    args = ap.parse_args()
    n = args.batch
    dt = _lib.BF16
    print("%-34s %9s %9s %9s %9s" % ("launch", "MB", "us", "GB/s", "copy GB/s"))
    for cin, cout, h, r, stride, tr, cnt in SHAPES:
        pad = r // 2
        p = (h + 2 * pad - r) // stride + 1
        if tr:          # backward-data of a cin -> cout conv over an h x h input: dy [p x p x cout] -> dx [h x h x cin]
            in_shape, cin_k, cout_k = (n, p, p, cout), cout, cin
        else:
            in_shape, cin_k, cout_k = (n, h, h, cin), cin, cout
        out_px = h * h if tr else p * p
        nbytes = 2 * (in_shape[0] * in_shape[1] * in_shape[2] * in_shape[3] + n * out_px * cout_k)
        if stride == 2 and not tr and r == 1:
            nbytes = 2 * (n * p * p * cin + n * p * p * cout)       # a 1x1 stride-2 conv reads a quarter of its input
        sets = max(2, int(1.6e9 // nbytes) + 1)
        xs = [torch.randn(in_shape, device="cuda").to(torch.bfloat16) for _ in range(sets)]
        w = torch.randn(cout, cin, r, r, device="cuda") * 0.05
        wp = ops.pack_weight(w, tr, cout if tr else cin, 0, dt)

        fused = tr and stride == 1 and args.fused and (r == 1 or args.fused == "bnb")      # the step's forms of the backward-data launches
        if fused:
            others = [torch.randn(n, h, h, cin, device="cuda").to(torch.bfloat16) for _ in range(sets)]
            coef = torch.rand(4, cin, device="cuda") + 0.5
            alpha = torch.full((cin,), 0.25, device="cuda")
            ones, zeros = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
            nbytes += 2 * n * h * h * cin                             # the second operand of the epilogue

        def run(i):
            if fused and args.fused == "add":
                out = torch.empty(n, h, h, cin, dtype=torch.bfloat16, device="cuda")
                _lib.call("msml_conv2d_fused", xs[i % sets], cout, None, 0, wp, wp.shape[0], ones, zeros, None,
                          others[i % sets], 0, out, cin, n, h, h, h, h, 1, 1, 1, 0, 0, 1)
                return out
            if fused:
                return ops.conv_dgrad_bnbwd(xs[i % sets], wp, cin, r, r, 1, pad, pad, h, h, others[i % sets], coef, alpha)
            if tr:
                return ops.conv2d(xs[i % sets], None, wp, None, cin, r, r, stride, pad, pad, True, p=h, q=h)
            return ops.conv2d(xs[i % sets], None, wp, None, cout, r, r, stride, pad, pad, False, want_stats=True)
        run(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters * sets):
            run(i + 1)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / (args.iters * sets)
        # the same bytes as a device copy (half read, half written)
        src = [torch.empty(nbytes // 4, dtype=torch.bfloat16, device="cuda") for _ in range(sets)]
        dst = torch.empty(nbytes // 4, dtype=torch.bfloat16, device="cuda")
        dst.copy_(src[0])
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.iters * sets):
            dst.copy_(src[i % sets])
        e1.record()
        torch.cuda.synchronize()
        tc = e0.elapsed_time(e1) * 1e-3 / (args.iters * sets)
        name = "%s %d->%d @%d k%d s%d x%d" % (("T+" + args.fused if fused else "T") if tr else "N", cin, cout, h, r, stride, cnt)
        print("%-34s %9.1f %9.1f %9.0f %9.0f" % (name, nbytes / 1e6, t * 1e6, nbytes / t / 1e9, nbytes / tc / 1e9))
        del xs, src, dst
        if fused:
            del others


if __name__ == "__main__":
    main()
