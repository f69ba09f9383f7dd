#!/usr/bin/env python
"""Microbenchmark of the accumulator-mode BatchNorm prologue (msml_conv2d_bnin_acc: coefficients from the producer's f64
sums, in-LDS transform, write-through) against the two launches it replaces (msml_bn_fin_act_fwd + msml_conv2d_acc), batch
256, cold operands (every launch on another buffer set, ~2 GB rotation, as in the step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402

SHAPES = [(64, 64, 56), (64, 64, 112), (128, 128, 28), (256, 256, 14)]


def main():
    n = 256
    for cin, cout, h in SHAPES:
        nbuf = max(2, int(2.0e9 // (n * h * h * cin * 2 * 3)))
        xs = [torch.randn(n, h, h, cin, device="cuda").bfloat16() for _ in range(nbuf)]
        acts = [torch.empty_like(xs[0]) for _ in range(nbuf)]
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        gamma, beta = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
        alpha = torch.rand(cin, device="cuda") * 0.3
        rm, rv = torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")
        wp = ops.pack_weight(w, False, cin, 0, _lib.BF16)
        m = n * h * h
        acc = ops.stats_acc(cin, xs[0].device)
        _lib.call("msml_bn_stats_acc", xs[0], m, cin, acc, _lib.BF16)
        coef = torch.empty(4, cin, device="cuda")
        kind = _lib.value("msml_conv2d_bnin_acc_applies", cin, cout, n, h, h, h, h, 3, 3, 1, 1, 1)

        def two(i):
            _lib.call("msml_bn_fin_act_fwd", acc, float(m), gamma, beta, rm, rv, 0.1, 1e-5, coef[0], coef[1], coef[2], coef[3],
                      xs[i], alpha, None, 0, acts[i], m, cin, None, _lib.BF16)
            ops.conv2d(acts[i], None, wp, None, cout, 3, 3, 1, 1, 1, False, want_stats=True)

        def one(i):
            ops.conv2d_bnin_acc(xs[i], acc, (gamma, beta, rm, rv, 0.1, 1e-5), alpha, wp, cout)

        def conv_only(i):
            ops.conv2d(acts[i], None, wp, None, cout, 3, 3, 1, 1, 1, False, want_stats=True)

        res = []
        for fn in (two, one, conv_only):
            for i in range(nbuf):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            e0.record()
            for _ in range(reps):
                for i in range(nbuf):
                    fn(i)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / (reps * nbuf) * 1e3)
        print("%4d->%4d @%3d (kind %d, %d buffer sets)  bn + conv %7.1f us  ->  one launch %7.1f us   (conv alone %7.1f us)"
              % (cin, cout, h, kind, nbuf, res[0], res[1], res[2]))


if __name__ == "__main__":
    main()
