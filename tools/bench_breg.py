#!/usr/bin/env python
"""Interleaved A/B of the halo conv with its weights in registers (MSML_HALO_BREG=1, read per call) against the LDS weight
rings, batch 256, random operands: plain forward, forward with the BatchNorm prologue (accumulator mode), backward-data with
the fused BatchNorm sums; the outputs of the two variants are compared bit for bit on the way.
Needs the experiment build (MSML_LIB=variants/libmsml_MSML_EXPERIMENTS.so): the shipped library ignores the switch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402

SHAPES = [(256, 256, 14), (256, 512, 14), (128, 256, 28)]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n = 256
    if not _lib.value("msml_has_experiments"):
        sys.exit("tools/bench_breg.py: build and load the experiment library (tools/build_variant.py --all MSML_EXPERIMENTS)")
    ops.ACC_STATS = True
    for cin, cout, h in SHAPES:
        x = torch.randn(n, h, h, cin, device="cuda").bfloat16()
        dy = torch.randn(n, h, h, cout, device="cuda").bfloat16()
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        wp = ops.pack_weight(w, False, cin, 0, _lib.BF16)
        wpt = ops.pack_weight(w, True, cout, 0, _lib.BF16)
        gamma, beta = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
        alpha = torch.rand(cin, device="cuda") * 0.3
        rm, rv = torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")
        m = n * h * h
        acc = ops.stats_acc(cin, x.device)
        _lib.call("msml_bn_stats_acc", x, m, cin, acc, _lib.BF16)
        coef = torch.empty(4, cin, device="cuda")
        act = torch.empty_like(x)
        k = ops.rows4(coef)
        _lib.call("msml_bn_fin_act_fwd", acc, float(m), gamma, beta, rm, rv, 0.1, 1e-5, k[0], k[1], k[2], k[3], x, alpha, None, 0,
                  act, m, cin, None, _lib.BF16)
        fns = {
            "fwd": lambda: ops.conv2d(x, None, wp, None, cout, 3, 3, 1, 1, 1, False, want_stats=True)[0],
            "fwd+bn": lambda: ops.conv2d_bnin_acc(x, acc, (gamma, beta, rm, rv, 0.1, 1e-5), alpha, wp, cout)[2],
            "dgrad+bnb": lambda: ops.conv_dgrad_bnbwd(dy, wpt, cin, 3, 3, 1, 1, 1, h, h, x, coef, alpha)[0],
        }
        for name, fn in fns.items():
            outs = {}
            for v in ("0", "1"):
                os.environ["MSML_HALO_BREG"] = v
                outs[v] = fn().clone()
            same = torch.equal(outs["0"], outs["1"])
            t = {"0": [], "1": []}
            for _ in range(4):
                for v in ("0", "1"):
                    os.environ["MSML_HALO_BREG"] = v
                    t[v].append(timed(fn, 30))
            flops = 2.0 * n * h * h * cin * cout * 9
            print("%4d->%4d @%2d %-10s lds ring %s us   registers %s us   (%.0f -> %.0f TFLOP/s)  bit-identical: %s"
                  % (cin, cout, h, name, " ".join("%6.1f" % u for u in t["0"]), " ".join("%6.1f" % u for u in t["1"]),
                     flops / min(t["0"]) / 1e6, flops / min(t["1"]) / 1e6, same))
    os.environ.pop("MSML_HALO_BREG", None)


if __name__ == "__main__":
    main()
