#!/usr/bin/env python
"""Microbenchmark: BatchNorm apply + conv (+ weight gradient) with the activation materialised
(msml_bn_act_fwd -> msml_conv2d / msml_conv_wgrad) vs applied in LDS inside the halo-tile kernels
(msml_conv2d_bnin / msml_conv_wgrad_bnin), batch 256."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

SHAPES = [(256, 256, 14), (128, 128, 28), (128, 256, 28), (64, 64, 56), (64, 64, 112)]


def main():
    n = 256
    for cin, cout, h in SHAPES:
        x = torch.randn(n, h, h, cin, device="cuda").bfloat16()
        dy = torch.randn(n, h, h, cout, device="cuda").bfloat16()
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        coef = torch.rand(2, cin, device="cuda") + 0.5
        alpha = torch.rand(cin, device="cuda") * 0.3
        wp = ops.pack_weight(w, False, cin, 0, _lib.BF16)
        act = torch.empty_like(x)
        dw = torch.zeros(cout, cin, 3, 3, device="cuda")
        m = n * h * h
        t_bn = timeit(lambda: _lib.call("msml_bn_act_fwd", x, coef[0], coef[1], alpha, None, 0, act, m, cin, _lib.BF16))
        t_conv = timeit(lambda: ops.conv2d(act, None, wp, None, cout, 3, 3, 1, 1, 1, False, want_stats=True))
        t_fconv = timeit(lambda: ops.conv2d_bnin(x, coef, alpha, wp, cout))
        t_wg = timeit(lambda: ops.conv_wgrad(dy, act, dw, cout, cin, cin, 0, 3, 3, 1, 1, 1, accumulate=True))
        t_fwg = timeit(lambda: ops.conv_wgrad_bnin(dy, x, coef, alpha, dw, cout, cin, cin, 0, accumulate=True))
        print("%4d->%4d @%3d  bn %6.1f us  conv %6.1f -> bnin %6.1f | wgrad %6.1f -> bnin %6.1f | total %6.1f -> %6.1f"
              % (cin, cout, h, t_bn * 1e6, t_conv * 1e6, t_fconv * 1e6, t_wg * 1e6, t_fwg * 1e6,
                 (t_bn + t_conv + t_wg) * 1e6, (t_fconv + t_fwg) * 1e6))


if __name__ == "__main__":
    main()
