#!/usr/bin/env python
"""Achieved GB/s of the element-wise BatchNorm kernels at the tensor sizes of the step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib  # noqa: E402
from msml_amd._lib import call  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402


def main():
    n = 256
    for c, h in ((64, 112), (64, 56), (128, 28), (256, 14), (512, 7), (32, 56)):
        m = n * h * h
        x = torch.randn(m, c, device="cuda").bfloat16()
        dy = torch.randn(m, c, device="cuda").bfloat16()
        y = torch.empty_like(x)
        coef = torch.rand(6, c, device="cuda") + 0.5
        part = torch.zeros(32, 3, c, device="cuda")
        cw = torch.empty(98 * c, device="cuda")
        pg = torch.zeros(3, c, device="cuda")
        t1 = timeit(lambda: call("msml_bn_act_fwd", x, coef[0], coef[1], coef[2], None, 0, y, m, c, 1), 20)
        t2 = timeit(lambda: call("msml_bn_act_fwd", x, coef[0], coef[1], None, dy, 0, y, m, c, 1), 20)
        t3 = timeit(lambda: call("msml_bn_act_bwd_apply", dy, x, coef[0], coef[1], coef[2], coef[3], coef[4], part, 32, None,
                                 y, pg[0], pg[1], pg[2], 0, m, c, cw, 1), 20)
        t4 = timeit(lambda: call("msml_bn_act_bwd_apply", dy, x, coef[0], coef[1], None, coef[3], coef[4], part, 32, dy,
                                 y, pg[0], pg[1], None, 0, m, c, cw, 1), 20)
        b = m * c * 2
        print("C %3d @%3d (%6.1f MB): fwd %5.1f us %5.0f GB/s | fwd+res %5.1f us %5.0f | bwd_apply %5.1f us %5.0f | bwd_apply+add %5.1f us %5.0f"
              % (c, h, b / 1e6, t1 * 1e6, 2 * b / t1 / 1e9, t2 * 1e6, 3 * b / t2 / 1e9, t3 * 1e6, 3 * b / t3 / 1e9,
                 t4 * 1e6, 4 * b / t4 / 1e9))


if __name__ == "__main__":
    main()
