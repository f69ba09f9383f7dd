#!/usr/bin/env python
"""Microbenchmark: backward-data conv + BatchNorm backward, unfused vs fused reduce
(msml_conv2d + msml_bn_act_bwd  vs  msml_conv2d_bnbwd + msml_bn_act_bwd_apply) at batch 256."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

SHAPES = [(256, 256, 14, 1), (128, 128, 28, 1), (64, 64, 56, 1), (64, 64, 112, 1), (512, 512, 7, 1),
          (64, 64, 112, 2), (128, 128, 56, 2), (256, 256, 28, 2)]


def main():
    n = 256
    for k, c, h, stride in SHAPES:
        ho = (h + 2 - 3) // stride + 1
        dy = torch.randn(n, ho, ho, k, device="cuda").bfloat16()
        w = torch.randn(k, c, 3, 3, device="cuda") * 0.05
        x = torch.randn(n, h, h, c, device="cuda").bfloat16()
        coef = torch.rand(4, c, device="cuda") + 0.5
        alpha = torch.rand(c, device="cuda") * 0.3
        wp = ops.pack_weight(w, True, k, 0, _lib.BF16)
        m = n * h * h
        pg = torch.zeros(3, c, device="cuda")
        rows = ops.bn_stats_rows(m, c)
        ws = torch.empty(rows * 3 * c + 2 * c, device="cuda")
        cw = torch.empty(98 * c, device="cuda")
        dxb = torch.empty_like(x)
        t_conv = timeit(lambda: ops.conv2d(dy, None, wp, None, c, 3, 3, stride, 1, 1, True, p=h, q=h))
        dx, _ = ops.conv2d(dy, None, wp, None, c, 3, 3, stride, 1, 1, True, p=h, q=h)
        t_bn = timeit(lambda: _lib.call("msml_bn_act_bwd", dx, x, coef[0], coef[1], alpha, coef[2], coef[3], None, dxb,
                                        None, pg[0], pg[1], pg[2], 0, m, c, ws, ws.numel(), _lib.BF16))
        t_fconv = timeit(lambda: ops.conv_dgrad_bnbwd(dy, wp, c, 3, 3, stride, 1, 1, h, h, x, coef, alpha))
        dxc, part = ops.conv_dgrad_bnbwd(dy, wp, c, 3, 3, stride, 1, 1, h, h, x, coef, alpha)
        if part.dtype == torch.float64:            # accumulator protocol (ops.ACC_STATS): finalize + apply in one launch
            t_app = timeit(lambda: _lib.call("msml_bn_fin_bwd_apply", dxc, x, coef[0], coef[1], alpha, coef[2], coef[3], part,
                                             None, None, 0, 0, dxb, None, pg[0], pg[1], pg[2], 0, m, c, None, None, None,
                                             None, _lib.BF16))
        else:
            t_app = timeit(lambda: _lib.call("msml_bn_act_bwd_apply", dxc, x, coef[0], coef[1], alpha, coef[2], coef[3],
                                             part, part.shape[0], None, dxb, pg[0], pg[1], pg[2], 0, m, c, cw, _lib.BF16))
        print("%4d->%4d @%3d s%d  conv %6.1f us  bn_bwd %6.1f us | fused conv %6.1f us  apply %6.1f us (rows %d) | %6.1f -> %6.1f"
              % (k, c, h, stride, t_conv * 1e6, t_bn * 1e6, t_fconv * 1e6, t_app * 1e6, part.shape[0],
                 (t_conv + t_bn) * 1e6, (t_fconv + t_app) * 1e6))


if __name__ == "__main__":
    main()
