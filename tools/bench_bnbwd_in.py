#!/usr/bin/env python
"""Microbenchmark: BatchNorm backward-apply + backward-data conv (+ next BatchNorm's sums) as two launches
(msml_bn_fin_bwd_apply -> msml_conv2d_bnbwd_acc) vs one (msml_conv2d_bnbwd_in_acc), and the forward pair
(msml_bn_fin_act_fwd -> msml_conv2d_acc vs msml_conv2d_bnin_acc), batch 256."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

SHAPES = [(256, 256, 14), (128, 128, 28), (512, 256, 14), (256, 512, 14)]


def main():
    n = 256
    for cdy, cdx, h in SHAPES:
        m = n * h * h
        dy = torch.randn(n, h, h, cdy, device="cuda").bfloat16()
        upx = torch.randn(n, h, h, cdy, device="cuda").bfloat16()
        upk = torch.rand(4, cdy, device="cuda") + 0.5
        alpha = torch.rand(cdy, device="cuda") * 0.3
        acc_in = torch.randn(8, 3, cdy, device="cuda", dtype=torch.float64)
        w = torch.randn(cdy, cdx, 3, 3, device="cuda") * 0.05
        wpt = ops.pack_weight(w, True, cdy, 0, _lib.BF16)
        bnx = torch.randn(n, h, h, cdx, device="cuda").bfloat16()
        k = torch.rand(4, cdx, device="cuda") + 0.5
        a2 = torch.rand(cdx, device="cuda") * 0.3
        pg = [torch.zeros(cdy, device="cuda") for _ in range(3)]
        dc = torch.empty_like(dy)

        def two():
            _lib.call("msml_bn_fin_bwd_apply", dy, upx, upk[0], upk[1], alpha, upk[2], upk[3], acc_in, None, None, 0, 0, dc,
                      None, pg[0], pg[1], pg[2], 1, m, cdy, None, None, None, None, _lib.BF16)
            ops.conv_dgrad_bnbwd(dc, wpt, cdx, 3, 3, 1, 1, 1, h, h, bnx, k, a2)
        t_apply = timeit(lambda: _lib.call("msml_bn_fin_bwd_apply", dy, upx, upk[0], upk[1], alpha, upk[2], upk[3], acc_in, None,
                                           None, 0, 0, dc, None, pg[0], pg[1], pg[2], 1, m, cdy, None, None, None, None, _lib.BF16))
        t_conv = timeit(lambda: ops.conv_dgrad_bnbwd(dc, wpt, cdx, 3, 3, 1, 1, 1, h, h, bnx, k, a2))
        t_two = timeit(two)
        t_one = timeit(lambda: ops.conv_dgrad_bnbwd_in(dy, upx, upk, alpha, acc_in, pg, True, wpt, cdx, bnx, k, a2))
        # forward pair
        x = torch.randn(n, h, h, cdy, device="cuda").bfloat16()
        wp = ops.pack_weight(torch.randn(cdx, cdy, 3, 3, device="cuda") * 0.05, False, cdy, 0, _lib.BF16)
        acc_f = torch.randn(8, 2, cdy, device="cuda", dtype=torch.float64).abs() * m
        bnp = (torch.ones(cdy, device="cuda"), torch.zeros(cdy, device="cuda"), torch.zeros(cdy, device="cuda"),
               torch.ones(cdy, device="cuda"), 0.1, 1e-5)
        coef = torch.empty(4, cdy, device="cuda")
        act = torch.empty_like(x)

        def two_f():
            _lib.call("msml_bn_fin_act_fwd", acc_f, float(m), bnp[0], bnp[1], bnp[2], bnp[3], 0.1, 1e-5, coef[0], coef[1],
                      coef[2], coef[3], x, alpha, None, 0, act, m, cdy, None, _lib.BF16)
            ops.conv2d(act, None, wp, None, cdx, 3, 3, 1, 1, 1, False, want_stats=True)
        t_two_f = timeit(two_f)
        t_one_f = timeit(lambda: ops.conv2d_bnin_acc(x, acc_f, bnp, alpha, wp, cdx))
        print("%4d->%4d @%3d  backward: apply %5.1f + conv %5.1f = %5.1f us (back to back %5.1f) -> one launch %5.1f | "
              "forward: two launches %5.1f -> one %5.1f"
              % (cdy, cdx, h, t_apply * 1e6, t_conv * 1e6, (t_apply + t_conv) * 1e6, t_two * 1e6, t_one * 1e6,
                 t_two_f * 1e6, t_one_f * 1e6))


if __name__ == "__main__":
    main()
