#!/usr/bin/env python
"""Element-wise BatchNorm kernels on COLD data: every launch works on another buffer set (rotating through > 1 GB, beyond
L2 and the 256 MB Infinity Cache), as inside the training step; torch's copy on the same rotation is the yardstick."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import ops  # noqa: E402
from msml_amd._lib import call  # noqa: E402


def rot_time(fn, nbuf, iters):
    for k in range(nbuf):
        fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % nbuf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    n = 256
    for c, h in ((256, 14), (128, 28), (64, 56), (512, 7), (64, 112)):
        m = n * h * h
        b = m * c * 2
        nbuf = max(4, int(1.5e9 / (3 * b)))
        xs = [torch.randn(m, c, device="cuda").bfloat16() for _ in range(nbuf)]
        rs = [torch.randn(m, c, device="cuda").bfloat16() for _ in range(nbuf)]
        ys = [torch.empty(m, c, device="cuda", dtype=torch.bfloat16) for _ in range(nbuf)]
        coef = torch.rand(6, c, device="cuda") + 0.5
        part = torch.zeros(32, 3, c, device="cuda")
        cw = torch.empty(98 * c, device="cuda")
        pg = torch.zeros(3, c, device="cuda")
        accs = [torch.zeros(8, 2, c, dtype=torch.float64, device="cuda") + 1.0 for _ in range(nbuf)]
        acc3 = [torch.zeros(8, 3, c, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
        g = torch.ones(c, device="cuda")
        it = 4 * nbuf
        t0 = rot_time(lambda k: ys[k].copy_(xs[k]), nbuf, it)
        t1 = rot_time(lambda k: call("msml_bn_act_fwd", xs[k], coef[0], coef[1], coef[2], None, 0, ys[k], m, c, 1), nbuf, it)
        t2 = rot_time(lambda k: call("msml_bn_act_fwd", xs[k], coef[0], coef[1], None, rs[k], 0, ys[k], m, c, 1), nbuf, it)
        t3 = rot_time(lambda k: call("msml_bn_fin_act_fwd", accs[k], float(m), g, g, None, None, 0.1, 1e-5, coef[0], coef[1],
                                     coef[2], coef[3], xs[k], None, rs[k], 0, ys[k], m, c, None, 1), nbuf, it)
        t4 = rot_time(lambda k: call("msml_bn_act_bwd_apply", rs[k], xs[k], coef[0], coef[1], coef[2], coef[3], coef[4], part,
                                     32, None, ys[k], pg[0], pg[1], pg[2], 0, m, c, cw, 1), nbuf, it)
        t5 = rot_time(lambda k: call("msml_bn_fin_bwd_apply", rs[k], xs[k], coef[0], coef[1], coef[2], coef[3], coef[4],
                                     acc3[k], None, None, 0, 0, ys[k], None, pg[0], pg[1], pg[2], 0, m, c, None, None, None,
                                     None, 1), nbuf, it)
        print("C %3d @%3d (%5.1f MB x %2d sets): copy %5.1f us %4.0f GB/s | fwd %5.1f %4.0f | fwd+res %5.1f %4.0f | fin+fwd+res %5.1f %4.0f"
              " | bwd_apply %5.1f %4.0f | fin+bwd_apply %5.1f %4.0f"
              % (c, h, b / 1e6, nbuf, t0 * 1e6, 2 * b / t0 / 1e9, t1 * 1e6, 2 * b / t1 / 1e9, t2 * 1e6, 3 * b / t2 / 1e9,
                 t3 * 1e6, 3 * b / t3 / 1e9, t4 * 1e6, 3 * b / t4 / 1e9, t5 * 1e6, 3 * b / t5 / 1e9), flush=True)
        del xs, rs, ys


if __name__ == "__main__":
    main()
