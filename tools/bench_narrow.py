#!/usr/bin/env python
"""Microbenchmark of the narrow / HBM-bound layers of the step (OSB decoder, FM bottleneck, stems, fc) at the
BASELINE batch: time and achieved GB/s of algorithmic bytes per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

BF = _lib.BF16


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rows = []
    for p in (56, 28):
        x = torch.randn(n, p, p, 32, device="cuda").bfloat16()
        dy = torch.randn(n, 2 * p, 2 * p, 32, device="cuda").bfloat16()
        dw = torch.empty(18, 18, 4, 4, device="cuda")
        t = timeit(lambda: ops.conv_wgrad(x, dy, dw, 18, 18, 18, 0, 4, 4, 2, 1, 1))
        rows.append(("deconv 4x4 s2 wgrad 18->18 @%d" % p, t, (x.numel() + dy.numel()) * 2))
    x = torch.randn(n, 56, 56, 32, device="cuda").bfloat16()
    dy = torch.randn(n, 56, 56, 32, device="cuda").bfloat16()
    dw = torch.empty(32, 32, 3, 3, device="cuda")
    t = timeit(lambda: ops.conv_wgrad(dy, x, dw, 32, 32, 32, 0, 3, 3, 1, 1, 1))
    rows.append(("conv 3x3 s1 wgrad 32->32 @56", t, (x.numel() + dy.numel()) * 2))
    for name, t, b in rows:
        print("%-40s %8.1f us  %7.0f GB/s of algorithmic bytes" % (name, t * 1e6, b / t / 1e9))


if __name__ == "__main__":
    main()


def conv1x1():
    """1x1 convs of the FM bottlenecks / stems: streaming launches (X in, Y out)."""
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for cin, cout, h in ((32, 64, 112), (32, 64, 56), (64, 32, 56), (64, 128, 28), (128, 64, 28), (128, 256, 14)):
        x = torch.randn(n, h, h, cin, device="cuda").bfloat16()
        w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
        wp = ops.pack_weight(w, False, cin, 0, BF)
        t = timeit(lambda: ops.conv2d(x, None, wp, None, cout, 1, 1, 1, 0, 0, False, want_stats=True))
        b = (x.numel() + n * h * h * cout) * 2
        print("%-40s %8.1f us  %7.0f GB/s of algorithmic bytes" % ("conv 1x1 %d->%d @%d (+stats)" % (cin, cout, h), t * 1e6, b / t / 1e9))


if __name__ == "__main__":
    conv1x1()


def gcm():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for cin, cout, h, r, s in ((64, 18, 56, 7, 1), (64, 18, 56, 1, 7), (18, 18, 56, 7, 1), (18, 18, 56, 1, 7), (64, 18, 28, 7, 1), (18, 18, 28, 1, 7)):
        cinp, coutp = ops.cpad(cin), ops.cpad(cout)
        x = torch.randn(n, h, h, cinp, device="cuda").bfloat16()
        w = torch.randn(cout, cin, r, s, device="cuda") * 0.05
        wp = ops.pack_weight(w, False, cin, 0, BF)
        bp = torch.zeros(coutp, device="cuda")
        ph, pw = (r - 1) // 2, (s - 1) // 2
        t = timeit(lambda: ops.conv2d(x, None, wp, bp, coutp, r, s, 1, ph, pw, False))
        b = (x.numel() + n * h * h * coutp) * 2
        print("%-40s %8.1f us  %7.0f GB/s of algorithmic bytes" % ("gcm conv %dx%d %d->%d @%d" % (r, s, cin, cout, h), t * 1e6, b / t / 1e9))


if __name__ == "__main__":
    gcm()


def gcm_wgrad():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for cin, h, r, s in ((64, 56, 7, 1), (64, 56, 1, 7), (18, 56, 7, 1), (18, 28, 1, 7), (64, 28, 7, 1)):
        cinp = ops.cpad(cin)
        x = torch.randn(n, h, h, cinp, device="cuda").bfloat16()
        dy = torch.randn(n, h, h, 32, device="cuda").bfloat16()
        dw = torch.empty(18, cin, r, s, device="cuda")
        ph, pw = (r - 1) // 2, (s - 1) // 2
        t = timeit(lambda: ops.conv_wgrad(dy, x, dw, 18, cin, cin, 0, r, s, 1, ph, pw))
        b = (x.numel() + dy.numel()) * 2
        print("%-40s %8.1f us  %7.0f GB/s of algorithmic bytes" % ("gcm wgrad %dx%d %d->18 @%d" % (r, s, cin, h), t * 1e6, b / t / 1e9))


if __name__ == "__main__":
    gcm_wgrad()


def deconv():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for h in (56, 28, 14):
        x0 = torch.randn(n, h, h, 32, device="cuda").bfloat16()
        x1 = torch.randn(n, h, h, 32, device="cuda").bfloat16()
        wt = torch.randn(36, 18, 4, 4, device="cuda") * 0.1
        wp = ops.pack_weight(wt, True, 18, 18, BF)
        t = timeit(lambda: ops.conv2d(x0, x1, wp, None, 32, 4, 4, 2, 1, 1, True))
        b = (2 * x0.numel() + n * 4 * h * h * 32) * 2
        print("%-40s %8.1f us  %7.0f GB/s of algorithmic bytes" % ("deconv 4x4 s2 fwd 36->18 @%d" % h, t * 1e6, b / t / 1e9))


if __name__ == "__main__":
    deconv()


def fc():
    """fc (flatten + Linear 25 088 -> 512) weight gradient, and the head's element-wise kernels."""
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    x = torch.randn(n, 7, 7, 512, device="cuda").bfloat16()
    dy = torch.randn(n, 1, 1, 512, device="cuda").bfloat16()
    dw = torch.zeros(512, 512, 7, 7, device="cuda")
    for acc in (False, True):
        t = timeit(lambda: ops.conv_wgrad(dy, x, dw, 512, 512, 512, 0, 7, 7, 1, 0, 0, accumulate=acc))
        print("%-40s %8.1f us  %7.1f TFLOP/s" % ("fc wgrad 25088->512 acc=%d" % acc, t * 1e6, 2.0 * n * 25088 * 512 / t / 1e12))
    c, e = 85742, 512
    w = torch.randn(c, e, device="cuda")
    wn = torch.empty(85760, e, device="cuda", dtype=torch.bfloat16)
    inv = torch.empty(c, device="cuda")
    t = timeit(lambda: _lib.call("msml_rownorm_fwd", w, c, 85760, e, wn, e, inv, BF))
    print("%-40s %8.1f us  %7.0f GB/s" % ("rownorm_fwd 85742x512", t * 1e6, (c * e * 6) / t / 1e9))
    g = torch.randn(c, e, device="cuda")
    dwo = torch.empty(c, e, device="cuda")
    t = timeit(lambda: _lib.call("msml_rownorm_bwd", w, inv, g, e, c, e, dwo, 0))
    print("%-40s %8.1f us  %7.0f GB/s" % ("rownorm_bwd 85742x512", t * 1e6, (c * e * 12) / t / 1e9))
    cos = torch.rand(n, 85760, device="cuda") - 0.5
    lab = torch.randint(0, c, (n,), device="cuda")
    rm, rs = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    t = timeit(lambda: _lib.call("msml_pfc_rowstats", cos, 85760, n, c, lab, 0, 64.0, 0.5, 0.0, 0.0, rm, rs))
    print("%-40s %8.1f us  %7.0f GB/s" % ("pfc_rowstats 256x85742", t * 1e6, (n * c * 4) / t / 1e9))


if __name__ == "__main__":
    fc()
