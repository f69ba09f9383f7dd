#!/usr/bin/env python
"""Do an MFMA-bound kernel on one stream and HBM-bound BatchNorm kernels on another really overlap?  Serial time (both
on one stream) vs concurrent time (two streams) for: the 256->256@14x14 weight gradient (k_wgrad_halo<128>, 256 VGPRs x 2
waves per SIMD = the whole register file), the 64->64@56x56 one (k_wgrad_halo<64>, 165 VGPRs), the forward conv
(k_conv_halo<256>, ~200 VGPRs), each against BatchNorm forward-apply launches on cold 25.7 MB tensors."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402
from msml_amd._lib import call  # noqa: E402


def main():
    n = 256
    torch.manual_seed(0)
    bn_x = [torch.randn(n * 196, 256, device="cuda").bfloat16() for _ in range(16)]
    bn_y = [torch.empty_like(t) for t in bn_x]
    coef = torch.rand(4, 256, device="cuda") + 0.5

    def bn(k):
        call("msml_bn_act_fwd", bn_x[k % 16], coef[0], coef[1], coef[2], None, 0, bn_y[k % 16], n * 196, 256, 1)

    cases = {}
    x = torch.randn(n, 14, 14, 256, device="cuda").bfloat16()
    dy = torch.randn(n, 14, 14, 256, device="cuda").bfloat16()
    dw = torch.zeros(256, 256, 3, 3, device="cuda")
    cases["wgrad 256@14 (k_wgrad_halo<128>, 256 VGPR)"] = lambda: ops.conv_wgrad(dy, x, dw, 256, 256, 256, 0, 3, 3, 1, 1, 1)
    x6 = torch.randn(n, 56, 56, 64, device="cuda").bfloat16()
    dy6 = torch.randn(n, 56, 56, 64, device="cuda").bfloat16()
    dw6 = torch.zeros(64, 64, 3, 3, device="cuda")
    cases["wgrad 64@56 (k_wgrad_halo<64>, 165 VGPR)"] = lambda: ops.conv_wgrad(dy6, x6, dw6, 64, 64, 64, 0, 3, 3, 1, 1, 1)
    w = torch.randn(256, 256, 3, 3, device="cuda") * 0.03
    wp = ops.pack_weight(w, False, 256, 0, _lib.BF16)
    cases["conv fwd 256@14 (k_conv_halo<256>, 198 VGPR)"] = lambda: ops.conv2d(x, None, wp, None, 256, 3, 3, 1, 1, 1, False)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, fa in cases.items():
        for _ in range(3):
            fa()
        for k in range(16):
            bn(k)
        torch.cuda.synchronize()
        na, nb = 30, 120

        def run(concurrent):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if concurrent:
                with torch.cuda.stream(s1):
                    for _ in range(na):
                        fa()
                with torch.cuda.stream(s2):
                    for k in range(nb):
                        bn(k)
            else:
                for _ in range(na):
                    fa()
                for k in range(nb):
                    bn(k)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3
        run(True), run(False)
        ser = min(run(False) for _ in range(3))
        con = min(run(True) for _ in range(3))
        with torch.cuda.stream(s1):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(na):
            fa()
        torch.cuda.synchronize()
        ta = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for k in range(nb):
            bn(k)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) * 1e3
        print("%-52s A alone %6.2f ms  BN alone %6.2f ms  serial %6.2f  concurrent %6.2f  (ideal overlap %.2f)"
              % (name, ta, tb, ser, con, max(ta, tb)), flush=True)


if __name__ == "__main__":
    main()
