#!/usr/bin/env python
"""What stops an MFMA-bound kernel and HBM-bound BatchNorm launches of another stream from overlapping?  The same measurement
as bench_overlap.py with the calibration PROBES as kernel A -- their footprint is known and small: msml_probe_mfma (256 threads,
no LDS, < 64 VGPRs; one or two workgroups per CU) and msml_probe_mfma_lds (512 threads, 64 KB of static LDS) -- against BatchNorm
forward-apply launches on cold 25.7 MB tensors, and the production kernels for comparison."""
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

    seed = torch.randn(4096).to(torch.bfloat16).cuda()
    out = torch.empty(1024 * 256, dtype=torch.float32, device="cuda")
    cases = {}
    cases["probe_mfma 256 WGs x 4 waves (1 per CU), no LDS"] = lambda: call("msml_probe_mfma", seed, out, 256, 6000)
    cases["probe_mfma 512 WGs x 4 waves (2 per CU), no LDS"] = lambda: call("msml_probe_mfma", seed, out, 512, 3000)
    cases["probe_mfma 1024 WGs x 4 waves (4 per CU), no LDS"] = lambda: call("msml_probe_mfma", seed, out, 1024, 1500)
    cases["probe_mfma_lds 256 WGs x 8 waves, 64 KB LDS"] = lambda: call("msml_probe_mfma_lds", seed, out, 256, 3000)
    x = torch.randn(n, 14, 14, 256, device="cuda").bfloat16()
    dy = torch.randn(n, 14, 14, 256, device="cuda").bfloat16()
    dw = torch.zeros(256, 256, 3, 3, device="cuda")
    cases["wgrad 256@14 (k_wgrad_halo<128>, 256 VGPR, 96 KB LDS)"] = lambda: ops.conv_wgrad(dy, x, dw, 256, 256, 256, 0, 3, 3, 1, 1, 1)
    w = torch.randn(256, 256, 3, 3, device="cuda") * 0.03
    wp = ops.pack_weight(w, False, 256, 0, _lib.BF16)
    cases["conv fwd 256@14 (k_conv_halo<256>)"] = lambda: ops.conv2d(x, None, wp, None, 256, 3, 3, 1, 1, 1, False)
    x6 = torch.randn(n, 56, 56, 64, device="cuda").bfloat16()
    dy6 = torch.randn(n, 56, 56, 64, device="cuda").bfloat16()
    dw6 = torch.zeros(64, 64, 3, 3, device="cuda")
    cases["wgrad 64@56 (k_wgrad_halo<64>, 165 VGPR, 68 KB LDS)"] = lambda: ops.conv_wgrad(dy6, x6, dw6, 64, 64, 64, 0, 3, 3, 1, 1, 1)
    w6 = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    wp6 = ops.pack_weight(w6, False, 64, 0, _lib.BF16)
    cases["conv fwd 64@56 (k_conv_s2r<0>, 64 KB LDS)"] = lambda: ops.conv2d(x6, None, wp6, None, 64, 3, 3, 1, 1, 1, False)
    x5 = torch.randn(n, 7, 7, 512, device="cuda").bfloat16()
    dy5 = torch.randn(n, 7, 7, 512, device="cuda").bfloat16()
    dw5 = torch.zeros(512, 512, 3, 3, device="cuda")
    cases["wgrad 512@7 (pair strips)"] = lambda: ops.conv_wgrad(dy5, x5, dw5, 512, 512, 512, 0, 3, 3, 1, 1, 1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, fa in cases.items():
        for _ in range(3):
            fa()
        for k in range(16):
            bn(k)
        torch.cuda.synchronize()
        na, nb = 30, 240

        def run(mode):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "both":
                with torch.cuda.stream(s1):
                    for _ in range(na):
                        fa()
                with torch.cuda.stream(s2):
                    for k in range(nb):
                        bn(k)
            elif mode == "a":
                for _ in range(na):
                    fa()
            else:
                for k in range(nb):
                    bn(k)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3
        for m in ("both", "a", "b"):
            run(m)
        ta = min(run("a") for _ in range(3))
        tb = min(run("b") for _ in range(3))
        con = min(run("both") for _ in range(3))
        print("%-58s A alone %6.2f ms  BN alone %6.2f ms  concurrent %6.2f  (sum %.2f, ideal %.2f: hidden %.0f %%)"
              % (name, ta, tb, con, ta + tb, max(ta, tb), 100.0 * (ta + tb - con) / min(ta, tb)), flush=True)


if __name__ == "__main__":
    main()
