#!/usr/bin/env python
"""Per-stage timeline of k_conv_halo_p (the -DHALO_TRACE build, MSML_LIB=variants/libmsml_HALO_TRACE.so): one wave of
workgroup 0 stamps s_memrealtime (100 MHz) before the weights wait (1), after it (2), after the requests (3), after the
stage's MFMAs (4), after a slab-switch barrier (5), after the epilogue (6).  Prints the mean time per segment."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402


def main():
    c, h = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (128, 28)))
    n = 256
    x = torch.randn(n, h, h, c, device="cuda").bfloat16()
    w = torch.randn(c, c, 3, 3, device="cuda") * 0.03
    wp = ops.pack_weight(w, False, c, 0, _lib.BF16)
    for _ in range(3):
        ops.conv2d(x, None, wp, None, c, 3, 3, 1, 1, 1, False, want_stats=True)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(os.environ["MSML_LIB"])
    buf = (ctypes.c_ulonglong * 8192)()
    lib.msml_halo_trace_read(buf, 8192)
    wg = [(buf[4096 + 2 * i], buf[4096 + 2 * i + 1]) for i in range(256) if buf[4096 + 2 * i]]
    if wg:
        t0 = min(a for a, _ in wg)
        st = sorted((a - t0) * 0.01 for a, _ in wg)
        en = sorted((b - t0) * 0.01 for _, b in wg)
        du = sorted((b - a) * 0.01 for a, b in wg)
        print("workgroups %d: start spread %.1f us (median %.1f), end of tiles %.1f ... %.1f us, duration %.1f ... %.1f (median %.1f) us"
              % (len(wg), st[-1], st[len(st) // 2], en[0], en[-1], du[0], du[-1], du[len(du) // 2]))
    ev = [(v >> 56, v & ((1 << 56) - 1)) for v in list(buf)[:4096] if v]
    seg = {}
    for (k0, t0), (k1, t1) in zip(ev, ev[1:]):
        if t1 >= t0:
            seg.setdefault((k0, k1), []).append((t1 - t0) * 10.0)      # ns
    tot = (ev[-1][1] - ev[0][1]) * 10.0
    print("events %d, first to last %.1f us" % (len(ev), tot / 1e3))
    for k, v in sorted(seg.items()):
        print("  %d -> %d : n=%4d mean %7.1f ns  max %7.1f  sum %7.1f us" % (k[0], k[1], len(v), sum(v) / len(v), max(v), sum(v) / 1e3))


if __name__ == "__main__":
    main()
