#!/usr/bin/env python
"""The dominant launches of the training step in isolation, for rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE,
SQ_* in separate runs): the 256 -> 256 @ 14x14 backward-data conv with the fused BatchNorm backward sums
(k_conv_halo<256>, bench label `conv T+bnb c256+0->256 14x14 k3x3 s1 n256`), the forward conv of the same layer (plain, and
with the BatchNorm + PReLU in front of it formed in its prologue: `conv N+bn ...`, the step's dominant launch since round 4) and
its weight gradient as the step issues it -- four layers per launch pair (msml_conv_wgrad_group, label
`wgrad u256 v256 14x14 k3x3 s1 n256 x4`) -- and, for comparison, the single-layer launch.
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_shapes.py
Distinct operand sets per launch (8 x 51 MB > L2) so that the counters see HBM / Infinity-Cache traffic, not L2 hits."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import _lib, ops  # noqa: E402

N, C, H = 256, 256, 14
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def main():
    torch.manual_seed(0)
    xs = [torch.randn(N, H, H, C, device="cuda").bfloat16() for _ in range(REPS)]
    dys = [torch.randn(N, H, H, C, device="cuda").bfloat16() for _ in range(REPS)]
    w = torch.randn(C, C, 3, 3, device="cuda") * 0.03
    wp = ops.pack_weight(w, False, C, 0, _lib.BF16)
    wpt = ops.pack_weight(w, True, C, 0, _lib.BF16)
    coef = torch.randn(4, C, device="cuda").abs() + 0.5
    alpha = torch.full((C,), 0.25, device="cuda")
    dws = [torch.zeros(C, C, 3, 3, device="cuda") for _ in range(4)]
    need = _lib.value("msml_conv_wgrad_workspace", C, C, N, H, H, 3, 3)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    arr = ctypes.c_void_p * 4
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    rmean, rvar = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    m = N * H * H
    accs = []
    for r in range(REPS):                               # the statistics of each input, as its producer would have left them
        acc = ops.stats_acc(C, xs[r].device)
        _lib.call("msml_bn_stats_acc", xs[r], m, C, acc, _lib.BF16)
        accs.append(acc)
    for r in range(REPS):
        ops.conv2d(xs[r], None, wp, None, C, 3, 3, 1, 1, 1, False, want_stats=True)                    # forward
        # forward with the BatchNorm + PReLU in front of it formed in the prologue (bench label `conv N+bn ...`)
        ops.conv2d_bnin_acc(xs[r], accs[r], (gamma, beta, rmean, rvar, 0.1, 1e-5), alpha, wp, C)
        ops.conv_dgrad_bnbwd(dys[r], wpt, C, 3, 3, 1, 1, 1, H, H, xs[r], coef, alpha)                  # backward-data + bnb
        ops.conv_wgrad(dys[r], xs[r], dws[0], C, C, C, 0, 3, 3, 1, 1, 1, accumulate=True)              # one layer
        idx = [(r + i) % REPS for i in range(4)]
        _lib.call("msml_conv_wgrad_group", arr(*[dys[i].data_ptr() for i in idx]), arr(*[xs[i].data_ptr() for i in idx]),
                  arr(*[d.data_ptr() for d in dws]), 4, C, C, C, C, C, 0, N, H, H, H, H, 3, 3, 1, 1, 1, 1, ws, ws.numel(),
                  _lib.BF16)                                                                           # four layers
    # round 5: the halo2 family -- 256 -> 256 @ 28x28 stride 2 forward (four parity planes) and its backward-data conv with
    # the fused BatchNorm sums (four output classes), 512 -> 512 @ 7x7 forward (mosaic of four images)
    x28 = [torch.randn(N, 28, 28, C, device="cuda").bfloat16() for _ in range(4)]
    c5 = 512
    x7 = [torch.randn(N, 7, 7, c5, device="cuda").bfloat16() for _ in range(REPS)]
    w5 = torch.randn(c5, c5, 3, 3, device="cuda") * 0.02
    wp5 = ops.pack_weight(w5, False, c5, 0, _lib.BF16)
    for r in range(REPS):
        ops.conv2d(x28[r % 4], None, wp, None, C, 3, 3, 2, 1, 1, False, want_stats=True)
        ops.conv_dgrad_bnbwd(dys[r], wpt, C, 3, 3, 2, 1, 1, 28, 28, x28[(r + 1) % 4], coef, alpha)
        ops.conv2d(x7[r], None, wp5, None, c5, 3, 3, 1, 1, 1, False, want_stats=True)
    torch.cuda.synchronize()
    print("launched %d x (fwd, bn+fwd, dgrad+bnb, wgrad x1, wgrad x4) at %d x %d x %d x %d + the halo2 launches" % (REPS, N, H, H, C))


if __name__ == "__main__":
    main()
