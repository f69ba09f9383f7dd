#!/usr/bin/env python
"""Host issue time of the eager multi-stream training step BY PHASE (no GPU wait inside a phase: the device is drained
before each one, so a phase's wall time is what the Python / autograd threads need to enqueue it).
    python tools/host_phases.py [--frb iresnet50 --batch 256 --classes 85742]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msml_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frb", default="iresnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--classes", type=int, default=85742)
    a = ap.parse_args()
    args = argparse.Namespace(frb=a.frb, batch=a.batch, classes=a.classes, dtype="bf16", mode="train", emulate_world=1,
                              data="resident")
    tr = bench.Trainer(args, 0, 0, 1)
    ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    Fh = tr.Fh
    acc = {}

    def phase(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return out
    reps = 8
    for _ in range(reps):
        x, msk, label = tr.next_batch()
        phase("zero_grad", tr.opt.zero_grad)
        feature, final_seg, kd = phase("forward", lambda: tr.model(x))
        seg_loss = phase("seg loss", lambda: tr.seg_crit(final_seg, msk, msk))
        fn = phase("normalize", lambda: Fh.normalize(feature))
        x_grad, loss_v = phase("head forward_backward", lambda: tr.pfc.forward_backward(label, fn, tr.opt_pfc))
        phase("backward", lambda: torch.autograd.backward([fn, seg_loss], [x_grad, None]))
        phase("optimizer", lambda: (tr.opt.all_reduce_grads(1), tr.opt.step(), tr.opt_pfc.step()))
    torch.cuda.synchronize()
    tot = sum(acc.values())
    print("host issue time per step by phase (%s, batch %d): total %.2f ms" % (a.frb, a.batch, 1e3 * tot / reps))
    for k, v in acc.items():
        print("   %-24s %6.2f ms" % (k, 1e3 * v / reps))


if __name__ == "__main__":
    main()
