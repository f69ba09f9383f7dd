#!/usr/bin/env python
"""Where the Python side of the eager training step spends its host time: cProfile over the forward (issuing thread) and over
every autograd Function's backward (the autograd engine's device thread -- `sys.setprofile` does not reach it, so each
`backward` staticmethod is wrapped with enable / disable of ONE profile object).  The profiler roughly doubles the cost of a
Python call, so read the listing as a ranking, not as milliseconds.
    python tools/host_cprofile.py [--frb iresnet50 --batch 256 --classes 85742 --top 45]"""
import argparse
import cProfile
import io
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msml_amd import blocks, functional, ops  # noqa: E402


def wrap_backwards(prof):
    for mod in (blocks, functional):
        for name in dir(mod):
            cls = getattr(mod, name)
            if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function \
                    and "backward" in cls.__dict__:
                inner = cls.__dict__["backward"].__func__

                def make(inner):
                    def backward(ctx, *grads):
                        prof.enable()
                        try:
                            return inner(ctx, *grads)
                        finally:
                            prof.disable()
                    return backward
                cls.backward = staticmethod(make(inner))


def report(prof, title, top):
    s = io.StringIO()
    st = pstats.Stats(prof, stream=s)
    st.sort_stats("tottime").print_stats(top)
    print("==== %s, by own time" % title)
    print(s.getvalue())
    s = io.StringIO()
    st = pstats.Stats(prof, stream=s)
    st.sort_stats("cumtime").print_stats(top)
    print("==== %s, by cumulative time" % title)
    print(s.getvalue())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frb", default="iresnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--classes", type=int, default=85742)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    args = argparse.Namespace(frb=a.frb, batch=a.batch, classes=a.classes, dtype="bf16", mode="train", emulate_world=1,
                              data="resident")
    tr = bench.Trainer(args, 0, 0, 1)
    ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    pf, pb = cProfile.Profile(), cProfile.Profile()
    wrap_backwards(pb)
    Fh = tr.Fh
    reps = 6
    for _ in range(reps):
        x, msk, label = tr.next_batch()
        tr.opt.zero_grad()
        torch.cuda.synchronize()
        pf.enable()
        feature, final_seg, kd = tr.model(x)
        pf.disable()
        seg_loss = tr.seg_crit(final_seg, msk, msk)
        fn = Fh.normalize(feature)
        x_grad, loss_v = tr.pfc.forward_backward(label, fn, tr.opt_pfc)
        torch.cuda.synchronize()
        torch.autograd.backward([fn, seg_loss], [x_grad, None])
        tr.opt.all_reduce_grads(1)
        tr.opt.step()
        tr.opt_pfc.step()
    torch.cuda.synchronize()
    print("%d steps of %s, batch %d" % (reps, a.frb, a.batch))
    report(pf, "forward", a.top)
    report(pb, "backward (inside the Functions' backward)", a.top)


if __name__ == "__main__":
    main()
