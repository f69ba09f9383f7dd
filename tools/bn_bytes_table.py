#!/usr/bin/env python
"""BatchNorm family: ALGORITHMIC bytes per step against the HBM bytes the counters saw ("a per-launch-class
table algorithmic bytes vs counter bytes for the family").

    python tools/bn_bytes_table.py profiles/r06_bench_default_run.json profiles/r06_pmc_step.json

Algorithmic: the byte counts bench.py attaches to every instrumented BatchNorm launch (msml_amd/functional.py, blocks.py:
streams x elements x element size -- forward apply 2 streams, + residual 3; backward apply 3-5; reduce + apply 5-8; statistics
1), summed per label over the one-stream event pass of the bench line.  Counters: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
over `bench.py --steps 2` (tools/pmc_collect.sh), hbm = 2 x FETCH_SIZE + WRITE_SIZE KB per dispatch (the gfx950 correction of
MI355X_MICROARCH.md), summed per kernel name (tools/pmc_step.py).  Labels and kernel names do not map one to one (the apply
kernel of `bn_act_bwd` is the same k_bn_fin_bwd_apply the `bn_act_bwd_apply` label launches), so the classes are:
  forward   = labels bn_act_fwd                      <-> kernels k_bn_fin_act_fwd, k_bn_act_fwd
  backward  = labels bn_act_bwd_apply + bn_act_bwd   <-> kernels k_bn_fin_bwd_apply, k_bn_bwd_reduce, k_bn_bwd_apply, k_bn_act_bwd*
  statistics = label bn_stats                        <-> kernels k_bn_stats"""
import json
import re
import sys


def main():
    line = json.load(open(sys.argv[1]))
    pmc = json.load(open(sys.argv[2]))
    alg = {"forward": [0.0, 0, 0.0], "backward": [0.0, 0, 0.0], "statistics": [0.0, 0, 0.0]}       # bytes, launches, ms
    for name, v in line["kernels"].items():
        base = name.split(" ")[0]
        cls = {"bn_act_fwd": "forward", "bn_act_bwd_apply": "backward", "bn_act_bwd": "backward", "bn_stats": "statistics"}.get(base)
        if cls is None or not v.get("gbps"):
            continue
        alg[cls][0] += v["gbps"] * 1e9 * v["ms_per_step"] * 1e-3
        alg[cls][1] += v["launches_per_step"]
        alg[cls][2] += v["ms_per_step"]
    pat = {"forward": re.compile(r"k_bn_(fin_)?act_fwd"), "statistics": re.compile(r"k_bn_stats"),
           "backward": re.compile(r"k_bn_(fin_bwd_apply|bwd_reduce|bwd_apply|act_bwd)")}
    cnt = {k: [0.0, 0.0] for k in alg}
    for name, row in pmc["kernels"].items():
        for cls, rx in pat.items():
            if rx.search(name) and "hbm_bytes_per_launch" in row:
                cnt[cls][0] += row["hbm_bytes_per_launch"] * row["launches_per_step"]
                cnt[cls][1] += row["launches_per_step"]
    print("| class | launches / step (events / counters) | algorithmic GB / step | counter GB / step | counter / algorithmic | ms / step (one stream) | algorithmic TB/s |")
    print("|---|---|---|---|---|---|---|")
    ta = tc = tm = 0.0
    for cls in ("forward", "backward", "statistics"):
        a, n, ms = alg[cls]
        c, nc = cnt[cls]
        ta, tc, tm = ta + a, tc + c, tm + ms
        print("| %s | %d / %d | %.2f | %.2f | %.2f | %.3f | %.2f |" % (cls, n, nc, a / 1e9, c / 1e9, c / max(a, 1), ms, a / max(ms, 1e-9) / 1e9))
    print("| total | | %.2f | %.2f | %.2f | %.3f | %.2f |" % (ta / 1e9, tc / 1e9, tc / max(ta, 1), tm, ta / max(tm, 1e-9) / 1e9))


if __name__ == "__main__":
    main()
