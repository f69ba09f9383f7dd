#!/usr/bin/env python
"""profiles/<round>_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_shapes.py:
HBM bytes per LAUNCH of the dominant launches of the step, keyed by the labels bench.py gives them.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_shapes/fetch -- python3 tools/pmc_shapes.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_shapes/write -- python3 tools/pmc_shapes.py
    python3 tools/pmc_traffic.py gpurun_out/pmc_shapes profiles/r04_pmc_traffic.json

pmc_shapes.py issues, per repetition: forward conv, backward-data conv + BatchNorm sums, weight gradient of ONE layer
(k_wgrad_halo + k_wgrad_reduce_rows), weight gradient of FOUR layers in one launch pair -- the two weight-gradient
launches share kernel names and are told apart by dispatch order.  hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950
counts wide reads at half their bytes, MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import json
import re
import sys

N, C, H = 256, 256, 14
ACT = N * H * H * C * 2                 # one bf16 activation tensor
WB = C * C * 9 * 2                      # packed bf16 weights
DW = C * C * 9 * 4                      # f32 gradient


def rows(d):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])))
    return sorted(out)


def per_kind(d, counter):
    """{kind: [value per dispatch]} in dispatch order."""
    seq = {}
    for _, name, cn, val in rows(d):
        if cn != counter:
            continue
        kind = None
        m = re.search(r"k_conv_halo<\d+, \d+, (true|false), (true|false)", name)       # third / fourth template argument = FUSE / XF
        if m:
            kind = "dgrad" if m.group(1) == "true" else ("fwd_bn" if m.group(2) == "true" else "fwd")
        m2 = re.search(r"k_conv_halo2<\d+, \d+, (\d), (\d), (true|false)", name)   # GEOM, MOS, FUSE
        if m2:
            kind = {"1": "h2_s2_fwd", "2": "h2_s2_dgrad", "0": "h2_mosaic"}[m2.group(1)]
        elif "k_wgrad_halo" in name:
            kind = "wgrad"
        elif "k_wgrad_reduce" in name:
            kind = "reduce"
        if kind:
            seq.setdefault(kind, []).append(val)
    return seq


def main():
    src, dst = sys.argv[1], sys.argv[2]
    f, w = per_kind(src + "/fetch", "FETCH_SIZE"), per_kind(src + "/write", "WRITE_SIZE")
    mean = lambda v: sum(v) / max(len(v), 1)      # noqa: E731
    out = {"_comment": __doc__.split("\n\n")[0].replace("\n", " ") + "  Collected by the two commands in tools/pmc_traffic.py; "
           "distinct 51 MB operand sets per repetition so that the counters see HBM, not L2 hits."}

    def entry(label, kernel, fk, wk, alg, note):
        out[label] = {"kernel": kernel, "fetch_size_kb": round(fk), "write_size_kb": round(wk),
                      "hbm_bytes": int((2 * fk + wk) * 1024), "algorithmic_bytes": alg, "note": note}
    halo = "[k_conv_halo<14x14 px x 256 ch, 8 waves>]"
    entry("conv T+bnb c256+0->256 14x14 k3x3 s1 n256 " + halo, "k_conv_halo<256, 1, FUSE, 16x16x32 MFMA>",
          mean(f["dgrad"]), mean(w["dgrad"]), 3 * ACT + WB,
          "backward-data conv with the fused BatchNorm backward sums: dY + saved BatchNorm input read, dX written")
    entry("conv N c256+0->256 14x14 k3x3 s1 n256 " + halo, "k_conv_halo<256, 1, forward, 16x16x32 MFMA>",
          mean(f["fwd"]), mean(w["fwd"]), 2 * ACT + WB, "forward conv + statistics: input once, output once")
    if f.get("fwd_bn"):
        entry("conv N+bn c256+0->256 14x14 k3x3 s1 n256 " + halo, "k_conv_halo<256, 1, forward, BatchNorm in the prologue, 16x16x32 MFMA>",
              mean(f["fwd_bn"]), mean(w["fwd_bn"]), 3 * ACT + WB,
              "BatchNorm + PReLU + conv in one launch: raw input read, normalised input written through, output written")
    # weight gradients: dispatches alternate one layer / four layers
    for which, label, layers in ((0, "wgrad u256 v256 14x14 k3x3 s1 n256", 1), (1, "wgrad u256 v256 14x14 k3x3 s1 n256 x4", 4)):
        fk = mean(f["wgrad"][which::2]) + mean(f["reduce"][which::2])
        wk = mean(w["wgrad"][which::2]) + mean(w["reduce"][which::2])
        entry(label, "k_wgrad_halo<128> + k_wgrad_reduce_rows (%d layer%s per launch pair)" % (layers, "s" * (layers > 1)),
              fk, wk, layers * (2 * ACT + DW), "operands + split-K slabs written and re-read + the f32 gradient")
    if f.get("h2_s2_fwd"):
        a28 = N * 28 * 28 * C * 2
        entry("conv N c256+0->256 28x28 k3x3 s2 n256 [k_conv_halo2<stride-2 forward, 4 parity planes, 14x14 px>]",
              "k_conv_halo2<256, 1, stride-2 forward>", mean(f["h2_s2_fwd"]), mean(w["h2_s2_fwd"]), a28 + ACT + WB,
              "stride-2 forward + statistics: the 28x28 input once (as four parity planes), the 14x14 output once")
        entry("conv T+bnb c256+0->256 14x14 k3x3 s2 n256 [k_conv_halo2<stride-2 backward-data, 4 output classes, 14x14 px>]",
              "k_conv_halo2<256, 1, stride-2 backward-data, FUSE>", mean(f["h2_s2_dgrad"]), mean(w["h2_s2_dgrad"]),
              ACT + 2 * a28 + WB, "stride-2 backward-data + BatchNorm sums: dY (re-read by the four class slices), saved "
              "28x28 input read, 28x28 dX written")
        a7 = N * 7 * 7 * 512 * 2
        entry("conv N c512+0->512 7x7 k3x3 s1 n256 [k_conv_halo2<mosaic of four 7x7 images x 128 ch>]",
              "k_conv_halo2<128, 2, mosaic>", mean(f["h2_mosaic"]), mean(w["h2_mosaic"]), 2 * a7 + 512 * 512 * 9 * 2,
              "7x7 forward + statistics: input once per 128-channel block of the output (x 4), output once, weights per tile")
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print("%-100s %6.1f MB = %.2f x algorithmic" % (k[:100], v["hbm_bytes"] / 1e6, v["hbm_bytes"] / v["algorithmic_bytes"]))


if __name__ == "__main__":
    main()
