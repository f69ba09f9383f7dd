#!/usr/bin/env python
"""Per-kernel and per-family PMC summary of the WHOLE training step (counters for every conv / weight
gradient kernel of the round, not only the 256 -> 256 @ 14x14 trio).

Collect (one counter group per pass, --pmc only -- no trace domains; the program itself after `--`):
    for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
      rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_step/$n -- python3 bench.py --steps 2 --warmup 1 \
          --launch eager --no-extra-modes --no-cpu-baseline --no-kernel-events
    done
Summarise:
    python3 tools/pmc_step.py gpurun_out/pmc_step profiles/r04_pmc_step.json

Units / corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are KB per dispatch; on gfx950 FETCH_SIZE
counts a wide coalesced read at half its bytes, so hbm_bytes = 2 x FETCH + WRITE.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES
/ 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
Steps in a pass = launches of k_sgd / 3 (two backbone learning-rate groups + the head)."""
import collections
import csv
import glob
import json
import re
import sys

FAMILIES = {
    "conv": re.compile(r"k_conv_(halo|fast|ws|igemm|line|pw|r32)|k_deconv4_"),
    "wgrad": re.compile(r"k_wgrad_|k_conv_wgrad|k_fc_wgrad"),
    "bn": re.compile(r"k_bn_"),
}


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    sgd = max((v[1] for k, cs in acc.items() if k.startswith("k_sgd") for v in cs.values()), default=0)
    steps = max(1, sgd // 3)
    out = {"_comment": __doc__.split("\n\n")[0].replace("\n", " "), "_steps_per_pass": steps, "kernels": {}, "families": {}}
    fam = {k: collections.defaultdict(float) for k in FAMILIES}
    for name, cs in sorted(acc.items()):
        n = max(v[1] for v in cs.values())
        row = {"launches_per_step": round(n / steps, 2)}
        avg = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            row["fetch_size_kb"] = round(avg["FETCH_SIZE"], 1)
            row["write_size_kb"] = round(avg["WRITE_SIZE"], 1)
            row["hbm_bytes_per_launch"] = int((2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("GRBM_GUI_ACTIVE"):
            row["mfma_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (avg["GRBM_GUI_ACTIVE"] / 8.0), 4)
        if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_share"] = round(avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"], 4)
        out["kernels"][name] = row
        for fk, pat in FAMILIES.items():
            if pat.search(name):
                fam[fk]["launches"] += n / steps
                if "hbm_bytes_per_launch" in row:
                    fam[fk]["hbm_bytes_per_step"] += row["hbm_bytes_per_launch"] * n / steps
    for fk, v in fam.items():
        if v["launches"]:
            out["families"][fk] = {"launches_per_step": round(v["launches"], 1),
                                   "hbm_bytes_per_step": int(v["hbm_bytes_per_step"]),
                                   "hbm_bytes_per_launch": int(v["hbm_bytes_per_step"] / v["launches"])}
    json.dump(out, open(dst, "w"), indent=1)
    print("steps per pass %d; %d kernels; families %s" % (steps, len(out["kernels"]), json.dumps(out["families"])))


if __name__ == "__main__":
    main()
