#!/usr/bin/env python
"""Table of the calibration samples (gpurun_out/cal_*.json and profiles/<round>_bench_box_*.json): step, probes, families."""
import glob
import json
import sys
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cal_*.json")):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    c, k = d["calibration"], d["kernels"]
    print("%-28s T %.2f  gemm %s lds %.0f reg %.0f copy %.2f | conv %.2f wg %.2f | dom %.0f" % (
        f.split("/")[-1], d["ms_per_step"], c.get("gemm_tflops"), c["mfma_lds_tflops"], c["mfma_tflops"], c["copy_tbs"],
        k["conv_igemm"]["ms_per_step"], k["conv_wgrad"]["ms_per_step"], d["roofline"]["achieved"]))
