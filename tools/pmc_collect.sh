#!/bin/bash
# Round PMC collection on the GPU box (each pass: --pmc only, the program itself after `--`).  Results under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
R=${1:-r06}
O=gpurun_out/pmc_$R
rm -rf $O && mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/shapes/fetch -- python3 tools/pmc_shapes.py > $O/shapes_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/shapes/write -- python3 tools/pmc_shapes.py > $O/shapes_write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $O/shapes gpurun_out/${R}_pmc_traffic.json > gpurun_out/${R}_pmc_traffic.txt || exit 1
n=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$((n + 1))
  rocprofv3 --pmc $c --output-format csv -d $O/step/p$n -- python3 bench.py --steps 2 --warmup 1 --launch eager \
      --no-extra-modes --no-cpu-baseline --no-kernel-events --no-calibration > $O/step_p$n.log 2>&1 || exit 1
  echo "pass $n ($c) done"
done
python3 tools/pmc_step.py $O/step gpurun_out/${R}_pmc_step.json > gpurun_out/${R}_pmc_step.txt || exit 1
rm -rf $O/shapes $O/step          # raw CSVs (hundreds of MB) stay on the box
cat gpurun_out/${R}_pmc_traffic.txt gpurun_out/${R}_pmc_step.txt
