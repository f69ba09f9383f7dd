#!/bin/bash
# where does the eager step turn host-bound?  One GPU, the process pinned to K CPUs (what one rank of
# an 8-rank job gets on a 16-CPU box share is K = 2), eager multi-stream issue vs hipGraph replay, with and without the
# one-call-per-block C entry (MSML_BLOCK_C_ENTRY=1).  Prints K, mode, ms/step (mean, median).
for k in 16 4 2 1; do
  for cfg in "eager MSML_X=1" "eager MSML_BLOCK_C_ENTRY=1" "graph MSML_X=1"; do
    set -- $cfg
    r=$(env $2 python bench.py --cpu-share $k --launch $1 --no-extra-modes --no-cpu-baseline --no-kernel-events --no-calibration --steps 12 2>/dev/null |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d['config']['cpu_share'])")
    echo "K=$k launch=$1 $2: $r"
  done
done
