#!/bin/bash
# A/B of environment switches on one box: tools/ab_bench.sh "VAR=1" "VAR=2" ... (each run: 12 timed steps, eager)
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --launch eager --no-extra-modes --no-cpu-baseline --no-kernel-events --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d.get('loss'))"
done
