#!/bin/bash
# Config 2 (ires18-MSML + 10 000-id PartialFC, batch 128) under the kernel tracer: which queue idles, and for whom
# (VERDICT r5 item 2).  Eager issue with the side streams and the hipGraph replay, tools/stream_gaps.py on each.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/cfg2_trace
rm -rf $O && mkdir -p $O
ARGS="--frb iresnet18 --classes 10000 --batch 128 --steps 8 --warmup 3 --no-extra-modes --no-cpu-baseline --no-kernel-events --no-calibration"
for mode in eager graph; do
  rocprofv3 --kernel-trace --output-format csv -d $O/$mode -- python3 bench.py $ARGS --launch $mode > $O/$mode.json 2> $O/$mode.err || exit 1
  echo "== config 2, --launch $mode: $(python3 -c "import json;d=json.load(open('$O/$mode.json'));print(d['ms_per_step'],'ms per step under the tracer')")"
  python3 tools/stream_gaps.py $O/$mode || exit 1
  python3 tools/steady_profile.py $O/$mode 12 || exit 1
done
rm -rf $O/eager $O/graph
