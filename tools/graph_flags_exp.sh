#!/bin/bash
# hipGraph runtime knobs of libamdhip64.so against the captured (stream fork / join kept) training step: ms per step of 12
# timed replays each.  Round 3 on MI355X / ROCm 7.2: every setting within 32.8-33.0 ms.
# Every arm keeps its own stderr (gpurun_out/gexp_<n>.err) and reports its exit status: round 3 sent stderr to /dev/null
# and the DEBUG_HIP_FORCE_GRAPH_QUEUES=8 arm died without a JSON line and without a recorded cause.
mkdir -p gpurun_out
n=0
run() {
  n=$((n + 1))
  echo "== $*"
  env "$@" python bench.py --launch graph --no-extra-modes --no-cpu-baseline --no-kernel-events --steps 12 \
      > gpurun_out/gexp_$n.json 2> gpurun_out/gexp_$n.err
  rc=$?
  if [ $rc -eq 0 ] && [ -s gpurun_out/gexp_$n.json ]; then
    python -c "import sys,json; d=json.loads(open(sys.argv[1]).read()); print(d['ms_per_step'], d['config']['launch'], d['config']['warmup_ms_per_step'])" gpurun_out/gexp_$n.json
  else
    echo "   FAILED rc=$rc; last lines of gpurun_out/gexp_$n.err:"; tail -5 gpurun_out/gexp_$n.err | sed 's/^/   /'
  fi
}
run A=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
