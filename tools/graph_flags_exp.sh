#!/bin/bash
# hipGraph runtime knobs of libamdhip64.so against the captured (stream fork / join kept) training step: ms per step of 12
# timed replays each.  Round 3 on MI355X / ROCm 7.2: every setting within 32.8-33.0 ms.
run() { echo "== $*"; env "$@" python bench.py --launch graph --no-extra-modes --no-cpu-baseline --no-kernel-events --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['warmup_ms_per_step'])"; }
run A=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
