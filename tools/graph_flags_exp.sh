run() { echo "== $*"; env "$@" MSML_GRAPH_STREAMS=1 python bench.py --launch graph --no-extra-modes --no-cpu-baseline --no-kernel-events --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['warmup_ms_per_step'])"; }
run A=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
