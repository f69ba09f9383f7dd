#!/bin/bash
# Round profile on the GPU box: PMC summaries (tools/pmc_collect.sh), the steady-state kernel-trace summaries of the
# default three-stream step and of the one-stream step (isolated durations), the per-launch listing, the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
R=${1:-r06}
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
# counters FIRST, copied to profiles/ on the box: the bench line below then quotes THIS build's traffic (bench.pmc_traffic reads
# profiles/<round>_pmc_*.json); `nopmc` as second argument skips the passes (they were collected by an earlier call)
if [ "$2" != "nopmc" ]; then
  tools/pmc_collect.sh $R || exit 1
  cp gpurun_out/${R}_pmc_traffic.json gpurun_out/${R}_pmc_traffic.txt gpurun_out/${R}_pmc_step.json gpurun_out/${R}_pmc_step.txt profiles/ || exit 1
fi
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_default_run.json 2> $O/bench_default.err || exit 1
echo "default bench line written"
MSML_PROFILE_DETAIL=400 python3 bench.py --steps 12 --no-extra-modes --no-cpu-baseline > $O/detail.json 2> gpurun_out/${R}_profile_detail.txt || exit 1
echo "detail listing written"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -- python3 bench.py --steps 8 --warmup 3 --launch eager \
    --no-extra-modes --no-cpu-baseline --no-kernel-events --no-calibration > $O/eager.log 2>&1 || exit 1
python3 tools/steady_profile.py $O/eager 70 > gpurun_out/${R}_prof_eager_steady_state.txt || exit 1
MSML_BENCH_NO_SIDE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 bench.py --steps 8 --warmup 3 \
    --launch eager --no-extra-modes --no-cpu-baseline --no-kernel-events --no-calibration > $O/serial.log 2>&1 || exit 1
python3 tools/steady_profile.py $O/serial 70 > gpurun_out/${R}_prof_serial_steady_state.txt || exit 1
rm -rf $O/eager $O/serial
echo "kernel-trace summaries written"
