#!/bin/bash
# one calibration sample: a short bench line (calibration probes + 12 timed steps + the one-stream event pass) -> gpurun_out/cal_<tag>.json
python bench.py --steps 12 --warmup 3 --no-extra-modes --no-cpu-baseline 2>/dev/null > gpurun_out/cal_$1.json
