#!/bin/bash
# The -DMSML_LDS_GUARD build (python tools/build_variant.py --all MSML_LDS_GUARD -> variants/) over the conv tests, one
# bench-sized training step and config-5 inference -- tools/lds_guard_run.sh without its self-test (which TRAPS a kernel on
# purpose and is not something to repeat on a shared box).  Log -> profiles/r06_lds_guard.log
G=variants/libmsml_MSML_LDS_GUARD.so
echo "== guarded library (-DMSML_LDS_GUARD, round-6 sources) over tests/test_gpu_conv.py"
MSML_LIB=$PWD/$G timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
echo "== guarded library, bench-sized steps (ires50-MSML + 85 742-id head, batch 256; eager and captured)"
MSML_LIB=$PWD/$G timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-extra-modes --no-cpu-baseline --no-calibration 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ok: %.2f ms/step, loss %s' % (d['ms_per_step'], d.get('loss')))"
echo "== guarded library, config 5 inference (split-bf16)"
MSML_LIB=$PWD/$G timeout -k 10 300 python bench.py --mode infer --batch 1024 --steps 2 --warmup 1 --no-calibration 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('infer ok: %.2f ms/step' % d['ms_per_step'])"
