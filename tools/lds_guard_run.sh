#!/bin/bash
# the -DMSML_LDS_GUARD build (tools/build_variant.py --all MSML_LDS_GUARD -> variants/) run over the
# conv tests and one bench-sized step; then the self-test: the round-3 allocation of the fused BatchNorm-backward launch
# (MSML_LDS_GUARD_DROP_RLDS=1 leaves the `rlds` term of conv_fast.hip out) must TRAP.  Log -> profiles/r05_lds_guard.log
G=variants/libmsml_MSML_LDS_GUARD.so
echo "== guarded library over tests/test_gpu_conv.py"
MSML_LIB=$G timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -3
echo "== guarded library, bench-sized steps (ires50-MSML + 85 742-id head, batch 256; eager and captured)"
MSML_LIB=$G timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-extra-modes --no-cpu-baseline --no-calibration 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ok: %.2f ms/step, loss %s' % (d['ms_per_step'], d.get('loss')))"
echo "== guarded library, config 5 inference (split-bf16) and the exact-f32 step"
MSML_LIB=$G timeout -k 10 300 python bench.py --mode infer --batch 1024 --steps 2 --warmup 1 --no-calibration 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('infer ok: %.2f ms/step' % d['ms_per_step'])"
MSML_LIB=$G timeout -k 10 300 python bench.py --dtype f32 --steps 1 --warmup 1 --no-extra-modes --no-cpu-baseline --no-calibration --launch eager 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 step ok: %.2f ms/step' % d['ms_per_step'])"
echo "== self-test: the round-3 allocation (rlds dropped) on a single-stage 64-row fused launch must trap"
MSML_LDS_GUARD_DROP_RLDS=1 MSML_LIB=$G timeout -k 10 120 python -m pytest tests/test_gpu_conv.py -q -x -k "pointwise_conv_kernel and shape1" > /tmp/lds_self.log 2>&1
rc=$?
tail -4 /tmp/lds_self.log | cut -c1-300
echo "self-test exit code: $rc (non-zero = the guard trapped)"
