#!/bin/bash
# Tests of the measured-slower kernel variants that are NOT compiled into the shipped library (DESIGN section 8:
# R15 half-stage weight ring, XB BatchNorm backward in the backward-data prologue, BatchNorm in k_conv_ws's prologue,
# stride-2 strips of k_wgrad_halo, the stem's BatchNorm backward sums from the first block's apply kernel, the halo tile's
# weights in registers).  Build the experiment library HERE first (hipcc cross-compiles without a GPU):
#     python tools/build_variant.py --all MSML_EXPERIMENTS
# then on the GPU box:
#     bash tools/experiments_run.sh > gpurun_out/experiments_variant.log 2>&1
set -e -o pipefail
LIB=variants/libmsml_MSML_EXPERIMENTS.so
test -f $LIB || { echo "missing $LIB: python tools/build_variant.py --all MSML_EXPERIMENTS"; exit 2; }
export MSML_LIB=$PWD/$LIB
echo "== variant tests (msml_has_experiments = 1)"
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -q -m gpu -k "bn_in_lds or from_accumulator or backward_in_the_prologue or stride2_on_the_strip or sums_of_a_stem or stem_backward_sums or bit_neutral"
echo "== halo conv tests with the half-stage weight ring (MSML_HALO_R15=1)"
MSML_HALO_R15=1 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "halo and not halo2"
echo "== halo conv tests with the weights in registers (MSML_HALO_BREG=1)"
MSML_HALO_BREG=1 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "halo and not halo2"
