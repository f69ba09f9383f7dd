#!/bin/bash
# Timing-only builds of k_conv_ws (16x16x32 forward): where do the 87 us of 64 -> 64 @ 56x56 go?
for v in "" READS STORE LOADS READS_WS_ABLATE_STORE_WS_ABLATE_LOADS; do
  lib=msml_amd/libmsml_hip.so; [ -n "$v" ] && lib=variants/libmsml_WS_ABLATE_$v.so
  echo "== ${v:-full}"
  MSML_LIB=$lib python tools/bench_conv.py --only fwd --shapes 0,2 --iters 20 2>&1 | grep "64->64"
done
