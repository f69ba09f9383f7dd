#!/bin/bash
# Compile-time ablations of k_conv_halo<256> on 256 -> 256 @ 14x14 (tools/build_variant.py conv_halo.hip HALO_ABLATE_*):
# the full kernel, then with the epilogue / the loads / the LDS fragment reads removed (results are garbage: timing only).
for v in "" HALO_ABLATE_EPILOGUE HALO_ABLATE_LOADS HALO_ABLATE_READS HALO_ABLATE_READS_HALO_ABLATE_LOADS HALO_ABLATE_READS_HALO_ABLATE_LOADS_HALO_ABLATE_EPILOGUE; do
  if [ -z "$v" ]; then unset MSML_LIB; else export MSML_LIB=$PWD/variants/libmsml_$v.so; fi
  echo "== ${v:-full}"
  python tools/bench_conv.py --shapes 8 --only fwd,dgrad --iters 40 2>/dev/null | grep "256->256"
  python tools/bench_conv.py --shapes 8 --only fwd,dgrad --iters 40 2>/dev/null | grep "256->256"
done
