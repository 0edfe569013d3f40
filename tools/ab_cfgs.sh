#!/bin/bash
# usage: tools/ab_cfgs.sh build/libX.so "args1" "args2" ... : one bench per (lib, cfg)
ALT=$PWD/$1; shift
for cfg in "$@"; do
  for lib in $PWD/nbmf_mm_amd/libnbmf_hip.so $ALT; do
    NBMF_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline $cfg 2>/dev/null | tail -1 > gpurun_out/ab.json
    echo -n "[$cfg] $(basename $lib) "; python tools/benchline.py gpurun_out/ab.json | cut -c50-
  done
done
