#!/bin/bash
# usage: tools/ab3.sh <rounds> "<bench args>" lib1.so lib2.so ... : interleaved rounds of one bench configuration over several builds
R=$1; CFG=$2; shift 2
for r in $(seq $R); do
  for lib in "$@"; do
    NBMF_HIP_LIBRARY=$PWD/$lib python bench.py --no-cpu-baseline --no-f64-leg --no-u8-leg $CFG 2>/dev/null | tail -1 > gpurun_out/ab.json
    echo -n "[$CFG] $(basename $lib) "; python tools/benchline.py gpurun_out/ab.json | cut -c50-
  done
done
