#!/bin/bash
# usage: tools/ab_many.sh "lib1 lib2 ..." "cfg1" "cfg2" ... : every config through every library, twice, interleaved
LIBS="$1"; shift
for cfg in "$@"; do
  for rep in 1 2; do
    for lib in $LIBS; do
      NBMF_HIP_LIBRARY=$PWD/$lib python bench.py --no-cpu-baseline --no-u8-leg $cfg 2>/dev/null | tail -1 > gpurun_out/ab.json
      echo -n "[$cfg] $(basename $lib) "; python tools/benchline.py gpurun_out/ab.json | cut -c50-
    done
  done
done
