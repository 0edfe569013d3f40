#!/bin/bash
# A/B two builds of the library on the GPU box: parity test subset + interleaved bench rounds.
# usage: tools/ab_lib.sh build/libnbmf_X.so [bench args]
ALT=$PWD/$1; shift
NBMF_HIP_LIBRARY=$ALT timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do
  for lib in $PWD/nbmf_mm_amd/libnbmf_hip.so $ALT; do
    NBMF_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > gpurun_out/ab.json
    echo -n "$(basename $lib) "; python tools/benchline.py gpurun_out/ab.json
  done
done
