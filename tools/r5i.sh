#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
bash tools/ab_many.sh "build/ab/lib_prev.so nbmf_mm_amd/libnbmf_hip.so" "--M 8192 --N 8192 --K 32 --no-mask --projection normalize --steps 500 --warmup 5 --event-stride 1 --no-f64-leg" "--M 8192 --steps 100 --warmup 5 --no-f64-leg" "--steps 30 --warmup 3 --no-f64-leg" > $O/ab.txt 2>&1
cat $O/ab.txt
for lib in build/ab/lib_prev.so nbmf_mm_amd/libnbmf_hip.so; do echo "== $lib"; NBMF_HIP_LIBRARY=$PWD/$lib python tools/bench_c1_loop.py 20000 configs; done > $O/small.txt 2>&1; cat $O/small.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | tail -2
