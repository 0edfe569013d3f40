#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
timeout -k 10 300 ./build/microbench4 20000 > $O/microbench4.txt 2>&1; tail -4 $O/microbench4.txt
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash tools/ab_many.sh "nbmf_mm_amd/libnbmf_hip.so build/ab/lib_b4.so build/ab/lib_not2.so build/ab/lib_both.so" "--M 16384 --K 16 --storage f64 --steps 40 --warmup 5 --no-f64-leg" "--M 16384 --K 32 --storage f64 --steps 40 --warmup 5 --no-f64-leg" > $O/ab.txt 2>&1
cat $O/ab.txt
