#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
bash tools/ab_many.sh "nbmf_mm_amd/libnbmf_hip.so build/ab/lib_wgs4.so build/ab/lib_pf2.so build/ab/lib_noprio.so" "--M 16384 --K 16 --storage f64 --steps 40 --warmup 5 --no-f64-leg" "--M 16384 --K 32 --storage f64 --steps 40 --warmup 5 --no-f64-leg" "--storage f64 --steps 20 --warmup 3 --no-f64-leg" > $O/ab.txt 2>&1
cat $O/ab.txt
NBMF_HIP_LIBRARY=$PWD/build/ab/lib_pf2.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/pytest_pf2.log 2>&1; tail -3 $O/pytest_pf2.log
NBMF_UPLOAD_TRACE=1 python tools/bench_upload_u8.py > $O/upload.txt 2>&1; cat $O/upload.txt
