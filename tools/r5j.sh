#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
# correctness of the two-strip variant first: the real-valued / general-path tests with small K
NBMF_HIP_LIBRARY=$PWD/build/ab/lib_two.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_reference_contract.py -x -q > $O/pytest_two.log 2>&1; tail -3 $O/pytest_two.log
bash tools/ab_many.sh "nbmf_mm_amd/libnbmf_hip.so build/ab/lib_two.so" "--M 16384 --K 16 --storage f64 --steps 40 --warmup 5 --no-f64-leg" "--M 16384 --K 8 --storage f64 --steps 40 --warmup 5 --no-f64-leg" > $O/ab.txt 2>&1
cat $O/ab.txt
