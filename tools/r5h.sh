#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
NBMF_UPLOAD_TRACE=1 python tools/bench_upload_u8.py > $O/upload.txt 2>&1; grep -v "sample\|staging\|statistics" $O/upload.txt
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python tools/bench_fit_c3.py > $O/fit_c3.txt 2>&1; cat $O/fit_c3.txt
