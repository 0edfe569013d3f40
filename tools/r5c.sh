#!/bin/bash
# round 5, GPU call: the pruned build (no experiment switches, no scratch) -- the whole GPU suite, then A/B against the round-4 library
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
for lib in build/ab/lib_r4.so nbmf_mm_amd/libnbmf_hip.so; do
  echo "== $lib" >> $O/score.txt
  NBMF_HIP_LIBRARY=$PWD/$lib python tools/bench_score_general.py >> $O/score.txt 2>&1 || exit 1
  echo "== $lib" >> $O/small.txt
  NBMF_HIP_LIBRARY=$PWD/$lib python tools/bench_c1_loop.py 20000 >> $O/small.txt 2>&1 || exit 1
done
cat $O/score.txt $O/small.txt
