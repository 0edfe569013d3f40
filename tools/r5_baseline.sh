#!/bin/bash
# round 5, first GPU call: the general (8-byte) path below K = 64 at HEAD of round 4 -- unprofiled lines, then the six rocprofv3 passes
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-f64-leg --no-u8-leg"
for k in 16 32; do
  $B --M 16384 --K $k --storage f64 --steps 40 --warmup 5 > $O/line_general_k$k.json 2>$O/line_general_k$k.err || exit 1
done
$B --M 8192 --N 8192 --K 32 --no-mask --projection normalize --storage f64 --steps 200 --warmup 5 > $O/line_c2_general.json 2>$O/line_c2_general.err || exit 1
NBMF_PASS_TRACE=1 $B --M 16384 --K 16 --storage f64 --steps 2 --warmup 1 > $O/trace_k16.json 2> $O/trace_k16.err || exit 1
bash tools/profile_all.sh r5a/prof general_k16 general_k32 c2_general || exit 1
echo ok
