#!/bin/bash
# Unprofiled bench.py lines of every configuration DESIGN.md quotes, one JSON line per configuration (the figures in
# DESIGN.md 5 are these; the rocprofv3 runs behind profiles/rN_<cfg>.txt are 1-10 % slower).  usage: tools/bench_lines.sh <outfile>
export TMPDIR=/tmp
OUT=$1; : > $OUT
B="python3 bench.py --no-cpu-baseline --no-u8-leg --no-f64-leg"
run() { echo "# $1: bench.py $2" >> $OUT; $B $2 2>/dev/null | tail -1 >> $OUT; }
run c2 "--M 8192 --N 8192 --K 32 --no-mask --projection normalize --steps 500 --warmup 5 --event-stride 1"
run c2_stride4 "--M 8192 --N 8192 --K 32 --no-mask --projection normalize --steps 500 --warmup 5"
run general "--storage f64 --steps 50 --warmup 5"
run generalw "--storage f64w --steps 50 --warmup 5"
run general_k16 "--M 16384 --K 16 --storage f64 --steps 100 --warmup 5"
run general_k32 "--M 16384 --K 32 --storage f64 --steps 100 --warmup 5"
run general_k128 "--M 16384 --K 128 --storage f64 --steps 50 --warmup 5"
run binary_k128_16384 "--M 16384 --K 128 --steps 50 --warmup 5"
run c2_general "--M 8192 --N 8192 --K 32 --no-mask --projection normalize --storage f64 --steps 300 --warmup 5 --event-stride 1"
run shard8192 "--M 8192 --steps 200 --warmup 5"
run c4shard_peer "--M 32768 --force-comm --transport peer --steps 50 --warmup 5"
run c4shard_rccl "--M 32768 --force-comm --transport rccl --steps 50 --warmup 5"
run c5shape "--device-data --M 17000 --N 360000 --K 128 --projection normalize --steps 6 --warmup 2"
