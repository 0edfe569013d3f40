#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B --steps 4 --warmup 1 > $O/bench_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B --steps 4 --warmup 1 > $O/bench_write.log 2>&1
