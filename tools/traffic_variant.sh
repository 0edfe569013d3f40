#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and plain bench line of the headline workload for one build of the library.
# usage: tools/traffic_variant.sh <library.so relative to the repo, or "-"> <outdir under gpurun_out>
[ "$1" != "-" ] && export NBMF_HIP_LIBRARY=$PWD/$1
python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/$2.bench.log 2>&1 &&
tools/pmc_traffic.sh $2 &&
python3 tools/prof_summary.py gpurun_out/$2 > gpurun_out/$2.summary.txt &&
grep -o '"value": [0-9.]*\|"hpass_ms": [0-9.]*\|"wpass_ms": [0-9.]*' gpurun_out/$2.bench.log | tr '\n' ' ' && echo &&
grep "pass_kernel" gpurun_out/$2.summary.txt
