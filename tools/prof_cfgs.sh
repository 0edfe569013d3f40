#!/bin/bash
# usage (on the GPU box): tools/prof_cfgs.sh "M N K [extra bench args]" ...   -> gpurun_out/prof_cfg/<tag>/
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_cfg
for cfg in "$@"; do
  set -- $cfg
  M=$1; N=$2; K=$3; shift 3
  tag=m${M}_n${N}_k${K}$(echo "$*" | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg/$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --M $M --N $N --K $K "$@" > gpurun_out/prof_cfg/bench_$tag.log 2>&1
  tail -1 gpurun_out/prof_cfg/bench_$tag.log | cut -c1-160
done
