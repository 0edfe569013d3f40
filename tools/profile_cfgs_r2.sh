#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of the configurations DESIGN.md quotes besides configs[2]
# (one run each; program directly after `--`).  usage: tools/profile_cfgs_r2.sh <outdir under gpurun_out> [tags...]
#   c2       BASELINE configs[1]: 8192 x 8192, K=32, unmasked, normalize, 500 iterations
#   c5shape  BASELINE configs[4]'s shape on one GPU: internal 17000 x 360000 (dir-beta of 360000 x 17000), K=128, device-generated
#   general  configs[2] forced onto the 8-byte (real-valued / weighted) path: NBMF_FORCE_F64=1
#   c4shard  BASELINE configs[3]'s per-rank shard 32768 x 8192, K=64, 1-rank communicator (peer / rccl)
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
B="python3 bench.py --no-cpu-baseline"
for tag in "$@"; do
  case $tag in
    c2) A="--M 8192 --N 8192 --K 32 --no-mask --projection normalize --steps 500 --warmup 5";;
    c5shape) A="--device-data --M 17000 --N 360000 --K 128 --projection normalize --steps 5 --warmup 1";;
    general) export NBMF_FORCE_F64=1; A="--steps 10 --warmup 2";;
    c4shard_peer) A="--M 32768 --force-comm --transport peer --steps 20 --warmup 3";;
    c4shard_rccl) A="--M 32768 --force-comm --transport rccl --steps 20 --warmup 3";;
    c3) A="--steps 10 --warmup 2";;
    *) echo "unknown tag $tag"; exit 1;;
  esac
  rm -rf $O/$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/stats -- $B $A > $O/bench_$tag.log 2>&1 || { echo "FAILED $tag"; tail -5 $O/bench_$tag.log; exit 1; }
  unset NBMF_FORCE_F64
  tail -1 $O/bench_$tag.log | cut -c1-200
done
