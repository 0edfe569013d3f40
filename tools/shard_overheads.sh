#!/bin/bash
# One rank of the strong-scaling split of configs[2] over N GPUs (65536 / N of its rows), with each transport attached as a
# 1-rank communicator: what an iteration costs on the shard apart from the wires.  Ideal: N x the 1-GPU rate.
# usage: tools/shard_overheads.sh [out file] [rows per rank ...]     (default: 32768 16384 8192 = 2, 4, 8 ranks)
out=${1:-gpurun_out/shard_overheads.log}; shift
rows=${@:-32768 16384 8192}
: > $out
for m in $rows; do
for t in none peer peer2 rccl rccl2; do
  if [ $t = none ]; then extra=""; else extra="--force-comm --transport $t"; fi
  echo -n "rows $m, $((65536 / m)) ranks, transport $t: " >> $out
  timeout -k 10 120 python bench.py --M $m --steps 200 --warmup 20 --no-cpu-baseline --no-f64-leg --no-u8-leg $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('it/s %.1f  ms/it %.4f  hpass %.4f wpass %.4f  outside the sweeps %.4f' % (d['value'], d['ms_per_step'], r['hpass_ms'], r['wpass_ms'], d['ms_per_step'] - r['hpass_ms'] - r['wpass_ms']))" >> $out || exit 1
done; done
cat $out
