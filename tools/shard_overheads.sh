#!/bin/bash
# One rank of the strong-scaling split of configs[2] over 8 GPUs (8192 of the 65536 rows), with each transport attached
# as a 1-rank communicator: what an iteration costs on the shard apart from the wires.  Ideal: 8 x the 1-GPU rate.
out=${1:-gpurun_out/shard_overheads.log}
: > $out
for t in none peer peer2 rccl rccl2; do
  if [ $t = none ]; then extra=""; else extra="--force-comm --transport $t"; fi
  echo "== $t" >> $out
  timeout -k 10 120 python bench.py --M 8192 --steps 200 --warmup 20 --no-cpu-baseline $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('it/s %.1f  ms/it %.4f  hpass %.4f wpass %.4f' % (d['value'], d['ms_per_step'], r['hpass_ms'], r['wpass_ms']))" >> $out || exit 1
done
cat $out
