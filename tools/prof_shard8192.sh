#!/bin/bash
# Per-kernel times on the 8192-row strong-scaling shard (configs[2] over 8 GPUs), alone and with the peer transport
# attached as a 1-rank communicator.  usage: tools/prof_shard8192.sh <outdir under gpurun_out>
export TMPDIR=/tmp
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --M 8192 --steps 100 --warmup 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/none/stats -- $B > $O/none.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/peer/stats -- $B --force-comm --transport peer > $O/peer.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/peer2/stats -- $B --force-comm --transport peer2 > $O/peer2.log 2>&1 &&
for v in none peer peer2; do python3 tools/prof_summary.py $O/$v | grep -v "^$" | head -16; done
