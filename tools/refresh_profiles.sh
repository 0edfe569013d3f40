#!/bin/bash
# Turn one run of tools/profile_c3.sh + tools/profile_cfgs_r2.sh (+ default bench line, small shapes, shard overheads)
# under gpurun_out/<dir> into the summaries committed under profiles/.  usage: tools/refresh_profiles.sh <dir> <round tag, e.g. r2>
O=gpurun_out/$1; R=$2
python tools/prof_summary.py $O/c3 profiles/${R}_c3_k64_masked.json > profiles/${R}_c3_k64_masked.txt
grep '^{' $O/c3/bench_stats.log | tail -1 > profiles/${R}_c3_k64_masked_bench.json
for t in c2 c5shape general c4shard_peer c4shard_rccl; do
  [ -d $O/cfg/$t ] || continue
  python tools/prof_summary.py $O/cfg/$t > profiles/${R}_$t.txt
  grep '^{' $O/cfg/bench_$t.log | tail -1 > profiles/${R}_${t}_bench.json
done
[ -f $O/small.txt ] && cp $O/small.txt profiles/${R}_small_problems_single_launch_vs_five_kernels.txt
[ -f $O/bench_default.json ] && cp $O/bench_default.json profiles/${R}_bench_default_line.json
[ -f $O/shard.txt ] && cp $O/shard.txt profiles/${R}_shard8192_one_rank_transports.txt
for t in c3_k64_masked c2 c4shard_peer c4shard_rccl c5shape general; do python - $R $t <<'P'
import json,sys
R,t=sys.argv[1:3]
d=json.loads(open(f"profiles/{R}_{t}_bench.json").read().strip().splitlines()[-1]); r=d['roofline']
print("%-14s it/s %8.1f norm %8.1f hpass %.3f wpass %.3f frac %.3f exec %.3f wexec %.3f" % (t, d['value'], d['normalize_value'], r['hpass_ms'], r['wpass_ms'], r['frac'], r['executed_frac'], r['wpass_executed_frac']))
P
done
