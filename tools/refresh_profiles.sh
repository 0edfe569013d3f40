#!/bin/bash
# Turn one run of tools/profile_all.sh under gpurun_out/<dir> into the summaries committed under profiles/:
#   <round>_<tag>.txt         kernel-trace stats + every PMC pass + the derived block (tools/prof_summary.py)
#   <round>_<tag>.json        machine-readable per-launch means of the H-sweep (bench.py reads the c3 one for roofline.traffic)
#   <round>_<tag>_bench.json  the bench line of the profiled --stats run
# usage: tools/refresh_profiles.sh <dir under gpurun_out, e.g. r3i/prof> <round tag, e.g. r3>
# Every summary's first lines name the commit of the sources the profiled library was built from: the last commit that
# touches the product or the bench (`git log -1 --format=%h -- nbmf_mm_amd include bench.py` -- run it at HEAD to compare;
# later commits of the round add only documents and these files).  Refuses to run on a dirty product tree.
O=gpurun_out/$1; R=$2
if ! git diff --quiet HEAD -- nbmf_mm_amd include bench.py; then echo "uncommitted changes under nbmf_mm_amd/ include/ bench.py: commit first"; exit 1; fi
SRC=$(git log -1 --format=%h -- nbmf_mm_amd include bench.py)
declare -A NAME=([c3]=c3_k64_masked)
for d in $O/*/; do
  t=$(basename $d); n=${NAME[$t]:-$t}
  { echo "# sources: commit $SRC (git log -1 --format=%h -- nbmf_mm_amd include bench.py), library built from them by make"; python tools/prof_summary.py $O/$t profiles/${R}_$n.json; } > profiles/${R}_$n.txt
  grep '^{' $O/$t/bench_stats.log | tail -1 > profiles/${R}_${n}_bench.json
  python - $R $n <<'P'
import json,sys
R,t=sys.argv[1:3]
d=json.loads(open(f"profiles/{R}_{t}_bench.json").read().strip().splitlines()[-1]); r=d['roofline']
print("%-16s it/s %8.1f norm %8.1f hpass %.3f wpass %.3f frac %.3f exec %.3f wexec %.3f" % (t, d['value'], d['normalize_value'], r['hpass_ms'], r['wpass_ms'], r['frac'], r['executed_frac'], r['wpass_executed_frac']))
P
done
