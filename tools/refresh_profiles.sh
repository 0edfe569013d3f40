#!/bin/bash
# Turn one run of tools/profile_all.sh under gpurun_out/<dir> into the summaries committed under profiles/:
#   <round>_<tag>.txt         kernel-trace stats + every PMC pass + the derived block (tools/prof_summary.py)
#   <round>_<tag>.json        machine-readable per-launch means of the H-sweep (bench.py reads the c3 one for roofline.traffic)
#   <round>_<tag>_bench.json  the bench line of the profiled --stats run
# usage: tools/refresh_profiles.sh <dir under gpurun_out, e.g. r3i/prof> <round tag, e.g. r3>
# Every summary's first lines name the commit of the sources the profiled library was built from: the last commit that
# touches the product or the bench (`git log -1 --format=%h -- nbmf_mm_amd include bench.py` -- run it at HEAD to compare;
# later commits of the round add only documents and these files).  Refuses to run on a dirty product tree, and -- the
# library being a git-ignored artefact that travels to the GPU box on its own -- refuses a run whose bench line does not
# carry the content hash of THIS tree's sources (libnbmf_hip.so's nbmf_source_hash, printed by bench.py as
# library.source_hash; tools/src_hash.sh computes it from the tree): the binary that was profiled is then the build of
# the commit named.
O=gpurun_out/$1; R=$2
if ! git diff --quiet HEAD -- nbmf_mm_amd include bench.py; then echo "uncommitted changes under nbmf_mm_amd/ include/ bench.py: commit first"; exit 1; fi
SRC=$(git log -1 --format=%h -- nbmf_mm_amd include bench.py)
HASH=$(bash tools/src_hash.sh)
for d in $O/*/; do
  got=$(grep '^{' $d/bench_stats.log | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read()).get('library',{}).get('source_hash'))")
  if [ "$got" != "$HASH" ]; then echo "$d: profiled library was built from sources $got, the tree's are $HASH: not refreshed"; exit 1; fi
done
declare -A NAME=([c3]=c3_k64_masked)
for d in $O/*/; do
  t=$(basename $d); n=${NAME[$t]:-$t}
  { echo "# sources: commit $SRC (git log -1 --format=%h -- nbmf_mm_amd include bench.py); the profiled library carries their content hash $HASH (nbmf_source_hash = tools/src_hash.sh)"; python tools/prof_summary.py $O/$t profiles/${R}_$n.json; } > profiles/${R}_$n.txt
  grep '^{' $O/$t/bench_stats.log | tail -1 > profiles/${R}_${n}_bench.json
  python - $R $n <<'P'
import json,sys
R,t=sys.argv[1:3]
d=json.loads(open(f"profiles/{R}_{t}_bench.json").read().strip().splitlines()[-1]); r=d['roofline']
print("%-16s it/s %8.1f norm %8.1f hpass %.3f wpass %.3f frac %.3f exec %.3f wexec %.3f" % (t, d['value'], d['normalize_value'], r['hpass_ms'], r['wpass_ms'], r['frac'], r['executed_frac'], r['wpass_executed_frac']))
P
done
