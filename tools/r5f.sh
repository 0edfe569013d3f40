#!/bin/bash
# round 5: the clock the chip holds inside the sweeps' loops under sustained load (NBMF_PASS_TRACE=16: every 16th launch traced)
export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-f64-leg --no-u8-leg"
NBMF_PASS_TRACE=16 $B --M 16384 --K 16 --storage f64 --steps 200 --warmup 5 > $O/k16.json 2> $O/k16.err || exit 1
NBMF_PASS_TRACE=16 $B --M 16384 --K 32 --storage f64 --steps 200 --warmup 5 > $O/k32.json 2> $O/k32.err || exit 1
NBMF_PASS_TRACE=16 $B --storage f64 --steps 48 --warmup 3 > $O/k64_f64.json 2> $O/k64_f64.err || exit 1
NBMF_PASS_TRACE=16 $B --storage f64w --steps 48 --warmup 3 > $O/k64_f64w.json 2> $O/k64_f64w.err || exit 1
NBMF_PASS_TRACE=16 $B --steps 48 --warmup 3 > $O/k64_bin.json 2> $O/k64_bin.err || exit 1
NBMF_PASS_TRACE=16 $B --M 8192 --N 8192 --K 32 --no-mask --projection normalize --steps 400 --warmup 5 > $O/c2.json 2> $O/c2.err || exit 1
for f in k16 k32 k64_f64 k64_f64w k64_bin c2; do echo "== $f"; grep "shader clock" $O/$f.err | sed 's/.*pass</pass</; s/ workgroups.*|/ /' | sort | uniq -c | sort -rn | head -12; done > $O/clocks.txt
cat $O/clocks.txt
bash tools/clock_probe.sh $O/smi_k16.txt -- $B --M 16384 --K 16 --storage f64 --steps 3000 --warmup 5; tail -8 $O/smi_k16.txt
bash tools/clock_probe.sh $O/smi_k64_f64w.txt -- $B --storage f64w --steps 400 --warmup 5; tail -8 $O/smi_k64_f64w.txt
