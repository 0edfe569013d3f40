#!/bin/bash
# usage: tools/clock_probe.sh <outfile> -- <command...> : run the command, sampling the GPU's shader clock and socket power
# (rocm-smi) every 0.25 s while it runs.  Behind profiles/r3_clock_and_power_under_f64_mfma.txt: the sustained fp64-MFMA
# rate of this chip (71.7 TFLOP/s in a pure MFMA loop, tools/microbench.hip) is what the H-pass sweeps reach.
OUT=$1; shift; shift
"$@" > $OUT.cmd.log 2>&1 &
PID=$!
: > $OUT
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | tr -s ' ' | tr '\n' ';' >> $OUT
  echo >> $OUT
  sleep 0.25
done
wait $PID
