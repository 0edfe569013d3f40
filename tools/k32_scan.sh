#!/bin/bash
# K = 32 sweeps at growing M (N = 8192, unmasked, normalize): is the per-tile cost of configs[1] (M = 8192) the loop's, or
# does a 0.2 ms kernel pay for its start, its tail and its slab writes?  Prints ms per pass and SIMD cycles per tile at 2.4 GHz.
for M in 8192 16384 32768 65536; do
  python bench.py --no-cpu-baseline --no-f64-leg --no-u8-leg --M $M --N 8192 --K 32 --no-mask --projection normalize --steps 60 --warmup 5 --event-stride 1 2>/dev/null | tail -1 > gpurun_out/k32.json
  python - $M <<'P'
import json,sys
M=int(sys.argv[1]); d=json.loads(open("gpurun_out/k32.json").read()); r=d["roofline"]
tiles=M*8192/256
print("M=%6d it/s %8.1f hpass %.4f ms (%.0f cyc/tile) wpass %.4f ms (%.0f cyc/tile) frac %.3f wfrac %.3f" % (M, d["value"], r["hpass_ms"], r["hpass_ms"]*1e-3*2.4e9*1024/tiles, r["wpass_ms"], r["wpass_ms"]*1e-3*2.4e9*1024/tiles, r["frac"], r["wpass_executed_frac"]))
P
done
