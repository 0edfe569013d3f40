#!/bin/bash
# How finely should a sweep be cut into chunks (= workgroups per strip group)?  NBMF_MIN_BLOCKS / NBMF_TARGET_WGS swept
# for one bench configuration.  usage: tools/sweep_chunks.sh "<bench args>" "<min_blocks list>" "<target list>"
CFG=$1
for mb in $2; do for tg in $3; do
  NBMF_MIN_BLOCKS=$mb NBMF_TARGET_WGS=$tg python bench.py --no-cpu-baseline --no-f64-leg --no-u8-leg $CFG 2>/dev/null | tail -1 > gpurun_out/ab.json
  echo -n "[$CFG] min_blocks $mb target $tg: "; python tools/benchline.py gpurun_out/ab.json | cut -c50-
done; done
