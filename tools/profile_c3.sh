#!/bin/bash
# On the GPU box: the rocprofv3 passes behind profiles/r1_c3_k64_masked.{txt,json} (one run per counter group,
# --kernel-trace only, as MI355X_MICROARCH.md prescribes).  usage: tools/profile_c3.sh <outdir under gpurun_out>
export TMPDIR=/tmp
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 10 --warmup 2 > $O/bench_stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B --steps 4 --warmup 1 > $O/bench_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B --steps 4 --warmup 1 > $O/bench_write.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- $B --steps 4 --warmup 1 > $O/bench_sq.log 2>&1
ls $O | tr "\n" " "
