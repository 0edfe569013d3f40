#!/bin/bash
# On the GPU box: EVERY rocprofv3 pass behind profiles/ for the configurations DESIGN.md quotes -- per tag one
# --kernel-trace --stats run plus one --pmc run per counter group (separate runs, --kernel-trace only beside --pmc,
# program directly after `--`, as MI355X_MICROARCH.md prescribes).
#   usage: tools/profile_all.sh <outdir under gpurun_out> <tag> [<tag> ...]
#   c3          BASELINE configs[2] (the bench line): 65536 x 8192, K=64, 90 % mask, duchi
#   c2          BASELINE configs[1]: 8192 x 8192, K=32, unmasked, normalize (stats run: 500 iterations)
#   general     configs[2] on the 8-byte storage path (--storage f64: real-valued V semantics, bool mask)
#   generalw    ... with float64 weight tiles (--storage f64w: 16 bytes per entry)
#   general_k16 / general_k32   the 8-byte storage path below K = 64: 16384 x 8192, masked, K = 16 / 32
#   c2_general  BASELINE configs[1]'s shape (8192 x 8192, K=32, unmasked, normalize) on the 8-byte storage path
#   c5shape     BASELINE configs[4]'s shape on one GPU: internal 17000 x 360000, K=128, device-generated
#   c4shard_peer / c4shard_rccl   BASELINE configs[3]'s per-rank shard 32768 x 8192, K=64, 1-rank communicator
#   shard8192   the strong-scaling shard of configs[2]: 8192 x 8192, K=64, no communicator
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-f64-leg --no-u8-leg"
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES"
SQ3="SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM"
for tag in "$@"; do
  S=""   # arguments of the --stats run where they differ from the counter runs
  case $tag in
    c3) A="--steps 6 --warmup 2"; S="--steps 10 --warmup 2";;
    c2) A="--M 8192 --N 8192 --K 32 --no-mask --projection normalize --steps 50 --warmup 5"; S="--M 8192 --N 8192 --K 32 --no-mask --projection normalize --steps 500 --warmup 5";;
    general) A="--storage f64 --steps 6 --warmup 2";;
    generalw) A="--storage f64w --steps 6 --warmup 2";;
    general_k16) A="--M 16384 --K 16 --storage f64 --steps 20 --warmup 3";;
    general_k32) A="--M 16384 --K 32 --storage f64 --steps 20 --warmup 3";;
    c2_general) A="--M 8192 --N 8192 --K 32 --no-mask --projection normalize --storage f64 --steps 50 --warmup 5";;
    c5shape) A="--device-data --M 17000 --N 360000 --K 128 --projection normalize --steps 4 --warmup 1";;
    c4shard_peer) A="--M 32768 --force-comm --transport peer --steps 10 --warmup 3";;
    c4shard_rccl) A="--M 32768 --force-comm --transport rccl --steps 10 --warmup 3";;
    shard8192) A="--M 8192 --steps 30 --warmup 5";;
    *) echo "unknown tag $tag"; exit 1;;
  esac
  [ -z "$S" ] && S="$A"
  rm -rf $O/$tag; mkdir -p $O/$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/stats -- $B $S > $O/$tag/bench_stats.log 2>&1 || { echo "FAILED $tag stats"; tail -5 $O/$tag/bench_stats.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$tag/pmc_fetch -- $B $A > $O/$tag/bench_fetch.log 2>&1 || { echo "FAILED $tag fetch"; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/$tag/pmc_write -- $B $A > $O/$tag/bench_write.log 2>&1 || { echo "FAILED $tag write"; exit 1; }
  rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/$tag/pmc_sq -- $B $A > $O/$tag/bench_sq.log 2>&1 || { echo "FAILED $tag sq"; exit 1; }
  rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/$tag/pmc_sq2 -- $B $A > $O/$tag/bench_sq2.log 2>&1 || { echo "FAILED $tag sq2"; tail -3 $O/$tag/bench_sq2.log; exit 1; }
  rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $O/$tag/pmc_sq3 -- $B $A > $O/$tag/bench_sq3.log 2>&1 || { echo "FAILED $tag sq3"; tail -3 $O/$tag/bench_sq3.log; exit 1; }
  # keep only what the summaries need (the traces are large)
  find $O/$tag -name '*_kernel_trace.csv' -delete; find $O/$tag -name '*_agent_info.csv' -delete
  echo "done $tag: $(grep '^{' $O/$tag/bench_stats.log | tail -1 | cut -c1-120)"
done
