#!/bin/bash
# The round's committed rocprofv3 evidence, produced on the GPU box:   bash tests/profile_round.sh <tag>
# Writes gpurun_out/<tag>_*: kernel-trace step summary + stats of the bench command, PMC passes (separate runs, --kernel-trace
# only, as gpurun requires): GEMM FETCH_SIZE / WRITE_SIZE over one un-tuned training step, attention MFMA-busy, LayerNorm
# FETCH / WRITE on a working set larger than the Infinity Cache.  Copy what is to be judged into profiles/.
set -u
TAG=${1:-r02}
WHAT=${2:-all}      # "gemm": only the step trace and the GEMM passes (1, 2, 2b); "all": plus attention and LayerNorm
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-eager-baseline --no-rollout --no-loss-parity --no-other-configs --no-integration-levels"
# 1. kernel trace + stats of the bench command (3 timed steps)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_step -f csv -- $BENCH --steps 3 --warmup 1 --no-roofline > $OUT/${TAG}_bench_under_rocprof.log 2>&1
python $REPO/tests/prof_summary.py $OUT/${TAG}_prof_step $OUT/${TAG}_step_summary.txt > /dev/null 2>&1
cp $(find $OUT/${TAG}_prof_step -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null
# 2. GEMM traffic of the TUNED step: the tuner's locked choices are saved by an ordinary run and replayed (no trials) under the
#    counters, FETCH and WRITE in separate passes; the same run writes the per-shape breakdown the algorithmic bytes come from
DVLA_GEMM_BREAKDOWN=$OUT/${TAG}_gemm_breakdown.json $BENCH --steps 3 --warmup 1 --save-plan $OUT/${TAG}_gemm_plan.json > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_err.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_pmc_$C -f csv -- $BENCH --steps 2 --warmup 0 --plan $OUT/${TAG}_gemm_plan.json --no-roofline --no-fwd --no-rollout > $OUT/${TAG}_pmc_$C.log 2>&1
  python $REPO/tests/pmc_summary.py $OUT/${TAG}_pmc_$C $OUT/${TAG}_pmc_$C.json > $OUT/${TAG}_pmc_$C.txt 2>&1
done
python $REPO/tests/pmc_traffic.py $OUT/${TAG}_pmc_FETCH_SIZE.json $OUT/${TAG}_pmc_WRITE_SIZE.json $OUT/${TAG}_gemm_breakdown.json $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.log 2>&1
# 2b. GEMM matrix-pipe busy over the tuned step
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/${TAG}_pmc_gemm_mfma -f csv -- $BENCH --steps 2 --warmup 0 --plan $OUT/${TAG}_gemm_plan.json --no-roofline --no-fwd --no-rollout > $OUT/${TAG}_pmc_gemm_mfma.log 2>&1
python $REPO/tests/pmc_summary.py $OUT/${TAG}_pmc_gemm_mfma $OUT/${TAG}_pmc_gemm_mfma.json > $OUT/${TAG}_pmc_gemm_mfma.txt 2>&1
if [ "$WHAT" = "all" ]; then
# 3. attention: matrix-pipe busy per kernel
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/${TAG}_pmc_attn -f csv -- python $REPO/tests/gpu_pmc_attn.py > $OUT/${TAG}_pmc_attn.log 2>&1
python $REPO/tests/pmc_summary.py $OUT/${TAG}_pmc_attn $OUT/${TAG}_pmc_attn.json > $OUT/${TAG}_pmc_attn.txt 2>&1
# 4. LayerNorm: HBM bytes on > 256 MiB working sets, ONE SHAPE PER PASS (0: 353024 x 768, 1: 166656 x 1024)
for SH in 0 1; do
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_pmc_ln${SH}_$C -f csv -- python $REPO/tests/gpu_pmc_ln.py $SH > $OUT/${TAG}_pmc_ln${SH}_$C.log 2>&1
  python $REPO/tests/pmc_summary.py $OUT/${TAG}_pmc_ln${SH}_$C $OUT/${TAG}_pmc_ln${SH}_$C.json > $OUT/${TAG}_pmc_ln${SH}_$C.txt 2>&1
  rm -rf $OUT/${TAG}_pmc_ln${SH}_$C
done
done
fi
# keep the merge small: drop the raw traces
rm -rf $OUT/${TAG}_pmc_gemm_mfma $OUT/${TAG}_prof_step $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_attn
ls -la $OUT | head -40
