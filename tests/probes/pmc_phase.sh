#!/bin/bash
# Counter passes on gemm_phase_kernel (GPU box):  bash tests/probes/pmc_phase.sh <tag> [variant]
# LDS conflicts / LDS issue stalls / matrix-pipe busy per operand layout, from the standalone probe (no torch); each pass is
# its own rocprofv3 run with --kernel-trace only (gpurun's rule).  Output: gpurun_out/<tag>_pmc_phase_*.txt
set -u
TAG=${1:-r05}
VAR=${2:-8}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_WAIT_INST_[A-Z_]*" | sort -u > $OUT/${TAG}_counters_available.txt 2>&1
PROBE="$REPO/build/gemm_probe --cases pmc --variants $VAR --no-check --iters 2 --rounds 1"
pass() {   # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/${TAG}_pp_$name -f csv -- $PROBE > $OUT/${TAG}_pp_$name.log 2>&1
  python $REPO/tests/pmc_summary.py --full-names $OUT/${TAG}_pp_$name $OUT/${TAG}_pmc_phase_$name.json > $OUT/${TAG}_pmc_phase_$name.txt 2>&1
  rm -rf $OUT/${TAG}_pp_$name
}
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass lds2 SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES
pass mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES
ls -la $OUT | grep ${TAG}_
