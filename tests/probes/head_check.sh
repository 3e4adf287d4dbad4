#!/bin/bash
# A short check of a finished tree on the GPU box: the benchmarked head set's whole-model fixture and its reference gradients, then the
# default bench command (what the driver runs), with its wall time.   gpurun --timeout 330 -- bash tests/probes/head_check.sh
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 150 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "golden and C or gradients_vs_oracle and C" > $OUT/r05w_model.log 2>&1; echo "model rc=$?" | tee $OUT/r05w_rc.txt; tail -2 $OUT/r05w_model.log
S=$(date +%s); timeout 240 python bench.py > $OUT/r05w_bench_default.json 2> $OUT/r05w_bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $OUT/r05w_rc.txt
