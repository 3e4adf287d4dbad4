#!/bin/bash
# Same-box A/B of attention builds: tests/probes/attn_perf_mini.py once per library in build/ab/ and for the tree's own build,
# two rounds interleaved, then the attention parity cases under each variant library.
#   gpurun --timeout 420 -- bash tests/probes/attn_variants.sh
set -u
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/r05x_attn_variants.jsonl
for rep in 1 2; do
  timeout 60 python tests/probes/attn_perf_mini.py >> $OUT/r05x_attn_variants.jsonl 2>> $OUT/r05x_attn_variants.err
  for L in build/ab/*.so; do
    DVLA_LIB=$PWD/$L timeout 60 python tests/probes/attn_perf_mini.py >> $OUT/r05x_attn_variants.jsonl 2>> $OUT/r05x_attn_variants.err
  done
done
cat $OUT/r05x_attn_variants.jsonl
for L in ${PARITY_LIBS:-}; do
  DVLA_LIB=$PWD/$L timeout 150 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > $OUT/r05x_parity_$(basename $L .so).log 2>&1
  echo "$L parity rc=$?" | tee -a $OUT/r05x_rc.txt
done
