#!/bin/bash
# Round 5, attention fragment grouping: parity of every attention case, then the same-box A/B against the library built from the
# tree before the change (build/ab/libdvla_before_attn.so, loaded through DVLA_LIB), then a short training-step run.
#   (the older build: `cp dreamvla_amd/libdvla_hip.so build/ab/libdvla_before_attn.so` before making the change; build/ is not in git)
#   gpurun --timeout 540 -- bash tests/probes/attn_ab.sh
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 270 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > $OUT/r05y_attn_parity.log 2>&1
echo "parity rc=$?" > $OUT/r05y_rc.txt
tail -3 $OUT/r05y_attn_parity.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05y_smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/r05y_rc.txt
timeout 90 python tests/gpu_attn_perf.py > $OUT/r05y_attn_perf_new.jsonl 2> $OUT/r05y_perf_new.err
echo "perf new rc=$?" >> $OUT/r05y_rc.txt
DVLA_LIB=$PWD/build/ab/libdvla_before_attn.so timeout 90 python tests/gpu_attn_perf.py > $OUT/r05y_attn_perf_old.jsonl 2> $OUT/r05y_perf_old.err
echo "perf old rc=$?" >> $OUT/r05y_rc.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-rollout --no-loss-parity --no-other-configs --plan profiles/r05_gemm_plan.json"
timeout 120 $B > $OUT/r05y_bench_new.json 2> $OUT/r05y_bench_new.err
echo "bench new rc=$?" >> $OUT/r05y_rc.txt
DVLA_LIB=$PWD/build/ab/libdvla_before_attn.so timeout 120 $B > $OUT/r05y_bench_old.json 2> $OUT/r05y_bench_old.err
echo "bench old rc=$?" >> $OUT/r05y_rc.txt
cat $OUT/r05y_rc.txt
