./build/gemm_probe --cases model --variants 4,6,7 --check-only --full-check > gpurun_out/r05f_check_ring.txt 2>&1
echo "ring checks ok: $(grep -c '"ok": true' gpurun_out/r05f_check_ring.txt) bad: $(grep -c '"ok": false' gpurun_out/r05f_check_ring.txt)"
mkdir -p gpurun_out/abdir/base gpurun_out/abdir/new
cp build/ab/base.so gpurun_out/abdir/base/libdvla_hip.so; cp build/ab/new.so gpurun_out/abdir/new/libdvla_hip.so
for r in 1 2; do for w in base new; do
  echo "== $w round $r"
  LD_LIBRARY_PATH=$PWD/gpurun_out/abdir/$w timeout 300 build/gemm_probe --cases model --no-check --variants 4,6,7 --iters 10 --rounds 3 2>&1 | grep "\"time\"" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['time'], r['epi'], r['v4']['us'], r['v6']['us'], r['v7']['us'])"
  DVLA_PROBE_KSUM=b LD_LIBRARY_PATH=$PWD/gpurun_out/abdir/$w timeout 300 build/gemm_probe --cases dw --no-check --variants 4 --iters 10 --rounds 3 2>&1 | grep "\"time\"" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('ksum', r['time'], r['split_k'], r['v4']['us'], r['v4']['TF'])"
done; done > gpurun_out/r05f_ring_ab.txt 2>&1
rm -rf gpurun_out/abdir
cat gpurun_out/r05f_ring_ab.txt | head -60
DVLA_PARITY_REPORT=$PWD/gpurun_out/r05f_parity.jsonl timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or attention" > gpurun_out/r05f_kernels.txt 2>&1
tail -3 gpurun_out/r05f_kernels.txt
DVLA_PARITY_REPORT=$PWD/gpurun_out/r05f_parity_model.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -k "golden and D or lockstep" > gpurun_out/r05f_model.txt 2>&1
tail -3 gpurun_out/r05f_model.txt
timeout 600 python bench.py --steps 10 --warmup 3 --torch-ddp --torch-adamw --no-cpu-baseline --no-eager-baseline --no-rollout --no-loss-parity > gpurun_out/r05f_bench_unchanged_caller.json 2> gpurun_out/r05f_bench_unchanged_caller.err
tail -c 800 gpurun_out/r05f_bench_unchanged_caller.json
DVLA_SAVE_OTHER_PLANS=1 DVLA_GEMM_BREAKDOWN=$PWD/gpurun_out/r05f_gemm_breakdown.json timeout 900 python bench.py --steps 20 --warmup 5 --save-plan gpurun_out/r05f_gemm_plan.json > gpurun_out/r05f_bench_default.json 2> gpurun_out/r05f_bench_default.err
tail -c 3000 gpurun_out/r05f_bench_default.json
cp profiles/r05_gemm_plan_*.json gpurun_out/ 2>/dev/null
