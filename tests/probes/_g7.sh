for ks in a b; do
DVLA_PROBE_KSUM=$ks ./build/gemm_probe --cases dw --variants 4,8,9 --iters 10 --rounds 3 > gpurun_out/r05g_ksum_$ks.txt 2>&1
grep -c '"ok": true' gpurun_out/r05g_ksum_$ks.txt; grep '"ok": false' gpurun_out/r05g_ksum_$ks.txt | head -3
grep '"time"' gpurun_out/r05g_ksum_$ks.txt | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('ksum', r['time'], r['split_k'], ' '.join(f\"{k}:{v['us']}us/{v['TF']}\" for k,v in r.items() if k.startswith('v')))"
done
DVLA_PARITY_REPORT=$PWD/gpurun_out/r05g_parity.jsonl timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > gpurun_out/r05g_kernels.txt 2>&1
tail -3 gpurun_out/r05g_kernels.txt
DVLA_PARITY_REPORT=$PWD/gpurun_out/r05g_parity_model.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -k "lockstep or gradients" > gpurun_out/r05g_model.txt 2>&1
tail -3 gpurun_out/r05g_model.txt
DVLA_GEMM_BREAKDOWN=$PWD/gpurun_out/r05g_gemm_breakdown.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-rollout --no-other-configs --save-plan gpurun_out/r05g_gemm_plan.json > gpurun_out/r05g_bench.json 2> gpurun_out/r05g_bench.err
tail -c 400 gpurun_out/r05g_bench.json
