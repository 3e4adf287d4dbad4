for ks in b; do
DVLA_PROBE_KSUM=$ks ./build/gemm_probe --cases dw --variants 4,8,9 --iters 10 --rounds 3 > gpurun_out/r05h_ksum_$ks.txt 2>&1
echo "ok $(grep -c '"ok": true' gpurun_out/r05h_ksum_$ks.txt) bad $(grep -c '"ok": false' gpurun_out/r05h_ksum_$ks.txt)"
grep '"time"' gpurun_out/r05h_ksum_$ks.txt | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('ksum', r['time'], r['split_k'], ' '.join(f\"{k}:{v['us']}us/{v['TF']}\" for k,v in r.items() if k.startswith('v')))"
done
./build/gemm_probe --cases dw --variants 8,9 --no-check --iters 10 --rounds 3 2>&1 | grep '"time"' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('plain', r['time'], r['split_k'], ' '.join(f\"{k}:{v['us']}us/{v['TF']}\" for k,v in r.items() if k.startswith('v')))"
DVLA_PARITY_REPORT=$PWD/gpurun_out/r05h_parity.jsonl timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or misc or mlp or linear" > gpurun_out/r05h_kernels.txt 2>&1
tail -3 gpurun_out/r05h_kernels.txt
DVLA_GEMM_BREAKDOWN=$PWD/gpurun_out/r05h_gemm_breakdown.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-rollout --no-other-configs --save-plan gpurun_out/r05h_gemm_plan.json > gpurun_out/r05h_bench.json 2> gpurun_out/r05h_bench.err
tail -c 300 gpurun_out/r05h_bench.json
