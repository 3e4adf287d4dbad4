./build/gemm_probe --cases model --variants 8,9,10 --check-only --full-check > gpurun_out/r05e_check.txt 2>&1
echo "checks ok: $(grep -c '"ok": true' gpurun_out/r05e_check.txt) bad: $(grep -c '"ok": false' gpurun_out/r05e_check.txt)"
grep '"ok": false' gpurun_out/r05e_check.txt | head -5
./build/gemm_probe --cases nn --variants 8,40,42,43 --no-check --iters 5 --rounds 5 > gpurun_out/r05e_probe_nn.txt 2>&1
./build/gemm_probe --cases model --variants 8,9,10 --no-check --iters 5 --rounds 3 > gpurun_out/r05e_probe_model.txt 2>&1
DVLA_STAMPS_ALL=1 ./build/gemm_probe --stamps 1024 --stamp-variants 89,41 > gpurun_out/r05e_stamps_all.txt 2>&1
./build/gemm_probe --stamps 1024 --stamp-variants 89,41 > gpurun_out/r05e_stamps.txt 2>&1
DVLA_PARITY_REPORT=$PWD/gpurun_out/r05e_parity_kernels.jsonl timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/r05e_kernels.txt 2>&1
tail -5 gpurun_out/r05e_kernels.txt
DVLA_PARITY_REPORT=$PWD/gpurun_out/r05e_parity_lockstep.jsonl timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -k "lockstep and all" > gpurun_out/r05e_lockstep.txt 2>&1
tail -3 gpurun_out/r05e_lockstep.txt
DVLA_GEMM_BREAKDOWN=$PWD/gpurun_out/r05e_gemm_breakdown.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-rollout --save-plan gpurun_out/r05e_gemm_plan.json > gpurun_out/r05e_bench.json 2> gpurun_out/r05e_bench.err
tail -c 600 gpurun_out/r05e_bench.json
