./build/gemm_probe --cases model --variants 8,9,10 --check-only --full-check > gpurun_out/r05d_check.txt 2>&1
echo "checks ok: $(grep -c '"ok": true' gpurun_out/r05d_check.txt) bad: $(grep -c '"ok": false' gpurun_out/r05d_check.txt)"
grep '"ok": false' gpurun_out/r05d_check.txt | head -5
./build/gemm_probe --cases model --variants 8,9,10 --no-check --iters 5 --rounds 3 > gpurun_out/r05d_probe_model.txt 2>&1
timeout 900 python -m pytest tests/test_collate.py tests/test_torch_ddp_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "collators or torch_ddp or lockstep or recovers" > gpurun_out/r05d_newtests.txt 2>&1
tail -15 gpurun_out/r05d_newtests.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-rollout > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err
tail -c 1500 gpurun_out/r05d_bench.json
