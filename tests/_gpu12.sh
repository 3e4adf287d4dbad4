cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tests/gpu_skinny_perf.py 2> gpurun_out/g12_skinny.err | cut -c1-400
tail -2 gpurun_out/g12_skinny.err
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "rollout or skinny or gemm_ln" > gpurun_out/g12_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/g12_pytest.log
grep -v Warning gpurun_out/g12_pytest.log | tail -5 | cut -c1-300
timeout 600 python tests/gpu_rollout_bench.py 1 > gpurun_out/g12_rollout_bench.log 2>&1
tail -1 gpurun_out/g12_rollout_bench.log | cut -c1-300
