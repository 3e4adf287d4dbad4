cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export DVLA_PARITY_REPORT=$PWD/gpurun_out/r04_parity_rollout.jsonl
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "skinny or rollout or golden[B] or golden[E] or golden[F] or golden[C] or (check_gemm and not plan)" > gpurun_out/g8_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/g8_pytest.log
tail -15 gpurun_out/g8_pytest.log
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/rt -f csv -- python $R/tests/gpu_rollout_trace.py run > $R/gpurun_out/g8_run.log 2>&1
cd $R
python tests/gpu_rollout_trace.py summary gpurun_out/rt gpurun_out/r04_rollout_step_summary_after.txt | head -30
rm -rf gpurun_out/rt
timeout 600 python tests/gpu_rollout_bench.py 1 64 > gpurun_out/g8_rollout_bench.log 2>&1
tail -3 gpurun_out/g8_rollout_bench.log
