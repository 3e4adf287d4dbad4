cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export DVLA_PARITY_REPORT=$PWD/gpurun_out/r04_parity_rollout2.jsonl
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -k "rollout or golden[B] or golden[F] or golden[E]" > gpurun_out/g11_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/g11_pytest.log
grep -v Warning gpurun_out/g11_pytest.log | tail -8 | cut -c1-400
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/rt -f csv -- python $R/tests/gpu_rollout_trace.py run > $R/gpurun_out/g11_run.log 2>&1
cd $R
python tests/gpu_rollout_trace.py summary gpurun_out/rt gpurun_out/r04_rollout_step_summary_after.txt | head -24
rm -rf gpurun_out/rt
timeout 600 python tests/gpu_rollout_bench.py 1 > gpurun_out/g11_rollout_bench.log 2>&1
tail -1 gpurun_out/g11_rollout_bench.log | cut -c1-300
