cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export DVLA_PARITY_REPORT=$PWD/gpurun_out/r04_parity_gemm.jsonl
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "gemm or skinny or rollout or golden[B] or golden[F] or golden[C]" > gpurun_out/g10_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/g10_pytest.log
grep -v Warning gpurun_out/g10_pytest.log | tail -12 | cut -c1-400
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/rt -f csv -- python $R/tests/gpu_rollout_trace.py run > $R/gpurun_out/g10_run.log 2>&1
cd $R
python tests/gpu_rollout_trace.py summary gpurun_out/rt gpurun_out/r04_rollout_step_summary_after.txt | head -24
rm -rf gpurun_out/rt
timeout 600 python tests/gpu_rollout_bench.py 1 64 > gpurun_out/g10_rollout_bench.log 2>&1
tail -2 gpurun_out/g10_rollout_bench.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-eager-baseline --no-rollout --no-loss-parity --save-plan gpurun_out/g10_plan.json > gpurun_out/g10_bench.json 2> gpurun_out/g10_bench.err
DVLA_GEMM_BREAKDOWN=$PWD/gpurun_out/g10_gemm_breakdown.json python - <<'PY'
import json
d=json.load(open('gpurun_out/g10_bench.json')); print('ms_per_step', d['ms_per_step'], 'gemm_ms', d['roofline']['gemm_ms_per_step'], 'TF', d['roofline']['achieved'], d['roofline']['tuner_wins_by_problem_key'])
PY
