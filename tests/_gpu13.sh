cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export DVLA_PARITY_REPORT=$PWD/gpurun_out/r04_parity_actbwd.jsonl
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "act_bwd or linear or mlp or dropout_on or golden[C] or gradients_vs_oracle[C] or gradients_vs_oracle[A]" > gpurun_out/g13_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/g13_pytest.log
grep -v Warning gpurun_out/g13_pytest.log | tail -6 | cut -c1-400
timeout 600 python tests/gpu_skinny_perf.py 2> gpurun_out/g13_skinny.err | cut -c1-330
timeout 600 python tests/gpu_rollout_bench.py 1 > gpurun_out/g13_rollout_bench.log 2>&1
tail -1 gpurun_out/g13_rollout_bench.log | cut -c1-300
DVLA_GEMM_BREAKDOWN=$PWD/gpurun_out/g13_gemm_breakdown.json timeout 600 python bench.py --no-cpu-baseline --no-eager-baseline --no-rollout --no-loss-parity --save-plan gpurun_out/g13_plan.json > gpurun_out/g13_bench.json 2> gpurun_out/g13_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/g13_bench.json')); print('ms_per_step', d['ms_per_step'], 'gemm_ms', d['roofline']['gemm_ms_per_step'], 'TF', d['roofline']['achieved'], d['roofline']['tuner_wins_by_problem_key'])
k=json.load(open('gpurun_out/g13_plan.json')); print([(key[0],key[1],key[2],key[5],key[15],v) for key,v in k if key[2]==20832])
PY
