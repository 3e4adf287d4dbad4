B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-rollout --no-other-configs --no-loss-parity --no-fwd"
for r in 1 2; do
  for w in old new; do
    if [ $w = old ]; then export DVLA_LIB=$PWD/build/ab/new.so; else unset DVLA_LIB; fi
    timeout 300 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],2))"
  done
done > gpurun_out/r05i_step_ab.txt 2>&1
unset DVLA_LIB
cat gpurun_out/r05i_step_ab.txt
DVLA_SAVE_OTHER_PLANS=1 DVLA_GEMM_BREAKDOWN=$PWD/gpurun_out/r05i_gemm_breakdown.json timeout 900 python bench.py --steps 20 --warmup 5 --save-plan gpurun_out/r05i_gemm_plan.json > gpurun_out/r05i_bench_default.json 2> gpurun_out/r05i_bench_default.err
cp profiles/r05_gemm_plan_*.json gpurun_out/ 2>/dev/null
tail -1 gpurun_out/r05i_bench_default.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], d['eager_rocm_baseline']['ours_over_eager'], d['rollout']['value'], d['rollout']['single_episode_ms_per_step']); print(json.dumps(d['other_configs'])[:900])"
