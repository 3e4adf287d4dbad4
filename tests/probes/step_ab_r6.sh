set -u
OUT=gpurun_out; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-rollout --no-loss-parity --no-other-configs --no-integration-levels"
rocm-smi --showclocks 2>/dev/null | head -20 > $OUT/r06m_clocks.txt
timeout 200 $B --save-plan $OUT/r06m_plan.json > $OUT/r06m_bench_new.json 2> $OUT/r06m_bench_new.err
for i in 1 2; do
DVLA_LIB=$PWD/build/ab/libdvla_before_perm.so timeout 120 $B --plan $OUT/r06m_plan.json > $OUT/r06m_bench_old$i.json 2> $OUT/r06m_bench_old.err
timeout 120 $B --plan $OUT/r06m_plan.json > $OUT/r06m_bench_new$i.json 2> $OUT/r06m_bench_new.err
done
python - <<'PY'
import json
for f in ["new","old1","new1","old2","new2"]:
    try:
        d=json.loads(open(f"gpurun_out/r06m_bench_{f}.json").read().strip().split("\n")[-1]); r=d.get("roofline") or {}
        print(f, round(d["ms_per_step"],2), round(d["value"],1), r.get("gemm_ms_per_step"), r.get("achieved"))
    except Exception as e: print(f, e)
PY
cat $OUT/r06m_clocks.txt | head -12
