#!/bin/bash
# Round 6: the instrumented step of bench.py showed the step's 706 GEMM launches at ~92 ms with the round-6 attention kernels and ~105 ms
# with the library from before them -- same GEMM code objects, same plan, same box.  Per-shape breakdowns of both, side by side.
set -u
OUT=gpurun_out; mkdir -p $OUT
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-eager-baseline --no-rollout --no-loss-parity --no-other-configs --no-integration-levels"
timeout 200 $B --save-plan $OUT/r06g_plan.json > $OUT/r06g_bench_tune.json 2> $OUT/r06g.err
for i in 1 2; do
DVLA_GEMM_BREAKDOWN=$OUT/r06g_bd_old$i.json DVLA_LIB=$PWD/build/ab/libdvla_before_r6attn.so timeout 120 $B --plan $OUT/r06g_plan.json > $OUT/r06g_bench_old$i.json 2>> $OUT/r06g.err
DVLA_GEMM_BREAKDOWN=$OUT/r06g_bd_new$i.json timeout 120 $B --plan $OUT/r06g_plan.json > $OUT/r06g_bench_new$i.json 2>> $OUT/r06g.err
done
rocm-smi --showpower --showmaxpower --showperflevel --showtemp 2>/dev/null | grep -v "^=\|^$" | head -20
python - <<'PY'
import json
def line(f):
    d=json.loads(open(f).read().strip().split("\n")[-1]); r=d.get("roofline") or {}
    return round(d["ms_per_step"],2), r.get("gemm_ms_per_step")
for t in ["tune","old1","new1","old2","new2"]:
    print(t, line(f"gpurun_out/r06g_bench_{t}.json"))
o=json.load(open("gpurun_out/r06g_bd_old1.json")); n=json.load(open("gpurun_out/r06g_bd_new1.json"))
key=lambda r:(r["M"],r["N"],r["K"],r["a_trans"],r["b_trans"],r["split_k"],r["epilogue"],r["variant"])
od={key(r):r for r in o}
rows=[]
for r in n:
    k=key(r)
    if k in od: rows.append((od[k]["ms"]-r["ms"], od[k]["ms"], r["ms"], k))
rows.sort(reverse=True)
print("total old", sum(r["ms"] for r in o), "new", sum(r["ms"] for r in n), "matched", len(rows), "of", len(n))
for d,a,b,k in rows[:25]: print(f"{a:7.3f} -> {b:7.3f}  ({d:+.3f})  {k}")
PY
