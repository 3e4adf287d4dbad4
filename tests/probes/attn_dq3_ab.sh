set -u
OUT=gpurun_out; mkdir -p $OUT
DVLA_LIB=$PWD/build/ab/libdvla_dq3.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -2
for i in 1 2; do
timeout 90 python tests/gpu_attn_perf.py > $OUT/r06b_attn_perf_w4_$i.jsonl 2>/dev/null
DVLA_LIB=$PWD/build/ab/libdvla_dq3.so timeout 90 python tests/gpu_attn_perf.py > $OUT/r06b_attn_perf_w3_$i.jsonl 2>/dev/null
done
python - <<'PY'
import json
def rd(p): return [json.loads(l) for l in open(p) if l.startswith("{")]
for i in (1, 2):
    new, old = rd(f"gpurun_out/r06b_attn_perf_w3_{i}.jsonl"), rd(f"gpurun_out/r06b_attn_perf_w4_{i}.jsonl")
    for n, o in zip(new, old):
        tag = f"B={n['B']} H={n['H']} L={n['L']} {n['mask']}"
        print(f"{tag:28s} bwd w4 {o['bwd_us']:7.1f} -> w3 {n['bwd_us']:7.1f}" + (f"   p=0.1: bwd {o['bwd_us_dropout']:7.1f} -> {n['bwd_us_dropout']:7.1f}" if 'bwd_us_dropout' in n else ""))
PY
