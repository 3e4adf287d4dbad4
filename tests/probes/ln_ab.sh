#!/bin/bash
# Same-box A/B of the LayerNorm kernels: the tree's library against an older build given as $1 (DVLA_LIB), tag $2; parity first.
set -u
OLD=$PWD/$1; TAG=$2
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "layernorm" 2>&1 | tail -2
for i in 1 2; do
DVLA_LIB=$OLD timeout 90 python tests/gpu_ln_perf.py > $OUT/${TAG}_ln_perf_old_$i.jsonl 2>/dev/null
timeout 90 python tests/gpu_ln_perf.py > $OUT/${TAG}_ln_perf_new_$i.jsonl 2>/dev/null
done
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
def rd(p): return [json.loads(l) for l in open(p) if l.startswith("{")]
for i in (1, 2):
    new, old = rd(f"gpurun_out/{tag}_ln_perf_new_{i}.jsonl"), rd(f"gpurun_out/{tag}_ln_perf_old_{i}.jsonl")
    for n, o in zip(new, old):
        print(f"{n['rows']:7d} x {n['cols']:5d}  fwd {o['fwd_us']:7.1f} -> {n['fwd_us']:7.1f} us ({o['fwd_TBps']:.2f} -> {n['fwd_TBps']:.2f} TB/s)   bwd {o['bwd_us']:7.1f} -> {n['bwd_us']:7.1f} us")
PY
