#!/bin/bash
# Same-box A/B of the attention kernels: the tree's library against an older build given as $1 (loaded through DVLA_LIB), tag $2.
#   gpurun --timeout 900 -- bash tests/probes/attn_ab_generic.sh build/ab/libdvla_dq3.so r06c
set -u
OLD=$PWD/$1; TAG=$2
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -2
for i in 1 2; do
DVLA_LIB=$OLD timeout 90 python tests/gpu_attn_perf.py > $OUT/${TAG}_attn_perf_old_$i.jsonl 2>/dev/null
timeout 90 python tests/gpu_attn_perf.py > $OUT/${TAG}_attn_perf_new_$i.jsonl 2>/dev/null
done
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
def rd(p): return [json.loads(l) for l in open(p) if l.startswith("{")]
for i in (1, 2):
    new, old = rd(f"gpurun_out/{tag}_attn_perf_new_{i}.jsonl"), rd(f"gpurun_out/{tag}_attn_perf_old_{i}.jsonl")
    for n, o in zip(new, old):
        t = f"B={n['B']} H={n['H']} L={n['L']} {n['mask']}"
        print(f"{t:28s} fwd {o['fwd_us']:7.1f} -> {n['fwd_us']:7.1f}  bwd {o['bwd_us']:7.1f} -> {n['bwd_us']:7.1f}" +
              (f"   p=0.1: fwd {o['fwd_us_dropout']:7.1f} -> {n['fwd_us_dropout']:7.1f}  bwd {o['bwd_us_dropout']:7.1f} -> {n['bwd_us_dropout']:7.1f}" if 'fwd_us_dropout' in n else ""))
PY
