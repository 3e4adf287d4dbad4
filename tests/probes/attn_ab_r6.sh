#!/bin/bash
# Round 6, attention instruction count (scale and 1 / keep folded into one fma, opaque in-place mask extract, one branch around the
# dK/dV kernel's dropout): parity of every attention case, then the same-box A/B against the library built from the tree before
# the change (build/ab/libdvla_before_r6attn.so, loaded through DVLA_LIB; build/ is not in git).
#   gpurun --timeout 600 -- bash tests/probes/attn_ab_r6.sh
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 330 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > $OUT/r06a_attn_parity.log 2>&1
echo "parity rc=$?" > $OUT/r06a_rc.txt
tail -3 $OUT/r06a_attn_parity.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r06a_smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/r06a_rc.txt
for i in 1 2; do
timeout 90 python tests/gpu_attn_perf.py > $OUT/r06a_attn_perf_new$i.jsonl 2> $OUT/r06a_perf_new.err
echo "perf new rc=$?" >> $OUT/r06a_rc.txt
DVLA_LIB=$PWD/build/ab/libdvla_before_r6attn.so timeout 90 python tests/gpu_attn_perf.py > $OUT/r06a_attn_perf_old$i.jsonl 2> $OUT/r06a_perf_old.err
echo "perf old rc=$?" >> $OUT/r06a_rc.txt
done
cat $OUT/r06a_rc.txt
python - <<'PY'
import json
def rd(p): return [json.loads(l) for l in open(p) if l.startswith("{")]
for i in (1, 2):
    new, old = rd(f"gpurun_out/r06a_attn_perf_new{i}.jsonl"), rd(f"gpurun_out/r06a_attn_perf_old{i}.jsonl")
    for n, o in zip(new, old):
        tag = f"B={n['B']} H={n['H']} L={n['L']} {n['mask']}"
        print(f"{tag:28s} fwd {o['fwd_us']:7.1f} -> {n['fwd_us']:7.1f}  bwd {o['bwd_us']:7.1f} -> {n['bwd_us']:7.1f}" +
              (f"   p=0.1: fwd {o['fwd_us_dropout']:7.1f} -> {n['fwd_us_dropout']:7.1f}  bwd {o['bwd_us_dropout']:7.1f} -> {n['bwd_us_dropout']:7.1f}" if 'fwd_us_dropout' in n else ""))
PY
