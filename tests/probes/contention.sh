#!/bin/bash
# Round 6 (round-5 VERDICT #7): profiles/r02_gemm_cu_contention.txt's experiment on TODAY's kernels and plan.
#  (a) gemm_probe, the model's three big plain problems, variants 8 / 9 / 10 / 7, next to H = 0 / 16 / 32 LDS-hogging workgroups on another
#      stream (what an overlapped RCCL kernel does to a persistent launch), under the default schedule (k = 1, stream-K on) and the
#      robust one `GradBucketReducer` switches to while collectives are outstanding (dvla_set_gemm_schedule(8, 0): k = 8, stream-K off);
#  (b) the whole training step with the robust schedule FORCED for the entire step (an upper bound: the reducer holds it from the
#      first bucket launch to the last wait, ~3/4 of backward) against the default step -- tuned separately, as the reducer tunes them.
mkdir -p gpurun_out; O=gpurun_out/r06_contention.txt; : > $O
for k in 1 8; do
  sk=1; [ $k = 8 ] && sk=0
  for H in 0 16 32; do
    echo "## k=$k stream_k=$sk H=$H" >> $O
    DVLA_GEMM_OVERSUBSCRIBE=$k DVLA_GEMM_STREAMK=$sk timeout 300 build/gemm_probe --cases plain --hog $H --no-check --variants 8,9,10,7 --iters 10 --rounds 3 2>&1 \
      | grep '"time"' | head -3 | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('  %dx%dx%-5d ' % (r['M'], r['N'], r['K']) + '  '.join('%s %7.1f' % (k, v['us']) for k, v in r.items() if k.startswith('v') and isinstance(v, dict)))" >> $O
  done
done
B="python bench.py --no-cpu-baseline --no-eager-baseline --no-rollout --no-loss-parity --no-other-configs --no-integration-levels --no-fwd --steps 10 --warmup 2"
for tag in default robust; do
  if [ $tag = robust ]; then export DVLA_GEMM_OVERSUBSCRIBE=8 DVLA_GEMM_STREAMK=0; fi
  timeout 600 $B 2> /dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('## whole step, $tag schedule: %.2f ms/step, %.1f samples/s, GEMM %.2f ms at %.0f TFLOP/s' % (d['ms_per_step'], d['value'], d['roofline']['gemm_ms_per_step'], d['roofline']['achieved']))" >> $O
done
cat $O
