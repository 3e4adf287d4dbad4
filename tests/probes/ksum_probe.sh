#!/bin/bash
# what the fused k-sums (bias gradient on the weight-gradient GEMM) cost per launch: the dW problems of the model with and without
# DVLA_PROBE_KSUM, every configuration.   bash tests/probes/ksum_probe.sh   (on the GPU box, after build_probes.sh)
for ks in none b a; do
  echo "== ksum $ks"
  DVLA_PROBE_KSUM=$ks timeout 200 build/gemm_probe --cases dw --no-check --variants 4,7,8 --iters 20 --rounds 5 2>&1 | grep "\"time\"" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['time'], r['M'], r['N'], r['K'], 'sk', r['split_k'], ' '.join('v%s %.1f' % (k[1:], r[k]['us']) for k in r if k[0]=='v' and k[1:].isdigit()))"
done
