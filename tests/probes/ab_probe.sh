#!/bin/bash
# Same-box A/B of two builds of libdvla_hip.so under gemm_probe (box-to-box spread is +-3 %: only pairs from one gpurun call mean
# anything).  Put the two libraries at build/ab/base.so and build/ab/new.so (e.g. `git stash; build; cp ...; git stash pop; build;
# cp ...`), build the probes, then on the GPU box:   bash tests/probes/ab_probe.sh
# (gemm_probe finds the library through RUNPATH, which LD_LIBRARY_PATH precedes.)
mkdir -p gpurun_out/abdir/base gpurun_out/abdir/new
cp build/ab/base.so gpurun_out/abdir/base/libdvla_hip.so; cp build/ab/new.so gpurun_out/abdir/new/libdvla_hip.so
for r in 1 2; do for w in base new; do
  echo "== $w round $r"
  LD_LIBRARY_PATH=$PWD/gpurun_out/abdir/$w timeout 200 build/gemm_probe --cases model --no-check --variants 8,9 --iters 20 --rounds 5 2>&1 | grep "\"time\"" | head -13 | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['time'], r['epi'], r['v8']['us'], r['v9']['us'])"
done; done
rm -rf gpurun_out/abdir
