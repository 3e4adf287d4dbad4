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
