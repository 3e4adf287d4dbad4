V=8,50,65,66,67,68,69,70,71,72,73,74
./build/gemm_probe --cases nn --variants $V --check-only --full-check > gpurun_out/r05c_check.txt 2>&1
grep -c '"ok": true' gpurun_out/r05c_check.txt; grep '"ok": false' gpurun_out/r05c_check.txt | head -20
./build/gemm_probe --cases nn --variants $V --no-check --iters 5 --rounds 5 > gpurun_out/r05c_time.txt 2>&1
cat gpurun_out/r05c_time.txt
DVLA_STAMPS_ALL=1 ./build/gemm_probe --stamps 1024 --stamp-variants 75,76,77,78 > gpurun_out/r05c_stamps_all.txt 2>&1
./build/gemm_probe --stamps 1024 --stamp-variants 75,76,77,78 > gpurun_out/r05c_stamps.txt 2>&1
