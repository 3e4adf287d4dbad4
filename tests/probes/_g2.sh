V=8,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56
./build/gemm_probe --cases nn --variants $V --check-only --full-check > gpurun_out/r05b_check.txt 2>&1
grep -c '"ok": true' gpurun_out/r05b_check.txt; grep '"ok": false' gpurun_out/r05b_check.txt | head -20
./build/gemm_probe --cases nn --variants $V --no-check --iters 5 --rounds 5 > gpurun_out/r05b_time.txt 2>&1
cat gpurun_out/r05b_time.txt
DVLA_STAMPS_ALL=1 ./build/gemm_probe --stamps 1024 --stamp-variants 89,57,58,59,60,61,62,63,64 > gpurun_out/r05b_stamps_all.txt 2>&1
./build/gemm_probe --stamps 1024 --stamp-variants 89,57,58,59,60,61,62,63,64 > gpurun_out/r05b_stamps.txt 2>&1
