bash tests/probes/pmc_phase.sh r05a 8
./build/gemm_probe --cases pmc --variants 8,9 --no-check --iters 5 --rounds 3 > gpurun_out/r05a_probe_pmc_cases.txt 2>&1
DVLA_STAMPS_ALL=1 ./build/gemm_probe --stamps 1024 --stamp-variants 89 > gpurun_out/r05a_stamps_k1024.txt 2>&1
./build/gemm_probe --cases model --variants 8,9 --no-check --iters 5 --rounds 3 > gpurun_out/r05a_probe_model.txt 2>&1
tail -n 8 gpurun_out/r05a_probe_pmc_cases.txt
