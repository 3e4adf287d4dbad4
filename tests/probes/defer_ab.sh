#!/bin/bash
# Round 6: the phase kernel's deferred epilogue and buffer-descriptor DMA path (gemm_phase.h DBG & 16384 / 32768) against the product
# kernel, same process / same box; NN layout, plain epilogue (the measurement builds of csrc/gemm_inst/phase_NN_e0_exp6.hip):
#   8 = product (round-5 boundary)        44 = product boundary on the buffer-descriptor DMA path
#   47 = deferred epilogue                45 = 47 with the fragment addresses hoisted        46 = 45 with the in-loop epilogue at s_setprio 2
# and the s_memtime timeline of 89 (product) / 88 (deferred).  Every command under `timeout`.
# Result (profiles/r06_gemm_defer_ab.txt, r06_gemm_defer_stamps.txt): measured, NOT adopted -- DESIGN.md section 4.1 "Round 6".
mkdir -p gpurun_out
O=gpurun_out/r06_defer
timeout 300 build/gemm_probe --cases nn --variants 8,44,45,46,47 --iters 20 --rounds 5 > $O.nn.txt 2>&1; echo "rc=$?" >> $O.nn.txt
timeout 100 build/gemm_probe --stamps 1024 --stamp-variants 89,88 > $O.stamps.txt 2>&1
DVLA_STAMPS_ALL=1 timeout 100 build/gemm_probe --stamps 1024 --stamp-variants 88 > $O.stamps_all.txt 2>&1
grep -c '"ok": true' $O.nn.txt; grep '"ok": false' $O.nn.txt | head; grep '"time"' $O.nn.txt | cut -c1-400
