"""GEMM ablation timings (GPU box only, not a test): DVLA_GEMM_VARIANT 0 = real kernel; 11 = no loads/LDS writes,
12 = no MFMAs, 14 = no barriers, 15 = no loads + no barriers (MFMA + ds_read only), 16 = no MFMA + no barriers."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dreamvla_amd import ops, _lib
from tests.gpu_perf import timeit
lib = _lib.load()
for (M, N, K) in [(20832, 1024, 4096), (20832, 4096, 1024), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.03
    for v in (0, 11, 12, 14, 15, 16):
        lib.dvla_set_gemm_variant(v)
        t = timeit(lambda: ops.gemm(a, b))
        print(json.dumps({"shape": [M, N, K], "variant": v, "us": t * 1e6, "TFLOPs_equiv": 2 * M * N * K / t / 1e12}), flush=True)
    lib.dvla_set_gemm_variant(0)
