"""GEMM ablation timings (GPU box only, not a test; ablated builds give wrong results by design).
Register-staged 128x128 kernel: DVLA_GEMM_VARIANT 11 = no loads / LDS writes, 12 = no MFMAs, 14 = no barriers,
15 = no loads + no barriers, 16 = no MFMA + no barriers.
Ring 256x256 kernel: 30 + m (ping-pong loop) / 40 + m (plain loop), m bits: 1 = no LDS-DMA after the prologue,
2 = no MFMA, 4 = no fragment reads."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dreamvla_amd import ops, _lib
from tests.gpu_perf import timeit
lib = _lib.load()
out = []
for (M, N, K, tt, sk) in [(20832, 4096, 1024, False, 1), (1024, 4096, 20832, True, 4), (8192, 8192, 8192, False, 1)]:
    a = torch.randn((K, M) if tt else (M, K), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((K, N) if tt else (N, K), device="cuda", dtype=torch.bfloat16) * 0.03
    kw = dict(a_trans=tt, b_trans=tt, split_k=sk)
    if tt:
        kw["out_dtype"] = torch.float32
    for v in [2, 11, 12, 14] + list(range(30, 38)) + list(range(40, 48)):
        lib.dvla_set_gemm_variant(v)
        t = timeit(lambda: ops.gemm(a, b, **kw), iters=10, warmup=2)
        r = {"shape": [M, N, K], "tt": tt, "variant": v, "us": round(t * 1e6, 1), "TFLOPs_equiv": round(2 * M * N * K / t / 1e12, 1)}
        out.append(r)
        print(json.dumps(r), flush=True)
    lib.dvla_set_gemm_variant(0)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_ablate.json"), "w"), indent=1)
