"""Round 6 (GPU box only, not a test): weight-gradient GEMMs with SMALL outputs and a mid-length contraction -- the DiT head's
(768 | 2304 | 3072) x (768 | 3072) x 10752 and the resampler's / CLIP-sized ones -- over split-K counts and kernel configurations.
ops.auto_split_k sizes the split for 256 x 256 tiles and slices of >= 1.3 k of K; a 768 x 768 output is 9 such tiles (8 slices =
72 work items on 256 CUs).  Prints one JSON line per (shape, split_k) with us per launch (GEMM + its reduction pass) per configuration.
    python tests/probes/small_dw_sweep.py > gpurun_out/r06_small_dw_sweep.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import _lib, ops  # noqa: E402
from tests.gpu_perf import timeit  # noqa: E402

BF = torch.bfloat16
VARIANTS = (0, 2, 6, 7, 8, 10)


def main():
    lib = _lib.load()
    ops.GemmTuner.enabled = False
    torch.manual_seed(0)
    shapes = [(768, 768, 10752), (768, 3072, 10752), (3072, 768, 10752), (2304, 768, 10752),
              (768, 768, 7168), (768, 1536, 94976), (1024, 768, 7168)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
    for (M, N, K) in shapes:
        a = torch.randn(K, M, device="cuda", dtype=BF)
        b = torch.randn(K, N, device="cuda", dtype=BF) * 0.03
        out = torch.zeros(M, N, device="cuda", dtype=BF)
        cur = ops.auto_split_k(M, N, K)
        for sk in sorted({cur, 1, 2, 4, 6, 7, 8, 10, 12, 14, 16, 21, 24, 28}):
            if K // sk < 256 or (K // 64) < sk:
                continue
            row = {"M": M, "N": N, "K": K, "split_k": sk, "auto_split_k": cur}
            for v in VARIANTS:
                lib.dvla_set_gemm_variant(v)
                try:
                    t = timeit(lambda: ops.gemm(a, b, a_trans=True, b_trans=True, split_k=sk, out=out), iters=10, warmup=2)
                    row[f"v{v}_us"] = round(t * 1e6, 1)
                except Exception as e:  # noqa: BLE001
                    row[f"v{v}_us"] = None
            lib.dvla_set_gemm_variant(0)
            best = min((x for x in (row[f"v{v}_us"] for v in VARIANTS) if x), default=None)
            row["best_us"] = best
            row["best_TF"] = round(2 * M * N * K / best / 1e6, 1) if best else None
            print(json.dumps(row), flush=True)
        del a, b, out


if __name__ == "__main__":
    main()
