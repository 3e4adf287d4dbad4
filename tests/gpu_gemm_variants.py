"""GEMM kernel-configuration sweep at the DreamVLA shapes (GPU box only, not a test).  For every (shape, epilogue)
case times the automatic choice (variant 0) and each forced configuration: 2 = register-staged 128x128, 4 / 5 / 6 / 7 =
LDS-DMA ring 256x256 / 256x128 (8 waves) / 128x128 / 256x128 (4 waves, 2 workgroups per CU).  A forced configuration
that does not apply to a case falls back to variant 2, so equal times mean "not applicable".
Writes gpurun_out/gemm_variants.json; the table is kept in profiles/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import _lib, ops  # noqa: E402
from tests.gpu_perf import timeit  # noqa: E402

BF = torch.bfloat16
VARIANTS = (0, 2, 4, 5, 6, 7)


def main():
    lib = _lib.load()
    ops.GemmTuner.enabled = False   # the variants are forced by hand here
    torch.manual_seed(0)
    out = []
    cases = []
    # (label, M, N, K, a_trans, b_trans, epilogue, split_k)
    for (M, N, K) in [(20832, 1024, 1024), (20832, 3072, 1024), (20832, 4096, 1024), (20832, 1024, 4096),
                      (88256, 2304, 768), (88256, 3072, 768), (88256, 768, 3072), (88256, 768, 768),
                      (91840, 4096, 1024), (8192, 8192, 8192)]:
        cases.append(("NT plain", M, N, K, False, False, "none", 1))
    cases += [("NT bias+gelu_erf", 88256, 3072, 768, False, False, "gelu_erf", 1),
              ("NT bias+gelu_tanh+preact", 20832, 4096, 1024, False, False, "gelu_tanh_preact", 1),
              ("NT bias+drop+res", 20832, 1024, 4096, False, False, "drop_res", 1),
              ("NT bias+res", 88256, 768, 3072, False, False, "res", 1),
              ("NN plain", 20832, 1024, 4096, False, True, "none", 1),
              ("NN plain", 20832, 3072, 1024, False, True, "none", 1),
              ("NN dact", 20832, 4096, 1024, False, True, "dact", 1),
              ("NN dact", 91840, 4096, 1024, False, True, "dact", 1),
              ("TT dW", 1024, 4096, 20832, True, True, "f32", 4),
              ("TT dW", 4096, 1024, 20832, True, True, "f32", 4),
              ("TT dW", 1024, 3072, 20832, True, True, "f32", 6),
              ("TT dW", 1024, 1024, 20832, True, True, "f32", 10)]
    for (label, M, N, K, at, bt, epi, sk) in cases:
        a = torch.randn((K, M) if at else (M, K), device="cuda", dtype=BF)
        b = torch.randn((K, N) if bt else (N, K), device="cuda", dtype=BF) * 0.03
        kw = dict(a_trans=at, b_trans=bt, split_k=sk)
        if epi in ("gelu_erf", "gelu_tanh_preact", "drop_res", "res"):
            kw["bias"] = torch.randn(N, device="cuda", dtype=BF)
        if epi == "gelu_erf":
            kw["act"] = 1
        if epi == "gelu_tanh_preact":
            kw["act"] = 2
            kw["want_preact"] = True
        if epi in ("drop_res", "res"):
            kw["residual"] = torch.randn(M, N, device="cuda", dtype=BF)
        if epi == "drop_res":
            kw["dropout_p"] = 0.1
            kw["seed"] = (1, 2)
        if epi == "dact":
            kw["dact_aux"] = torch.randn(M, N, device="cuda", dtype=BF)
            kw["dact"] = 2
        if epi == "f32":
            kw["out_dtype"] = torch.float32
        row = {"case": label, "M": M, "N": N, "K": K, "split_k": sk}
        for v in VARIANTS:
            lib.dvla_set_gemm_variant(v)
            t = timeit(lambda: ops.gemm(a, b, **kw), iters=10, warmup=2)
            row[f"v{v}_us"] = round(t * 1e6, 1)
            row[f"v{v}_TF"] = round(2 * M * N * K / t / 1e12, 1)
        lib.dvla_set_gemm_variant(0)
        out.append(row)
        print(json.dumps(row), flush=True)
        del a, b, kw
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gemm_variants.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
