"""A handful of GEMM launches for a rocprofv3 --pmc pass (GPU box only, not a test).  Each case is launched
DVLA_PMC_REPS times (default 3) after one warm-up; DVLA_GEMM_VARIANT selects the kernel configuration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402

BF = torch.bfloat16
REPS = int(os.environ.get("DVLA_PMC_REPS", "3"))


def run(M, N, K, at=False, bt=False, **kw):
    a = torch.randn((K, M) if at else (M, K), device="cuda", dtype=BF)
    b = torch.randn((K, N) if bt else (N, K), device="cuda", dtype=BF) * 0.03
    for _ in range(1 + REPS):
        ops.gemm(a, b, a_trans=at, b_trans=bt, **kw)
    torch.cuda.synchronize()


def main():
    torch.manual_seed(0)
    run(8192, 8192, 8192)
    run(1024, 4096, 20832, at=True, bt=True, split_k=4, out_dtype=torch.float32)
    run(20832, 4096, 1024)
    bias = torch.randn(4096, device="cuda", dtype=BF)
    run(20832, 4096, 1024, bias=bias, act=2, want_preact=True)
    aux = torch.randn(20832, 4096, device="cuda", dtype=BF)
    run(20832, 4096, 1024, bt=True, dact_aux=aux, dact=2)
    run(20832, 1024, 4096)


if __name__ == "__main__":
    main()
