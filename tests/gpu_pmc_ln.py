"""LayerNorm launches on working sets LARGER than the 256-MiB Infinity Cache for a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
pass (GPU box only, not a test): 353024 x 768 (4 x the ViT's 88256 rows: 542 MB in + 542 MB out) and 166656 x 1024 (8 x the
trunk's 20832 rows: 341 MB each way), forward and backward (with the residual-gradient add and parameter gradients)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    torch.manual_seed(0)
    shapes = [(353024, 768), (166656, 1024)]
    if len(sys.argv) > 1:          # one shape per process: the counter summary then has one row per (kernel, shape)
        shapes = [shapes[int(sys.argv[1])]]
    for rows, cols in shapes:
        x = torch.randn(rows, cols, device="cuda", dtype=BF)
        w = torch.ones(cols, device="cuda", dtype=BF)
        b = torch.zeros(cols, device="cuda", dtype=BF)
        dy = torch.randn(rows, cols, device="cuda", dtype=BF)
        for _ in range(4):
            y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5, True)
            ops.layernorm_bwd(dy, x, w, mean, rstd, True, dres2=dy, grad_dtype=BF)
        torch.cuda.synchronize()
        del x, dy, y


if __name__ == "__main__":
    main()
