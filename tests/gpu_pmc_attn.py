"""A few attention launches for a rocprofv3 --pmc pass (GPU box only, not a test)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    torch.manual_seed(0)
    for (B, H, L) in [(32, 16, 651), (448, 12, 197), (448, 16, 265)]:
        qkv = torch.randn(B, L, 3 * H * 64, device="cuda", dtype=BF, requires_grad=True)
        for _ in range(3):
            o = ops.self_attention(qkv, num_heads=H)
            o.backward(torch.ones_like(o))
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
