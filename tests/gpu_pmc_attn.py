"""A few attention launches for a rocprofv3 --pmc pass (GPU box only, not a test)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    from dreamvla_amd.dreamvla_model import generate_attention_mask
    torch.manual_seed(0)
    # the step's attention mix: the trunk under its real mask with dropout (ring kernels), the ViT and the decoders (short-sequence
    # kernels of round 4), one dense long sequence
    mask = generate_attention_mask(7, 36, 57, 0, False, False, False, 0.0, 54, 3)
    mt = ops.build_mask_tables(mask, device="cuda")
    for (B, H, L, tables, p) in [(32, 16, 651, mt, 0.1), (448, 12, 197, None, 0.0), (448, 16, 205, None, 0.0), (448, 16, 265, None, 0.0),
                                 (32, 16, 651, None, 0.0)]:
        qkv = torch.randn(B, L, 3 * H * 64, device="cuda", dtype=BF, requires_grad=True)
        for _ in range(3):
            o = ops.self_attention(qkv, num_heads=H, mask_tables=tables, dropout_p=p)
            o.backward(torch.ones_like(o))
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
