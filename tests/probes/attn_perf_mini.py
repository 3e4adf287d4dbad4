"""Same-box A/B of attention builds (GPU box only, not a test): the trunk (real mask, dropout on) and the dense shapes of the
training step, forward and backward, 40 launches each.  The library is whatever DVLA_LIB names (tests/probes/attn_variants.sh runs
this once per build, twice over, interleaved).  One JSON line: {"lib": ..., "<shape> fwd": us, "<shape> bwd": us, ...}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402
from dreamvla_amd.dreamvla_model import generate_attention_mask  # noqa: E402
from tests.gpu_perf import timeit  # noqa: E402

BF = torch.bfloat16


def main():
    torch.manual_seed(0)
    out = {"lib": os.path.basename(os.environ.get("DVLA_LIB", "libdvla_hip.so"))}
    for (B, H, L, mk) in [(32, 16, 651, "trunk"), (32, 16, 651, "dense"), (448, 12, 197, "dense"), (448, 16, 205, "dense"), (448, 16, 265, "dense")]:
        qkv = torch.randn(B, L, 3 * H * 64, device="cuda", dtype=BF)
        v5 = qkv.view(B, L, 3, H, 64)
        mt, kw = None, {}
        if mk == "trunk":
            mt = ops.build_mask_tables(generate_attention_mask(L // 93, 36, 57, 0, False, False, False, 0.0, 54, 3), device="cuda")
            kw = dict(dropout_p=0.1, seed=(3, 4))
        f = lambda: ops.attn_fwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], scale=0.125, mask_tables=mt, **kw)
        o, lse = f()
        do = torch.randn_like(o)
        d5 = torch.zeros_like(qkv).view(B, L, 3, H, 64)
        g = lambda: ops.attn_bwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, lse, do, d5[:, :, 0], d5[:, :, 1], d5[:, :, 2],
                                     scale=0.125, mask_tables=mt, **kw)
        out[f"{mk}{L} fwd"] = round(timeit(f, iters=40) * 1e6, 1)
        out[f"{mk}{L} bwd"] = round(timeit(g, iters=40) * 1e6, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
