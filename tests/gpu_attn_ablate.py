"""Forward attention ring kernel under its ablation builds (env DVLA_ATTN_DBG, read per launch; results garbage by design).
Needs a MEASUREMENT build of the library: DVLA_ABLATIONS=1 python -c 'import __graft_entry__ as g; g.build(force=True)' (the product
library does not contain the ablation kernels):
which part of the tile loop paces it?  GPU box only, not a test.  Prints one line per shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402
from dreamvla_amd.dreamvla_model import generate_attention_mask  # noqa: E402
from tests.gpu_perf import timeit  # noqa: E402

NAMES = {0: "full", 1: "no barrier/wait", 2: "no softmax", 4: "no P.V", 8: "no K.Q", 16: "no DMA", 14: "no math at all",
         15: "no math, no sync", 31: "loop skeleton", 63: "skeleton without Q / O traffic", 64: "empty kernel"}


def main():
    torch.manual_seed(0)
    for (B, H, L, mk) in [(448, 16, 205, "dense"), (32, 16, 651, "dense"), (32, 16, 651, "trunk")]:
        qkv = torch.randn(B, L, 3 * H * 64, device="cuda", dtype=torch.bfloat16)
        v5 = qkv.view(B, L, 3, H, 64)
        mt = None
        if mk == "trunk":
            mt = ops.build_mask_tables(generate_attention_mask(L // 93, 36, 57, 0, False, False, False, 0.0, 54, 3), device="cuda")
        f = lambda: ops.attn_fwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], scale=0.125, mask_tables=mt)
        out = []
        for dbg in (0, 1, 2, 4, 8, 16, 14, 15, 31, 63, 64):
            os.environ["DVLA_ATTN_DBG"] = str(dbg)
            out.append("%s %.1f" % (NAMES[dbg], timeit(f, iters=10) * 1e6))
        os.environ["DVLA_ATTN_DBG"] = "0"
        print(f"fwd B={B} H={H} L={L} {mk}: " + " | ".join(out) + "  (us)", flush=True)


if __name__ == "__main__":
    main()
