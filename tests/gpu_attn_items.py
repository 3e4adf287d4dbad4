import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from dreamvla_amd import ops
from dreamvla_amd.dreamvla_model import generate_attention_mask
from tests.gpu_perf import timeit
BF = torch.bfloat16
torch.manual_seed(0)
for (B, H, L) in [(32, 16, 651), (64, 16, 930)]:
    qkv = torch.randn(B, L, 3 * H * 64, device="cuda", dtype=BF)
    v5 = qkv.view(B, L, 3, H, 64)
    mask = generate_attention_mask(L // 93, 36, 57, 0, False, False, False, 0.0, 54, 3)
    mt = ops.build_mask_tables(mask, device="cuda")
    f = lambda: ops.attn_fwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], scale=0.125, mask_tables=mt)
    print(json.dumps({"B": B, "L": L, "items_env": os.environ.get("DVLA_ATTN_ITEMS", "auto"), "fwd_us": timeit(f, iters=20) * 1e6}), flush=True)
