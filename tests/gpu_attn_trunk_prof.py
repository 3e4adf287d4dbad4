"""Trunk attention (real generate_attention_mask, B = 32, H = 16, L = 651) under rocprofv3 --kernel-trace: N launches with the
key list in ascending column order, then N with the audience-grouped order ops.build_mask_tables produces.  Not a test.
    rocprofv3 --kernel-trace -d <dir> -f csv -- python tests/gpu_attn_trunk_prof.py ; python tests/gpu_attn_trunk_prof.py --summarise <dir>"""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 6


def run():
    import numpy as np
    import torch
    from dreamvla_amd import ops
    from dreamvla_amd.dreamvla_model import generate_attention_mask
    torch.manual_seed(0)
    B, H, L = 32, 16, 651
    qkv = torch.randn(B, L, 3 * H * 64, device="cuda", dtype=torch.bfloat16)
    v5 = qkv.view(B, L, 3, H, 64)
    mask = generate_attention_mask(L // 93, 36, 57, 0, False, False, False, 0.0, 54, 3)
    grouped = ops.build_mask_tables(mask, device="cuda")
    asc = ops.build_mask_tables(mask, device="cuda", key_order=np.sort(grouped.key_index.cpu().numpy()))
    for drop in (0.0, 0.1):
        for mt in (asc, grouped):
            for _ in range(N):
                o, lse = ops.attn_fwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], scale=0.125, mask_tables=mt, dropout_p=drop, seed=(1, 2))
                do = torch.randn_like(o)
                dqkv = torch.zeros_like(qkv)
                d5 = dqkv.view(B, L, 3, H, 64)
                ops.attn_bwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, lse, do, d5[:, :, 0], d5[:, :, 1], d5[:, :, 2],
                                 scale=0.125, mask_tables=mt, dropout_p=drop, seed=(1, 2))
            torch.cuda.synchronize()


def summarise(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if "attn_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by = {}
    for r in rows:
        by.setdefault(re.search(r"attn_\w+", r["Kernel_Name"]).group(0), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("trunk attention B=32 H=16 L=651, microseconds per launch (median of %d): ascending key list -> audience-grouped key list" % N)
    for k, v in by.items():
        assert len(v) == 4 * N, (k, len(v))
        med = lambda x: sorted(x)[len(x) // 2]
        print("  %-28s no dropout %7.1f -> %7.1f    dropout 0.1 %7.1f -> %7.1f" % (k, med(v[:N]), med(v[N:2 * N]), med(v[2 * N:3 * N]), med(v[3 * N:])))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run()
