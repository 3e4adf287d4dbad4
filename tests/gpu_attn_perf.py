"""Attention micro-benchmarks at the DreamVLA shapes (GPU box only, not a test): forward and backward of the trunk (real
generate_attention_mask, L = 651 / 930), the ViT (L = 197), the dream-head decoders (L = 205 / 265), with the separate
backward kernels timed through rocprof-free HIP events.  Prints JSON lines; writes gpurun_out/attn_perf.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402
from dreamvla_amd.dreamvla_model import generate_attention_mask  # noqa: E402
from tests.gpu_perf import timeit  # noqa: E402

BF = torch.bfloat16


def main():
    torch.manual_seed(0)
    res = []
    for (B, H, L, mk) in [(32, 16, 651, "trunk"), (32, 16, 651, "dense"), (448, 12, 197, "dense"), (448, 16, 205, "dense"),
                          (448, 16, 265, "dense"), (64, 16, 930, "trunk")]:
        qkv = torch.randn(B, L, 3 * H * 64, device="cuda", dtype=BF)
        v5 = qkv.view(B, L, 3, H, 64)
        mt, vis = None, 1.0
        if mk == "trunk":
            mask = generate_attention_mask(L // 93, 36, 57, 0, False, False, False, 0.0, 54, 3)
            mt = ops.build_mask_tables(mask, device="cuda")
            vis = float((mask == 0).float().mean())
        fl = 4.0 * B * H * L * L * 64 * vis
        f = lambda: ops.attn_fwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], scale=0.125, mask_tables=mt)
        tf = timeit(f, iters=10)
        o, lse = f()
        do = torch.randn_like(o)
        dqkv = torch.zeros_like(qkv)
        d5 = dqkv.view(B, L, 3, H, 64)
        g = lambda: ops.attn_bwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, lse, do, d5[:, :, 0], d5[:, :, 1], d5[:, :, 2],
                                     scale=0.125, mask_tables=mt)
        tb = timeit(g, iters=10)
        r = {"B": B, "H": H, "L": L, "mask": mk, "visible": round(vis, 3), "fwd_us": tf * 1e6, "bwd_us": tb * 1e6,
             "fwd_TF_visible": fl / tf / 1e12, "bwd_TF_visible": 2.5 * fl / tb / 1e12}
        el = B * H * L * 64 * 2                                   # bytes of one (B, L, H, 64) bf16 operand
        r["fwd_TBps_algorithmic"] = 4 * el / tf / 1e12            # q, k, v read + o written
        r["bwd_TBps_algorithmic"] = 8 * el / tb / 1e12            # q, k, v, o, do read + dq, dk, dv written
        if mk == "dense" and L <= 288:
            # same-process A/B against the ring kernels (round 3's path for these shapes): the switches are read per launch
            os.environ["DVLA_ATTN_SHORT_FWD"] = "0"
            os.environ["DVLA_ATTN_SHORT"] = "0"
            r["fwd_us_ring_kernels"] = timeit(f, iters=10) * 1e6
            r["bwd_us_ring_kernels"] = timeit(g, iters=10) * 1e6
            del os.environ["DVLA_ATTN_SHORT_FWD"], os.environ["DVLA_ATTN_SHORT"]
        if mk == "trunk":
            fd = lambda: ops.attn_fwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], scale=0.125, mask_tables=mt, dropout_p=0.1, seed=(3, 4))
            gd = lambda: ops.attn_bwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, lse, do, d5[:, :, 0], d5[:, :, 1], d5[:, :, 2],
                                          scale=0.125, mask_tables=mt, dropout_p=0.1, seed=(3, 4))
            r["fwd_us_dropout"] = timeit(fd, iters=10) * 1e6
            r["bwd_us_dropout"] = timeit(gd, iters=10) * 1e6
        res.append(r)
        print(json.dumps(r), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "attn_perf.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
