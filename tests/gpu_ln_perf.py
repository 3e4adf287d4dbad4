"""LayerNorm kernel timing (GPU box, measurement only): forward / backward (with the residual-gradient add and parameter
gradients) at the model's shapes and at working sets beyond the Infinity Cache; algorithmic bytes / time.
    python tests/gpu_ln_perf.py > gpurun_out/ln_perf.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402

BF = torch.bfloat16


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    torch.manual_seed(0)
    for rows, cols in [(20832, 1024), (88256, 768), (91840, 1024), (166656, 1024), (353024, 768)]:
        x = torch.randn(rows, cols, device="cuda", dtype=BF)
        w = torch.ones(cols, device="cuda", dtype=BF)
        b = torch.zeros(cols, device="cuda", dtype=BF)
        dy = torch.randn(rows, cols, device="cuda", dtype=BF)
        y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5, True)
        tf = timed(lambda: ops.layernorm_fwd(x, w, b, 1e-5, True))
        tb = timed(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, True, dres2=dy, grad_dtype=BF))
        n = rows * cols * 2
        print(json.dumps({"rows": rows, "cols": cols, "fwd_us": tf, "fwd_TBps": 2 * n / tf / 1e6, "bwd_us": tb,
                          "bwd_TBps_algorithmic_4_streams": 4 * n / tb / 1e6}), flush=True)
        del x, dy, y


if __name__ == "__main__":
    main()
