"""Few-rows GEMM kernel (csrc/gemm_skinny.h) at the DiT head's evaluation shapes, per launch, under hipGraph replay (the way the
rollout engine runs them): configurations DVLA_SKINNY_CFG = 1 (4 waves x 12 steps), 2 (8 x 12), 3 (8 x 6), the on-the-fly
LayerNorm variant, and the register-staged kernel (configuration 2) they replaced.  GPU box only, not a test.
Prints JSON lines; writes gpurun_out/skinny_perf.jsonl."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402

BF = torch.bfloat16


def graph_time(fn, n=200):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


def main():
    torch.manual_seed(0)
    ops.GemmTuner.enabled = False
    out = []
    for (M, N, K, res, act) in [(120, 2304, 768, False, 0), (120, 768, 768, True, 0), (120, 3072, 768, False, 2), (120, 768, 3072, True, 0),
                                (394, 2304, 768, False, 0), (394, 768, 3072, True, 0), (77, 2048, 512, False, 5), (60, 768, 1024, False, 0)]:
        a = torch.randn(M, K, device="cuda", dtype=BF)
        # a fresh weight per launch of the graph would be the honest model of a 12-layer network: rotate over 12 weights
        ws = [torch.randn(N, K, device="cuda", dtype=BF) / K ** 0.5 for _ in range(12)]
        b = torch.randn(N, device="cuda", dtype=BF)
        r = torch.randn(M, N, device="cuda", dtype=BF) if res else None
        row = {"M": M, "N": N, "K": K, "residual": res, "act": act}
        state = {"i": 0}

        def call(variant, ln=None):
            w = ws[state["i"] % 12]
            state["i"] += 1
            return ops.gemm(a, w, bias=b, act=act, residual=r, variant=variant, a_ln_eps=ln)
        for cfg in (1, 2, 3, 4):
            os.environ["DVLA_SKINNY_CFG"] = str(cfg)
            row[f"skinny_cfg{cfg}_us"] = graph_time(lambda: call(11))
        os.environ["DVLA_SKINNY_CFG"] = "0"
        row["skinny_default_us"] = graph_time(lambda: call(11))
        if 512 <= K <= 1536:
            row["skinny_layernorm_us"] = graph_time(lambda: call(None, 1e-6))
        row["register_staged_us"] = graph_time(lambda: call(2))
        out.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "skinny_perf.jsonl"), "w") as f:
        for row in out:
            f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
