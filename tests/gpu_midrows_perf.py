"""The trunk's GEMMs at the evaluation engine's row count (one episode: S x 93 = 930 token rows), per launch under hipGraph
replay, for every kernel configuration and split-K the dispatcher has: what does the best available configuration leave on the
table at M ~ 1000?  (Conv1D weights: B is (K, N), n-contiguous; the k-contiguous copy is timed too.)  GPU box only, not a test.
Prints JSON lines; writes gpurun_out/midrows_perf.jsonl."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402
from tests.gpu_skinny_perf import graph_time  # noqa: E402

BF = torch.bfloat16


def main():
    torch.manual_seed(0)
    ops.GemmTuner.enabled = False
    out = []
    M = int(os.environ.get("MIDROWS_M", "930"))
    for (N, K, res, act, name) in [(3072, 1024, False, 0, "c_attn"), (1024, 1024, True, 0, "c_proj"), (4096, 1024, False, 2, "c_fc"),
                                   (1024, 4096, True, 0, "mlp.c_proj")]:
        a = torch.randn(M, K, device="cuda", dtype=BF)
        b = torch.randn(N, device="cuda", dtype=BF)
        r = torch.randn(M, N, device="cuda", dtype=BF) if res else None
        for b_trans in (True, False):
            ws = [(torch.randn(K, N, device="cuda", dtype=BF) if b_trans else torch.randn(N, K, device="cuda", dtype=BF)) / K ** 0.5
                  for _ in range(24)]
            row = {"name": name, "M": M, "N": N, "K": K, "b_layout": "(K, N) Conv1D" if b_trans else "(N, K)", "GFLOP": 2e-9 * M * N * K}
            state = {"i": 0}
            best = None
            for variant in (0, 2, 4, 6, 7, 8, 9, 10):      # (12 in profiles/r04_midrows_perf.jsonl: a BK-64 four-stage 128 x 128 ring configuration, measured and removed)
                for sk in (1, 2, 4):
                    def call():
                        w = ws[state["i"] % 24]
                        state["i"] += 1
                        return ops.gemm(a, w, b_trans=b_trans, bias=b, act=act, residual=r, variant=variant, split_k=sk)
                    try:
                        us = graph_time(call, n=96)
                    except Exception as e:      # configuration does not take this problem
                        row[f"v{variant}_sk{sk}"] = None
                        continue
                    row[f"v{variant}_sk{sk}"] = round(us, 2)
                    if best is None or us < best[0]:
                        best = (us, variant, sk)
            row["best"] = {"us": round(best[0], 2), "variant": best[1], "split_k": best[2], "TFLOPs": round(row["GFLOP"] / best[0] * 1e-3, 1)}
            out.append(row)
            print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "midrows_perf.jsonl"), "w") as f:
        for row in out:
            f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
