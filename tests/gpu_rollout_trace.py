"""Kernel-trace view of ONE single-episode control step of the rollout engine (BASELINE configs[4] at B = 1: the reference's
evaluation runs one episode per rank, utils/eval_utils_calvin.py:82-147).

    rocprofv3 --kernel-trace -d gpurun_out/rt -f csv -- python tests/gpu_rollout_trace.py run        (GPU box)
    python tests/gpu_rollout_trace.py summary gpurun_out/rt gpurun_out/rollout_step_summary.txt

`run` drives the hipGraph engine for warm-up + 24 steady steps and prints the wall time per step; `summary` takes the last
`window_ms` of the trace (all steady steps), and reports kernel launches, kernel time and idle time PER STEP, by kernel name."""
import csv
import glob
import json
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    from tests.gpu_rollout_bench import build_model
    from dreamvla_amd.rollout import RolloutEngine
    m, S = build_model()
    B, dev, BF = 1, "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(1)
    frames = [(torch.randn(B, 3, 224, 224, generator=g).to(dev, BF), torch.randn(B, 3, 224, 224, generator=g).to(dev, BF),
               torch.cat([torch.rand(B, 6, generator=g), torch.ones(B, 1)], -1).to(dev, BF)) for _ in range(4)]
    text = torch.randint(1, 49000, (B, 77), generator=g).to(dev)
    eng = RolloutEngine(m, B, use_graph=True, warmup_decodes=6)
    for i in range(S + 8):
        eng.step(*frames[i % 4], text)
    torch.cuda.synchronize()
    n = 24
    t0 = time.perf_counter()
    for i in range(n):
        eng.step(*frames[i % 4], text)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"B": B, "steps": n, "ms_per_step": ms, "graphs": eng.graphs_captured}), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "rollout_trace_run.json"), "w") as f:
        json.dump({"ms_per_step": ms, "steps": n}, f)


def summary(d, out):
    from tests.prof_summary import short
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no *_kernel_trace.csv under " + d
    rows = []
    for f in files:
        with open(f) as fh:
            rd = csv.DictReader(fh)
            kn = next(c for c in rd.fieldnames if "Kernel_Name" in c)
            st = next(c for c in rd.fieldnames if "Start_Timestamp" in c)
            en = next(c for c in rd.fieldnames if "End_Timestamp" in c)
            for r in rd:
                name = r[kn]
                if "gemm_skinny_kernel" in name:       # keep the template arguments: <waves, LayerNorm on the fly>
                    import re
                    m = re.search(r"gemm_skinny_kernel<([^>]*)>", name)
                    rows.append((int(r[st]), int(r[en]), "gemm_skinny_kernel<" + (m.group(1) if m else "?") + ">"))
                else:
                    rows.append((int(r[st]), int(r[en]), short(name)))
    rows.sort()
    with open(os.path.join(os.path.dirname(out), "rollout_trace_run.json")) as f:
        run_info = json.load(f)
    step_ns = run_info["ms_per_step"] * 1e6
    nsteps = 16
    t1 = rows[-1][1]
    t0 = t1 - nsteps * step_ns
    win = [r for r in rows if r[0] >= t0]
    agg = defaultdict(lambda: [0, 0])
    busy, prev_end = 0, None
    for s, e, n in win:
        agg[n][0] += 1
        agg[n][1] += e - s
        busy += e - s
    lines = [f"single-episode rollout step (B = 1, S = 10, DiT + DDIM-10 + CFG, hipGraph encode + decode): {run_info['ms_per_step']:.2f} ms wall "
             f"per step under the tracer; over the last {nsteps} steps: {len(win) / nsteps:.0f} kernel launches and "
             f"{busy / nsteps / 1e6:.2f} ms of kernel time per step ({busy / nsteps / step_ns:.0%} of the step: the rest is launch / dependency latency)",
             f"{'kernel':50s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>8s}"]
    for n, (c, tns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{n:50s} {c / nsteps:10.1f} {tns / nsteps / 1e6:9.3f} {tns / c / 1e3:8.1f}")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summary(sys.argv[2], sys.argv[3])
