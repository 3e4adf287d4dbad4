"""Summarise a `rocprofv3 --kernel-trace --stats` run of bench.py into per-step kernel time.

    python tests/prof_summary.py <dir with *_kernel_trace.csv> [out.txt]

One steady-state training step is the window between the last optimizer kernel (adamw_kernel) of one step and of the
next; the second-to-last complete window of the run is reported (the timed steps; the last ones belong to the
forward-only / roofline legs of bench.py)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void\s+", "", n)
    n = n.split("(")[0]
    depth, out = 0, []
    for ch in n:                       # drop template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif depth == 0:
            out.append(ch)
    return "".join(out).strip()[:48]


def main():
    d = sys.argv[1]
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
                rows.append((int(r[st]), int(r[en]), short(r[kn])))
    rows.sort()
    ends = []            # end time of the last adamw kernel of every optimizer phase
    last_adamw = None
    for s, e, n in rows:
        if n.startswith("adamw_kernel"):
            last_adamw = e
        elif last_adamw is not None and not n.startswith(("sumsq", "adamw")):
            ends.append(last_adamw)
            last_adamw = None
    if last_adamw is not None:
        ends.append(last_adamw)
    assert len(ends) >= 4, f"only {len(ends)} optimizer phases in the trace"
    # windows between consecutive optimizer phases; pick the median-length one among the last 4 full training steps
    wins = [(ends[i], ends[i + 1]) for i in range(len(ends) - 1)]
    cand = sorted(wins[-5:-1] if len(wins) >= 6 else wins, key=lambda w: w[1] - w[0])
    w0, w1 = cand[len(cand) // 2]
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        if s >= w0 and e <= w1:
            agg[n][0] += 1
            agg[n][1] += e - s
    busy = sum(v[1] for v in agg.values())
    lines = [f"one steady-state training step: wall {(w1 - w0) / 1e6:.1f} ms, GPU busy {busy / 1e6:.1f} ms, "
             f"{sum(v[0] for v in agg.values())} kernel launches",
             f"{'kernel':<50}{'calls':>6}{'ms':>10}{'%':>7}{'avg us':>10}"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{n:<50}{c:>6}{t / 1e6:>10.2f}{100.0 * t / busy:>7.1f}{t / c / 1e3:>10.1f}")
    # where the GPU idles inside the window: gaps between the end of one kernel and the start of the next
    win = [(s, e, n) for s, e, n in rows if s >= w0 and e <= w1]
    gaps, by_prev, t_end, prev = [], defaultdict(lambda: [0, 0]), None, None
    for s, e, n in win:
        if t_end is not None and s > t_end:
            gaps.append((s - t_end, prev, n, (t_end - w0) / 1e6))
            by_prev[prev + " -> " + n][0] += 1
            by_prev[prev + " -> " + n][1] += s - t_end
        if t_end is None or e > t_end:
            t_end, prev = e, n
    idle = sum(g[0] for g in gaps)
    lines.append("")
    lines.append(f"idle inside the window: {idle / 1e6:.2f} ms in {len(gaps)} gaps (median {sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3:.1f} us); "
                 f"gaps > 20 us: {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms in {sum(1 for g in gaps if g[0] > 20000)}")
    lines.append("largest gaps (us, at ms into the step, after kernel -> before kernel):")
    for g in sorted(gaps, reverse=True)[:15]:
        lines.append(f"  {g[0] / 1e3:>8.1f}  @{g[3]:>7.2f}  {g[1]} -> {g[2]}")
    lines.append("idle by kernel pair (ms, count):")
    for k, (c, t) in sorted(by_prev.items(), key=lambda kv: -kv[1][1])[:15]:
        lines.append(f"  {t / 1e6:>7.2f} {c:>5}  {k}")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
