"""Aggregate a `rocprofv3 --pmc ... --kernel-trace` output directory per kernel (measurement infrastructure).

    python tests/pmc_summary.py <dir> [out.json] [--by-grid]

--by-grid: one row per (kernel, launch grid) instead of per kernel -- launches of one kernel on different problem sizes are
not averaged together (round-2 VERDICT: the LayerNorm summary mixed a 353024 x 768 and a 166656 x 1024 problem).

Reads every *counter_collection.csv (one row per dispatch and counter) and *kernel_trace.csv (durations) under <dir>; prints
and optionally writes, per kernel (template arguments dropped): launches, average duration, the SUM and the per-launch average
of every counter, plus derived figures where their inputs are present:
    mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)     (MI355X_MICROARCH.md constants)
    fetch_bytes      = FETCH_SIZE [KiB] x 1024 x 2   (gfx950: the counter tallies 128-B requests at 64 B -- the guide's correction)
    write_bytes      = WRITE_SIZE [KiB] x 1024       (uncalibrated, as the guide says)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import short  # noqa: E402


def col(fields, *names):
    for n in names:
        for f in fields:
            if n.lower() == f.lower():
                return f
    for n in names:
        for f in fields:
            if n.lower() in f.lower():
                return f
    return None


def main():
    by_grid = "--by-grid" in sys.argv
    if by_grid:
        sys.argv.remove("--by-grid")
    full = "--full-names" in sys.argv      # keep the template arguments: one row per instantiation (operand layout, epilogue class)
    if full:
        sys.argv.remove("--full-names")
    d = sys.argv[1]

    def key_of(r, kn, gs):
        k = r[kn].split("(")[0].replace("void ", "").strip()[:110] if full else short(r[kn])
        return f"{k} grid={r[gs]}" if (by_grid and gs) else k
    agg = defaultdict(lambda: {"launches": 0, "dur_ns": 0.0, "counters": defaultdict(float), "disp": set()})
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rd = csv.DictReader(fh)
            kn, cn, cv = col(rd.fieldnames, "Kernel_Name"), col(rd.fieldnames, "Counter_Name"), col(rd.fieldnames, "Counter_Value")
            di = col(rd.fieldnames, "Dispatch_Id")
            gs = col(rd.fieldnames, "Grid_Size", "Grid_Size_X")
            for r in rd:
                a = agg[key_of(r, kn, gs)]
                a["counters"][r[cn]] += float(r[cv])
                a["disp"].add(r[di])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rd = csv.DictReader(fh)
            kn, st, en = col(rd.fieldnames, "Kernel_Name"), col(rd.fieldnames, "Start_Timestamp"), col(rd.fieldnames, "End_Timestamp")
            gs = col(rd.fieldnames, "Grid_Size", "Grid_Size_X")
            for r in rd:
                a = agg[key_of(r, kn, gs)]
                a["launches"] += 1
                a["dur_ns"] += float(r[en]) - float(r[st])
    out = {}
    for k, a in agg.items():
        n = max(a["launches"], len(a["disp"]), 1)
        c = dict(a["counters"])
        row = {"launches": n, "avg_us": a["dur_ns"] / n / 1e3 if a["launches"] else None,
               "counters_sum": c, "counters_per_launch": {x: v / n for x, v in c.items()}}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            row["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        if "FETCH_SIZE" in c:
            row["fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024.0 * 2.0 / n
            if a["launches"] and a["dur_ns"] > 0:
                row["fetch_GBps"] = c["FETCH_SIZE"] * 1024.0 * 2.0 / a["dur_ns"]
        if "WRITE_SIZE" in c:
            row["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / n
            if a["launches"] and a["dur_ns"] > 0:
                row["write_GBps"] = c["WRITE_SIZE"] * 1024.0 / a["dur_ns"]
        out[k] = row
    rows = sorted(out.items(), key=lambda kv: -(kv[1]["avg_us"] or 0) * kv[1]["launches"])
    for k, r in rows[:30]:
        extra = " ".join(f"{x}={r[x]:.4g}" for x in ("mfma_busy_frac", "fetch_bytes_per_launch", "write_bytes_per_launch", "fetch_GBps", "write_GBps") if x in r)
        if full:
            extra += " " + " ".join(f"{x}={v:.4g}" for x, v in sorted(r["counters_per_launch"].items()))
        print(f"{k[:110 if full else 70]:{110 if full else 70}s} x{r['launches']:5d} avg {r['avg_us'] or 0:9.1f} us  {extra}")
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
