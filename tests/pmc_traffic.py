"""Turn the two PMC passes of tests/profile_round.sh (FETCH_SIZE, WRITE_SIZE over one training step) into
profiles/<tag>_pmc_traffic.json -- the `roofline.traffic` source of bench.py -- stamped with the commit it was measured on.

    python tests/pmc_traffic.py gpurun_out/r02_pmc_FETCH_SIZE.json gpurun_out/r02_pmc_WRITE_SIZE.json gpurun_out/gemm_breakdown.json profiles/r02_pmc_traffic.json
"""
import json
import subprocess
import sys


def gemm_rows(d):
    return {k: v for k, v in d.items() if "gemm" in k and "splitk" not in k}


def main():
    fetch, write, breakdown, out = sys.argv[1:5]
    F, W = gemm_rows(json.load(open(fetch))), gemm_rows(json.load(open(write)))
    launches = sum(v["launches"] for v in F.values())
    fbytes = sum(v["fetch_bytes_per_launch"] * v["launches"] for v in F.values())
    wbytes = sum(v["write_bytes_per_launch"] * v["launches"] for v in W.values())
    wl = sum(v["launches"] for v in W.values())
    rows = json.load(open(breakdown))
    n = sum(r["launches"] for r in rows)
    alg = sum(r["launches"] * 2.0 * (r["M"] * r["K"] + r["N"] * r["K"] + r["M"] * r["N"] * (2.0 if r["epilogue"] == "fused" else 1.0) / 1.0)
              for r in rows) / n
    # (fused launches write or read a second M x N tensor: pre-activation / act' operand / residual -- counted once more)
    import os
    commit = os.environ.get("DVLA_COMMIT") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()   # (the GPU box has no .git: the launching script passes the commit)
    res = {"gemm_bytes_per_launch": fbytes / launches + wbytes / wl,
           "fetch_bytes_per_launch_corrected": fbytes / launches, "write_bytes_per_launch": wbytes / wl,
           "launches_in_pass": launches, "algorithmic_bytes_per_launch": alg, "commit": commit,
           "per_kernel": {k: {"launches": v["launches"], "fetch_bytes_per_launch": v["fetch_bytes_per_launch"],
                              "write_bytes_per_launch": W.get(k, {}).get("write_bytes_per_launch")} for k, v in F.items()},
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 0 "
                     "--plan <the tuner's locked choices, saved by an ordinary run of the same tree> --no-roofline --no-fwd`: two TUNED "
                     "training steps, no tuner trials, all hand-written GEMM dispatches (tests/profile_round.sh); FETCH_SIZE (KiB) doubled as "
                     "MI355X_MICROARCH.md prescribes for 16-B/lane reads on gfx950, WRITE_SIZE uncalibrated; the counters sit on the L2's "
                     "fabric side and include Infinity-Cache hits"}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("gemm_bytes_per_launch", "algorithmic_bytes_per_launch", "commit")}))


if __name__ == "__main__":
    main()
