"""Round 6 (GPU box only, not a test): energy per launch of the phase kernel and of its ablation builds (variants 81 no MFMA, 83 no
MFMA + no fragment reads, 84 no DMA, 85 no epilogue -- results are garbage by design) at 8192^3 on N(0, 1) and on zero operands,
~1.5 s of back-to-back launches per point, clock and package power read while the queue runs (idle power of the box printed first).
    python tests/probes/gemm_energy_ablation.py > gpurun_out/r06_gemm_energy_ablation.jsonl"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import _lib, ops  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)["card0"]
        return {"sclk": d.get("sclk clock speed:"), "power_W": float(d.get("Current Socket Graphics Package Power (W)"))}
    except Exception as e:  # noqa: BLE001
        return {"err": repr(e)}


def main():
    lib = _lib.load()
    ops.GemmTuner.enabled = False
    BF = torch.bfloat16
    torch.zeros(1, device="cuda"); torch.cuda.synchronize(); time.sleep(2)
    print(json.dumps({"idle": smi()}), flush=True)
    M = N = K = 8192
    for kind in ("randn", "zeros"):
        a = (torch.randn if kind == "randn" else torch.zeros)(M, K, device="cuda", dtype=BF)
        b = (torch.randn if kind == "randn" else torch.zeros)(N, K, device="cuda", dtype=BF)
        out = torch.empty(M, N, device="cuda", dtype=BF)
        for var in (8, 81, 83, 84, 85):
            lib.dvla_set_gemm_variant(var)
            try:
                for _ in range(20):
                    ops.gemm(a, b, out=out)
                torch.cuda.synchronize()
                n = 1500
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    ops.gemm(a, b, out=out)
                e1.record()
                time.sleep(0.4)
                s = smi()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / n
                row = {"variant": var, "data": kind, "us": round(us, 1), **s}
                if "power_W" in s:
                    row["J_per_launch"] = round(s["power_W"] * us * 1e-6, 3)
                print(json.dumps(row), flush=True)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"variant": var, "data": kind, "err": repr(e)}), flush=True)
            lib.dvla_set_gemm_variant(0)
            time.sleep(0.5)


if __name__ == "__main__":
    main()
