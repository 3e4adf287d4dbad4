"""Round 6 (GPU box only, not a test): the same GEMM launches on operands with different switching activity -- N(0, 1) bf16, all zeros,
all ones, NaN -- with the clock and power rocm-smi reports right behind each run.  (The NaN-model incident of DESIGN section 4.2: the
clocks of this chip follow the data.)    python tests/probes/gemm_data_power.py > gpurun_out/r06_gemm_data_power.jsonl"""
import ctypes as C
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
        return {"sclk": d.get("sclk clock speed:"), "power_W": d.get("Current Socket Graphics Package Power (W)")}
    except Exception as e:  # noqa: BLE001
        return {"err": repr(e)}


def main():
    lib = _lib.load()
    try:
        cmp = _lib.load_comparator()
        ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    except Exception:  # noqa: BLE001
        cmp = ws = None
    ops.GemmTuner.enabled = False
    BF = torch.bfloat16
    for (M, N, K, var) in [(8192, 8192, 8192, 8), (20832, 4096, 1024, 8), (20832, 1024, 4096, 10)]:
        for kind in ("randn", "zeros", "ones", "nan"):
            if kind == "randn":
                a = torch.randn(M, K, device="cuda", dtype=BF); b = torch.randn(N, K, device="cuda", dtype=BF)
            elif kind == "zeros":
                a = torch.zeros(M, K, device="cuda", dtype=BF); b = torch.zeros(N, K, device="cuda", dtype=BF)
            elif kind == "ones":
                a = torch.ones(M, K, device="cuda", dtype=BF); b = torch.ones(N, K, device="cuda", dtype=BF)
            else:
                a = torch.full((M, K), float("nan"), device="cuda", dtype=BF); b = torch.full((N, K), float("nan"), device="cuda", dtype=BF)
            out = torch.empty(M, N, device="cuda", dtype=BF)
            n = max(50, int(2.4e15 / (2.0 * M * N * K)))            # 1.2-1.8 s of launches: long enough for the clocks to settle

            def sustained(fn, tag):
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                s = {}
                for it in range(n):
                    fn()
                    if it == (2 * n) // 3:                          # the reading is a moving average: sample two thirds into the run, from
                        s = smi()                                   # inside the launch loop (a full queue blocks the loop: it runs with the GPU)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / n
                print(json.dumps({"M": M, "N": N, "K": K, "kernel": tag, "data": kind, "us": round(us, 1),
                                  "TFLOPs": round(2.0 * M * N * K / us / 1e6, 1), **s}), flush=True)
                time.sleep(0.5)

            lib.dvla_set_gemm_variant(var)
            sustained(lambda: ops.gemm(a, b, out=out), f"hand-written v{var}")
            lib.dvla_set_gemm_variant(0)
            if cmp is not None and kind in ("randn", "zeros"):     # the vendor library on the same operands, same protocol (yardstick)
                from dreamvla_amd._lib import GemmParams
                q = GemmParams()
                q.A, q.lda, q.a_trans = a.data_ptr(), a.stride(0), 0
                q.B, q.ldb, q.b_trans = b.data_ptr(), b.stride(0), 0
                q.C, q.ldc, q.c_dtype = out.data_ptr(), out.stride(0), 0
                q.M, q.N, q.K, q.split_k = M, N, K, 1
                if cmp.dvla_gemm_library_bf16(C.byref(q), ws.data_ptr(), ws.numel(), None) == 0:
                    sustained(lambda: cmp.dvla_gemm_library_bf16(C.byref(q), ws.data_ptr(), ws.numel(), None), "hipBLASLt")
            del a, b, out


if __name__ == "__main__":
    main()
