"""Run every kernel parity check without stopping at failures and dump the metrics (GPU box only).
usage: python tests/gpu_diag.py [out.json]"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_checks  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "gpurun_out", "diag.json")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    results, nfail = [], 0
    t0 = time.time()
    from tests import model_checks
    extra = []
    if "--models" in sys.argv or "--only-models" in sys.argv:
        extra = [(model_checks.hip_module_checks, {}), (model_checks.hip_full_model_checks, {"name": "A"}),
                 (model_checks.hip_full_model_checks, {"name": "B"}), (model_checks.hip_grad_checks, {})]
    base = [] if "--only-models" in sys.argv else gpu_checks.all_checks()
    for fn, kw in base + extra:
        try:
            ms = fn(**kw)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            ms = [{"name": f"{fn.__name__} {kw}", "ok": False, "error": repr(e), "trace": traceback.format_exc()[-1500:]}]
        for m in ms:
            results.append(m)
            if not m["ok"]:
                nfail += 1
            print(("ok   " if m["ok"] else "FAIL ") + m["name"] + "  rel_l2=%.3g max_abs=%.3g" % (m.get("rel_l2", -1), m.get("max_abs", -1))
                  + (("  " + m.get("error", "")) if "error" in m else ""), flush=True)
    with open(out_path, "w") as f:
        json.dump({"n": len(results), "failed": nfail, "seconds": time.time() - t0, "results": results}, f, indent=1)
    print(f"{len(results)} metrics, {nfail} failed, {time.time() - t0:.1f}s -> {out_path}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
