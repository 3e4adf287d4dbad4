"""Round 6 diagnostics (GPU box only, not a test): per-step wall time of the bench's training step over a long run in ONE process --
plan replay, no tuner trials -- to see whether the 123 / 134 ms modes seen between bench.py processes are a warm-up effect, an
allocator-layout effect (re-run after empty_cache) or a clock effect (sclk / power sampled beside the steps).
    python tests/probes/step_trace.py <plan.json> [n_steps]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)["card0"]
        return {k: v for k, v in d.items() if "sclk" in k or "Power" in k or "mclk" in k}
    except Exception as e:  # noqa: BLE001
        return {"err": repr(e)}


def main():
    import bench
    from dreamvla_amd import losses
    from dreamvla_amd.ddp import GradBucketReducer
    from dreamvla_amd.dreamvla_model import DreamVLA
    from dreamvla_amd.ops import GemmTuner
    from dreamvla_amd.optim import FlatAdamW
    from dreamvla_amd.synthetic import synthetic_batch
    plan = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    S, B = 7, 32
    cfg = bench.model_cfg("C", S, 24, "finetune")
    model = DreamVLA(clip_device="cpu", vit_checkpoint_path=None, **cfg).bfloat16()
    model.clip_model.requires_grad_(False)
    model.vision_encoder.requires_grad_(False)
    model = model.to(dev)
    model._init_model_type()
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    reducer = GradBucketReducer(params, direct_grads=True)
    opt = FlatAdamW(reducer, lr=1e-3, weight_decay=1e-4, max_grad_norm=0.1)
    b = synthetic_batch(B, S, window=S + 3, seed=1234, heads=bench.label_heads("C"))
    b["actions"][..., 6:] = (b["actions"][..., 6:] > 0.5).float()
    batch = {k: (v.to(dev, torch.bfloat16) if torch.is_floating_point(v) else v.to(dev)) for k, v in b.items()}
    lab = losses.label_actions(batch["actions"], S, 3)
    inputs = (batch["image_primary"][:, :S].contiguous(), batch["image_wrist"][:, :S].contiguous(),
              batch["state"][:, :S].contiguous(), batch["text_token"][:, :S].contiguous())
    GemmTuner.load_plan(plan)

    def step():
        reducer.zero_grad()
        out = model(*inputs, action=batch["actions"][:, :S], action_label=lab, mode="train")
        total, _ = losses.calvin_losses(out, batch, sequence_length=S, use_dit_head=cfg["use_dit_head"], label_action=lab)
        total.backward()
        reducer.finish()
        opt.step()

    def run(tag, k, sync_each):
        torch.cuda.synchronize()
        ts = []
        t0 = time.perf_counter()
        mid = None
        for i in range(k):
            t1 = time.perf_counter()
            step()
            if sync_each:
                torch.cuda.synchronize()
                ts.append(round((time.perf_counter() - t1) * 1e3, 1))
            elif i == (2 * k) // 3:
                mid = smi()              # free-running: the GPU is busy while this is read (the reading is a lagging average)
        torch.cuda.synchronize()
        print(json.dumps({"tag": tag, "steps": k, "ms_per_step": round((time.perf_counter() - t0) / k * 1e3, 2), "each": ts,
                          "mem_GB": round(torch.cuda.memory_reserved() / 2**30, 1), "smi_during": mid, "smi_after": smi()}), flush=True)

    mode = os.environ.get("STEP_TRACE_MODE", "")
    if mode == "idle":       # 4 s of idle GPU after initialisation: is the slow phase a count of steps or time since start?
        torch.cuda.synchronize(); time.sleep(4)
    if mode == "load":       # 4 s of plain GEMM load (torch matmul: another library's kernels) before the first step
        a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 4:
            for _ in range(20):
                a @ a
            torch.cuda.synchronize()
        print(json.dumps({"tag": "after 4 s of matmul load", "smi": smi()}), flush=True)
    run("first 10, synced each", 10, True)
    run("next 10, free-running", 10, False)
    run(f"next {n}, synced each", n, True)
    run("30 free-running", 30, False)
    torch.cuda.empty_cache()
    run("after empty_cache: 10 synced", 10, True)
    run("10 free-running", 10, False)
    time.sleep(5)
    run("after 5 s idle: 10 free-running", 10, False)
    run("10 free-running", 10, False)


if __name__ == "__main__":
    main()
