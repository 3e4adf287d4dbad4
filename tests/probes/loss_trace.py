"""Round 6 diagnostics (GPU box only, not a test): the bench's training step, plan replay, loss and a finiteness check of the trainable
parameters after every step.  `DVLA_LIB` selects the library build.
    python tests/probes/loss_trace.py <plan.json> [n_steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import bench
    from dreamvla_amd import losses
    from dreamvla_amd.ddp import GradBucketReducer
    from dreamvla_amd.dreamvla_model import DreamVLA
    from dreamvla_amd.ops import GemmTuner
    from dreamvla_amd.optim import FlatAdamW
    from dreamvla_amd.synthetic import synthetic_batch
    plan = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
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
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    reducer = GradBucketReducer(params, direct_grads=True)
    opt = FlatAdamW(reducer, lr=1e-3, weight_decay=1e-4, max_grad_norm=0.1)
    b = synthetic_batch(B, S, window=S + 3, seed=1234, heads=bench.label_heads("C"))
    b["actions"][..., 6:] = (b["actions"][..., 6:] > 0.5).float()
    batch = {k: (v.to(dev, torch.bfloat16) if torch.is_floating_point(v) else v.to(dev)) for k, v in b.items()}
    lab = losses.label_actions(batch["actions"], S, 3)
    inputs = (batch["image_primary"][:, :S].contiguous(), batch["image_wrist"][:, :S].contiguous(),
              batch["state"][:, :S].contiguous(), batch["text_token"][:, :S].contiguous())
    GemmTuner.load_plan(plan)
    for it in range(n):
        t0 = time.perf_counter()
        reducer.zero_grad()
        out = model(*inputs, action=batch["actions"][:, :S], action_label=lab, mode="train")
        total, parts = losses.calvin_losses(out, batch, sequence_length=S, use_dit_head=cfg["use_dit_head"], label_action=lab)
        total.backward()
        reducer.finish()
        bad_g = [nm for nm, p in zip(names, params) if p.grad is not None and not torch.isfinite(p.grad).all()]
        gn = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in params if p.grad is not None)))
        opt.step()
        torch.cuda.synchronize()
        bad_p = [nm for nm, p in zip(names, params) if not torch.isfinite(p).all()]
        print(json.dumps({"step": it, "ms": round((time.perf_counter() - t0) * 1e3, 1), "loss": float(total.detach()), "grad_norm": gn,
                          "nonfinite_grads": len(bad_g), "first_bad_grads": bad_g[:4], "nonfinite_params": len(bad_p), "first_bad_params": bad_p[:4]}), flush=True)


if __name__ == "__main__":
    main()
