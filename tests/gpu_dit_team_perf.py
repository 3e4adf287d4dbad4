"""The DiT evaluation sampler (DDIM-10 + CFG, DiT-B, one episode = 12 token rows) under hipGraph replay: the persistent one-XCD
kernel (csrc/dit_team.hip) against the launch-by-launch sampler, per call; plus a replay-stability run (the captured sampler
replayed with cache-thrashing work in between: every replay must reproduce the eager result bit for bit).  GPU box only, not a
test.  Prints JSON lines; writes gpurun_out/dit_team_perf.jsonl."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402

BF = torch.bfloat16


def build(model_type="DiT-B"):
    from dreamvla_amd.action_model.action_model import ActionModel
    from oracle import weights
    am = ActionModel(token_size=1024, model_type=model_type, in_channels=7, future_action_window_size=2, past_action_window_size=0)
    am.load_state_dict(weights.fill_state_dict(am.state_dict()), strict=True)
    am = am.to(BF).to("cuda").eval()
    am.create_ddim(10)
    return am


def graphed(fn, warm=3):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def time_graph(g, n=20):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def stamps_breakdown(am, cond, noise):
    """where an exchange's time goes: the kernel's own wall-clock stamps (members 0 and 17), medians over the last sampler step"""
    from dreamvla_amd import _lib
    lib = _lib.load()
    depth = len(am.net.blocks)
    n_ex = 10 * (1 + 5 * depth) + 1
    buf = torch.zeros(2 * 8 * n_ex, dtype=torch.int64, device="cuda")
    lib.dvla_dit_sample_set_stamps(buf.data_ptr())
    am.sample_ddim_cfg(cond, noise, 1.5)
    torch.cuda.synchronize()
    lib.dvla_dit_sample_set_stamps(None)
    st = buf.view(n_ex, 2, 8).cpu().double() * 0.01          # us
    names = ["qkv", "attention", "proj", "fc1", "fc2"]
    res = {}
    first = 9 * (1 + 5 * depth) + 1              # exchange index (epoch) of the last step's first block phase
    for member in (0, 1):
        for pi, nm in enumerate(names):
            rows = torch.stack([st[first + 5 * l + pi, member] for l in range(depth)])      # (depth, 8)
            d = {"weights_requested->producers_arrived": (rows[:, 1] - rows[:, 0]).median().item(),
                 "->operands_landed": (rows[:, 2] - rows[:, 1]).median().item() if nm != "attention" else None,
                 "->partials_in_lds": (rows[:, 3] - rows[:, 2]).median().item() if nm != "attention" else None,
                 "->stored": (rows[:, 4] - (rows[:, 3] if nm != "attention" else rows[:, 1])).median().item(),
                 "phase_total": (rows[:, 4] - rows[:, 0]).median().item()}
            res[f"member{(0, 17)[member]}.{nm}"] = {k: (None if v is None else round(v, 2)) for k, v in d.items()}
        blk = torch.stack([st[first + 5 * l, member, 0] for l in range(depth)])
        res[f"member{(0, 17)[member]}.block_period_us"] = round((blk[1:] - blk[:-1]).median().item(), 2)
    return res


def main():
    out = []
    ops.GemmTuner.enabled = False
    for model_type, bs in (("DiT-B", 1),):
        am = build(model_type)
        g0 = torch.Generator().manual_seed(3)
        cond = torch.randn(bs, 3, 1024, generator=g0).to("cuda", BF)
        noise = torch.randn(bs, 3, 7, generator=g0).to(BF).float().to("cuda")
        row = {"model": model_type, "bs": bs, "rows": 12 * bs, "steps": 10}
        am.team_sampler = True
        eager = am.sample_ddim_cfg(cond, noise, 1.5).clone()
        gt, ot = graphed(lambda: am.sample_ddim_cfg(cond, noise, 1.5))
        row["team_us"] = time_graph(gt)
        depth = len(am.net.blocks)
        row["team_us_per_exchange"] = row["team_us"] / (10 * (1 + 5 * depth))
        # replay stability: thrash the caches between replays, compare every replay with the eager result
        big = torch.randn(8192, 8192, device="cuda", dtype=BF)
        bad = 0
        for i in range(40):
            if i % 2:
                (big @ big).sum().item()
            gt.replay()
            torch.cuda.synchronize()
            bad += int(not torch.equal(ot, eager))
        row["replays_differing_from_eager"] = bad
        row["status_xccmask"] = list(ops.dit_team_status(am._fast_tables[("team", "cuda:0", 10)]["ws"]))
        am.team_sampler = True
        row["kernel"] = "call per phase" if os.environ.get("DVLA_DIT_AHEAD") == "0" else "one function, requests ahead"
        row["stamps"] = stamps_breakdown(am, cond, noise)
        am.team_sampler = False
        gl, ol = graphed(lambda: am.sample_ddim_cfg(cond, noise, 1.5))
        row["launch_by_launch_us"] = time_graph(gl)
        row["max_abs_team_vs_launch"] = float((ot - ol).abs().max())
        out.append(row)
        print(json.dumps(row), flush=True)
        del big
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dit_team_perf.jsonl"), "w") as f:
        for row in out:
            f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
