"""Parity of dreamvla_amd.rollout.RolloutEngine with the reference's evaluation wrapper semantics
(utils/eval_utils_calvin.py:82-147: history queues, last-frame padding, full-window `model(..., mode="test")`,
action of the newest real frame), used by tests/test_rollout.py (CPU host logic) and tests/test_model_gpu.py (GPU)."""
from collections import deque

import torch


class WindowOracle:
    """the reference wrapper's queue logic, one instance per episode (restated; no model inside)"""

    def __init__(self, S):
        self.S = S
        self.q = deque(maxlen=S)

    def push(self, frame):
        self.q.append(frame)
        frames = list(self.q)
        k = len(frames)
        window = frames + [frames[-1]] * (self.S - k)          # eval_utils_calvin.py:118-126
        pick = k - 1 if k < self.S else self.S - 1             # :141-146
        return window, pick


def gpu_rollout_checks(head="mlp", use_graph=True, steps=7, tol=2e-2):
    """engine vs full-window forward on cuda, B = 3 episodes, S = 4, one episode reset mid-way."""
    from dreamvla_amd.dreamvla_model import DreamVLA
    from dreamvla_amd.rollout import RolloutEngine
    from oracle import weights
    BF = torch.bfloat16
    S, B = 4, 3
    cfg = dict(finetune_type="calvin", sequence_length=S, num_resampler_query=16, num_obs_token_per_image=9,
               action_pred_steps=3, transformer_layers=2, hidden_dim=1024, transformer_heads=16, phase="finetune",
               obs_pred=True, use_dit_head=(head == "dit"), attn_implementation="sdpa")
    m = DreamVLA(clip_device="cpu", vit_checkpoint_path=None, **cfg)
    m.load_state_dict(weights.fill_state_dict(m.state_dict()), strict=True)
    m = m.to(BF).to("cuda")
    m._init_model_type()
    m.eval()
    eng = RolloutEngine(m, B, use_graph=use_graph, warmup_decodes=2)
    g = torch.Generator().manual_seed(5)
    text = torch.randint(1, 49000, (B, 77), generator=g)
    text[:, 20] = 49407
    text[:, 21:] = 0
    oracles = [WindowOracle(S) for _ in range(B)]
    res = []
    real_randn = torch.randn
    for t in range(steps):
        if t == 5:                                  # restart episode 1 only
            mask = torch.tensor([False, True, False])
            eng.reset(mask)
            oracles[1] = WindowOracle(S)
        fr = dict(ip=torch.randn(B, 3, 224, 224, generator=g).to(BF), iw=torch.randn(B, 3, 224, 224, generator=g).to(BF),
                  st=torch.cat([torch.rand(B, 6, generator=g), (torch.rand(B, 1, generator=g) > 0.5).float()], -1).to(BF))
        wins, picks = zip(*[o.push({k: v[b] for k, v in fr.items()}) for b, o in enumerate(oracles)])
        noise = real_randn(B * S, 3, 7, generator=g).to("cuda")
        if head == "dit" and not use_graph:
            torch.randn = lambda *a, **k: noise.clone()
        try:
            action, arm, grip = eng.step(fr["ip"], fr["iw"], fr["st"], text)
            stack = lambda key: torch.stack([torch.stack([f[key] for f in w]) for w in wins]).to("cuda")
            with torch.no_grad():
                out = m(stack("ip"), stack("iw"), stack("st"), text.unsqueeze(1).repeat(1, S, 1).to("cuda"), mode="test")
        finally:
            torch.randn = real_randn
        ra, rg = out[0], out[1]
        if head == "dit":
            ra, rg = ra.view(B, S, 3, 6), rg.view(B, S, 3, 1)
        want = torch.stack([ra[b, picks[b], 0].float() for b in range(B)])
        got = action[:, :6]
        ok_sel = bool((eng.count - 1 == torch.tensor(picks)).all())
        res.append({"name": f"rollout.{head}.graph{int(use_graph)}.t{t}.window_pick", "rel_l2": 0.0, "tol": 0.0, "ok": ok_sel})
        finite = bool(torch.isfinite(action).all()) and bool(((action[:, 6].abs() - 1).abs() < 1e-6).all())
        res.append({"name": f"rollout.{head}.graph{int(use_graph)}.t{t}.finite", "rel_l2": 0.0, "tol": 0.0, "ok": finite})
        if head == "mlp" or not use_graph:          # with the DiT head under a graph the sampler noise is the graph's own
            r = float((got - want).norm() / max(float(want.norm()), 1e-12))
            res.append({"name": f"rollout.{head}.graph{int(use_graph)}.t{t}.action", "rel_l2": r, "tol": tol, "ok": r <= tol})
            r2 = float((arm.float() - ra.float()).norm() / max(float(ra.float().norm()), 1e-12))
            res.append({"name": f"rollout.{head}.graph{int(use_graph)}.t{t}.arm_all_positions", "rel_l2": r2, "tol": tol, "ok": r2 <= tol})
    if use_graph:
        res.append({"name": f"rollout.{head}.graph_captured", "rel_l2": 0.0, "tol": 0.0, "ok": eng.graphs_captured})
    return res
