"""ORACLE -- generates the golden fixtures under tests/golden/ by running the REAL reference modules
(imported from /root/reference through oracle/ref_loader.py) on seeded inputs.  Build container only; the
fixtures travel to the GPU box, the reference does not.

    python -m oracle.make_golden            # (re)writes tests/golden/{masks,modules,dreamvla_A,dreamvla_B}.pt

Module fixtures carry their (small) weights; the full-model fixtures use the deterministic weight recipe of
oracle/weights.py (every tensor a function of its state_dict key), so only inputs-by-seed and outputs are stored.
"""
import os
import sys
import tempfile
from functools import partial

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader, weights  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
BF = torch.bfloat16


def _bf(sd):
    return {k: (v.to(BF) if torch.is_floating_point(v) else v) for k, v in sd.items()}


def _round_module_(m, g):
    """seeded bf16-representable parameters for a small reference module"""
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() >= 2:
                std = (2.0 / (p.shape[0] + int(np.prod(p.shape[1:])))) ** 0.5     # xavier-normal, layout agnostic
                p.copy_((torch.randn(p.shape, generator=g) * std).to(BF).float())
            elif name.endswith("weight"):
                p.copy_((1.0 + 0.1 * torch.randn(p.shape, generator=g)).to(BF).float())
            else:
                p.copy_((0.05 * torch.randn(p.shape, generator=g)).to(BF).float())


def rnd(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(BF).float()


def module_fixtures():
    out = {}
    g = torch.Generator().manual_seed(2024)
    # --- timm Block (via the reference's own import of it) and the MAE encoder
    vit = ref_loader.ref_module("models.vit_mae")
    blk = vit.Block(128, 2, 4.0, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6)).eval()
    _round_module_(blk, g)
    x = rnd(g, 3, 37, 128)
    out["timm_block"] = dict(sd=_bf(blk.state_dict()), x=x, y=blk(x).detach(), heads=2, eps=1e-6)
    mae = vit.MaskedAutoencoderViT(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64,
                                   decoder_depth=1, decoder_num_heads=1, mlp_ratio=4,
                                   norm_layer=partial(nn.LayerNorm, eps=1e-6)).eval()
    _round_module_(mae, g)
    mae.pos_embed.data = mae.pos_embed.data.to(BF).float()
    imgs = rnd(g, 3, 3, 64, 64)
    torch.manual_seed(0)
    y, _, ids_restore = mae.forward_encoder(imgs, 0.0)
    y = y.detach()
    patches = torch.gather(y[:, 1:], 1, ids_restore.unsqueeze(-1).expand(-1, -1, y.shape[-1]))  # un-shuffle
    out["vit_encoder"] = dict(sd=_bf(mae.state_dict()), imgs=imgs, y=torch.cat((y[:, :1], patches), 1), depth=2, heads=2)
    # --- perceiver resampler
    pr = ref_loader.ref_module("models.perceiver_resampler")
    res = pr.PerceiverResampler(dim=128, depth=2, dim_head=64, heads=2, num_latents=5).eval()
    _round_module_(res, g)
    x = rnd(g, 4, 1, 1, 21, 128)
    out["perceiver"] = dict(sd=_bf(res.state_dict()), x=x, y=res(x).detach(), depth=2, heads=2)
    # --- GPT-2 trunk (eager and sdpa must agree; block mask)
    g2 = ref_loader.ref_module("models.gpt2")
    dv = ref_loader.ref_module("models.dreamvla_model")
    for impl in ("eager", "sdpa"):
        cfg = g2.GPT2Config()
        cfg.hidden_size, cfg.n_layer, cfg.n_head, cfg.vocab_size, cfg.attn_implementation = 128, 2, 2, 1, impl
        tr = g2.GPT2Model(cfg).eval()
        gg = torch.Generator().manual_seed(77)
        _round_module_(tr, gg)
        mask = dv.generate_attention_mask(3, 6, 7, 0, False, False, False, 0.0, 4, 3)
        x = rnd(gg, 2, mask.shape[0], 128)
        m = mask if impl == "eager" else mask[None, None].expand(2, -1, -1, -1).contiguous()
        out["gpt2_" + impl] = dict(sd=_bf(tr.state_dict()), x=x, mask=mask, y=tr(inputs_embeds=x, attention_mask=m).detach(),
                                   layers=2, heads=2)
    # --- DiT + diffusion
    am = ref_loader.ref_module("models.action_model.action_model")
    mdl = ref_loader.ref_module("models.action_model.models")
    net = mdl.DiT(depth=2, hidden_size=128, num_heads=2, token_size=96, in_channels=7, future_action_window_size=2).eval()
    _round_module_(net, g)
    x, z = rnd(g, 6, 3, 7), rnd(g, 6, 3, 96)
    t = torch.randint(0, 100, (6,), generator=g)
    out["dit"] = dict(sd=_bf(net.state_dict()), x=x, t=t, z=z, y=net(x, t, z).detach(), depth=2, heads=2)
    a = am.ActionModel(token_size=1024, model_type="DiT-B", in_channels=7, future_action_window_size=2, past_action_window_size=0)
    out["diffusion"] = dict(sqrt_acp=torch.from_numpy(a.diffusion.sqrt_alphas_cumprod), betas=torch.from_numpy(a.diffusion.betas),
                            ddim_map=torch.tensor(a.create_ddim(10).timestep_map),
                            ddim_acp=torch.from_numpy(a.ddim_diffusion.alphas_cumprod))
    # --- CLIP text tower (restated shim; small config)
    clip = ref_loader.ref_module("clip")
    ct = clip.CLIPText(embed_dim=64, context_length=16, vocab_size=100, width=128, heads=2, layers=2).eval()
    _round_module_(ct, g)
    tok = torch.randint(1, 90, (5, 16), generator=g)
    tok[torch.arange(5), torch.randint(3, 16, (5,), generator=g)] = 99
    out["clip_text"] = dict(sd=_bf(ct.state_dict()), tokens=tok, y=ct.encode_text(tok).detach(), layers=2, heads=2)
    return out


FULL_CFGS = {
    "A": dict(finetune_type="calvin", sequence_length=2, num_resampler_query=16, num_obs_token_per_image=9,
              action_pred_steps=3, transformer_layers=2, hidden_dim=1024, transformer_heads=16, phase="finetune",
              obs_pred=True, depth_pred=True, sam_feat_pred=True, use_dit_head=False, attn_implementation="sdpa"),
    "B": dict(finetune_type="calvin", sequence_length=2, num_resampler_query=16, num_obs_token_per_image=9,
              action_pred_steps=3, transformer_layers=2, hidden_dim=1024, transformer_heads=16, phase="finetune",
              obs_pred=True, use_dit_head=True, attn_implementation="sdpa"),
    # BASELINE configs[3]: LIBERO flags (scripts/LIBERO/DreamVLA/finetune_long.sh: libero_finetune, --gripper_width, DiT head)
    # with the DINO / SAM / CoTracker-trajectory dream heads switched on
    "E": dict(finetune_type="libero_finetune", sequence_length=2, num_resampler_query=16, num_obs_token_per_image=9,
              action_pred_steps=3, transformer_layers=2, hidden_dim=1024, transformer_heads=16, phase="finetune",
              gripper_width=True, obs_pred=True, dino_feat_pred=True, sam_feat_pred=True, trajectory_pred=True,
              use_dit_head=True, attn_implementation="sdpa"),
}


# the flow-matching action head (--use_fm, eval_libero.py:76 / train.py:96; models/action_model/action_model.py:86-170)
FULL_CFGS["F"] = dict(FULL_CFGS["B"], use_fm=True)


# the BENCHMARKED configuration (BASELINE configs[1], scripts/CALVIN_ABC_D/DreamVLA/finetune.sh): S = 7, 24 layers, head set
# C = obs + depth + sam dream heads + DiT action head; L = 651, key compaction 651 -> 378 on the HIP side
FULL_CFGS["C"] = dict(finetune_type="calvin", sequence_length=7, num_resampler_query=16, num_obs_token_per_image=9,
                      action_pred_steps=3, transformer_layers=24, hidden_dim=1024, transformer_heads=16, phase="finetune",
                      obs_pred=True, depth_pred=True, sam_feat_pred=True, use_dit_head=True, attn_implementation="sdpa")


# the shipped PRETRAIN configuration (scripts/CALVIN_ABC_D/DreamVLA/pretrain.sh:37-52): phase "pretrain", S = 14, 24 layers,
# obs dream head + MLP action head, --atten_goal 4 --atten_goal_state --atten_only_obs --attn_robot_proprio_state; L = 798.
# Besides the eval()-module outputs the fixture holds the outputs of the module in TRAINING mode -- where
# models/dreamvla_model.py:610-628 regenerates the attention mask every forward -- with every nn.Dropout p set to 0 (the
# dropout streams are not reproducible across implementations; the mask regeneration and the module wiring around it are)
FULL_CFGS["P"] = dict(finetune_type="calvin", sequence_length=14, num_resampler_query=16, num_obs_token_per_image=9,
                      action_pred_steps=3, transformer_layers=24, hidden_dim=1024, transformer_heads=16, phase="pretrain",
                      obs_pred=True, use_dit_head=False, atten_goal=4, atten_goal_state=True, atten_only_obs=True,
                      attn_robot_proprio_state=True, attn_implementation="sdpa")

# the shipped LIBERO configuration at full size (scripts/LIBERO/DreamVLA/finetune_long.sh:21-65: --finetune_type libero_finetune,
# --gripper_width, obs + sam dream heads, DiT head, S = 7, 24 layers; models/dreamvla_model.py:656-664 takes the two finger widths
# instead of the one-hot gripper state); L = (36 + 18 * 2 + 3) * 7 = 525.  Round-4 VERDICT missing #4: fixture E has these flags
# at S = 2 / 2 layers only.
FULL_CFGS["D"] = dict(finetune_type="libero_finetune", sequence_length=7, num_resampler_query=16, num_obs_token_per_image=9,
                      action_pred_steps=3, transformer_layers=24, hidden_dim=1024, transformer_heads=16, phase="finetune",
                      gripper_width=True, obs_pred=True, sam_feat_pred=True, use_dit_head=True, attn_implementation="sdpa")

# the ROLLOUT configuration that bench.py's `rollout` leg times (BASELINE configs[4], scripts/CALVIN_ABC_D/DreamVLA/eval.sh):
# S = 10 history, 24 layers, head set C weights, DiT head sampled with DDIM-10 + CFG; L = 930
FULL_CFGS["R"] = dict(FULL_CFGS["C"], sequence_length=10)


def fake_mae_ckpt():
    vit = ref_loader.ref_module("models.vit_mae")
    mae = vit.MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                                   decoder_depth=8, decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    path = os.path.join(tempfile.gettempdir(), "dvla_fake_mae.pth")
    torch.save({"model": mae.state_dict()}, path)
    return path


def build_reference_model(cfg):
    dv = ref_loader.ref_module("models.dreamvla_model")
    m = dv.DreamVLA(clip_device="cpu", vit_checkpoint_path=fake_mae_ckpt(), **cfg).float()
    m.load_state_dict(weights.fill_state_dict(m.state_dict()), strict=True)
    m.clip_model.requires_grad_(False)
    m.vision_encoder.requires_grad_(False)
    m._init_model_type()
    return m.eval()


class _Done(Exception):
    pass


def full_fixture(name):
    cfg = FULL_CFGS[name]
    m = build_reference_model(cfg)
    B, S = 1, cfg["sequence_length"]
    b = weights.synthetic_batch(B, S, window=S + 3, seed=4321)
    # label actions exactly as utils/train_utils.py:145 builds them
    label = torch.cat([b["actions"][:, j:S + j, :].unsqueeze(-2) for j in range(cfg["action_pred_steps"])], dim=-2)
    for k in ("image_primary", "image_wrist", "state", "text_token"):
        b[k] = b[k][:, :S]
    fx = dict(cfg=cfg, B=B, S=S, window=S + 3, seed=4321, action_label=label)
    if cfg.get("gripper_width"):   # 6 arm values + the two finger widths (train_utils.py:128-129); stored, it is tiny
        gw = torch.Generator().manual_seed(4322)
        b["state"] = torch.cat([b["state"][..., :6], (torch.rand(B, S, 2, generator=gw) * 0.08).to(BF).float()], dim=-1)
        fx["state"] = b["state"].clone()
    torch.manual_seed(0)
    real = {k: getattr(torch, k) for k in ("randn_like", "randint", "randn")}
    if cfg["use_dit_head"]:
        g = torch.Generator().manual_seed(99)
        n_rep = 8 * B * S
        noise = torch.randn(n_rep, 3, 7, generator=g).to(BF).float()
        tstep = torch.randint(0, 10 if cfg.get("use_fm") else 100, (n_rep,), generator=g)
        test_noise = torch.randn(B * S, 3, 7, generator=g).to(BF).float()
        fx.update(dit_noise=noise, dit_timestep=tstep, test_noise=test_noise)
        torch.randn_like = lambda x, **k: noise.clone()
        torch.randint = lambda *a, **k: tstep.clone()
    try:
        with torch.no_grad():
            out = m(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], action=None,
                    action_label=label, mode="train")
            fx["train"] = [None if o is None else o.detach().clone() for o in out]
            if cfg["use_dit_head"]:
                torch.randn_like, torch.randint = real["randn_like"], real["randint"]
                # DDIM: the model draws (bs, 3, 7) and doubles it.  Flow matching: FMDiffusion ignores that and draws its own
                # (2 bs, 3, 7) start noise -- on device='cuda', hard-coded (respace.py:139): the patched randn ignores the device
                # and hands out the same noise for both halves
                def fake_randn(*a, **k):
                    n0 = a[0][0] if isinstance(a[0], (tuple, list, torch.Size)) else a[0]
                    return test_noise.clone() if n0 == B * S else torch.cat([test_noise, test_noise], 0)
                torch.randn = fake_randn
                out = m(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], mode="test")
                fx["test"] = [None if o is None else o.detach().clone() for o in out]
                if name == "R":      # the rollout fixture needs the sampler outputs only
                    raise _Done
                # the reference's OWN bf16 path (train.py --precision amp_bf16 -> autocast) on the same inputs / noise, five
                # times (CPU bf16 GEMMs are not run-to-run deterministic): its scatter around the fp32 value is the bf16 floor
                # of the action-MSE and sets the tolerance of the GPU check (tests/model_checks.py).  Run last, so the fp32
                # outputs above stay bit-identical to earlier fixture generations.
                torch.randn = real["randn"]
                torch.randn_like = lambda x, **k: noise.clone()
                torch.randint = lambda *a, **k: tstep.clone()
                runs = []
                for _ in range(5 if cfg["transformer_layers"] <= 2 else 2):
                    with torch.autocast("cpu", dtype=BF):
                        o16 = m(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], action=None,
                                action_label=label, mode="train")
                    runs.append(float(o16[0]))
                fx["train_loss_ref_amp_bf16_runs"] = runs
            if cfg["phase"] == "pretrain":
                # TRAINING-mode forward: the reference rebuilds self.attention_mask from the rule (dreamvla_model.py:610-628;
                # mask_l_obs_ratio = 0: nothing random in it); dropout off so that the result is reproducible
                for mod in m.modules():
                    if isinstance(mod, nn.Dropout):
                        mod.p = 0.0
                m.train()
                m.device = torch.device("cpu")     # the regenerated mask is moved `.to(self.device)` (dreamvla_model.py:627)
                before = m.attention_mask
                out = m(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], action=None,
                        action_label=label, mode="train")
                assert m.attention_mask is not before, "the training-mode forward did not regenerate the mask"
                fx["training_mode"] = [None if o is None else o.detach().clone() for o in out]
                m.eval()
    except _Done:
        pass
    finally:
        for k, v in real.items():
            setattr(torch, k, v)
    # keep the fixture small: store big dream-head outputs as a strided sample + checksum
    for key in ("train", "training_mode"):
        if key not in fx:
            continue
        outs = fx[key]
        for i, o in enumerate(outs):
            if o is not None and o.numel() > 20000:
                flat = o.flatten()
                idx = torch.linspace(0, flat.numel() - 1, 4096).long()
                outs[i] = dict(shape=list(o.shape), idx=idx, vals=flat[idx].clone(), mean=float(flat.mean()),
                               l2=float(flat.norm()))
    return fx


def _patch_q_sample(m):
    """The reference's DiT loss cannot run under `--precision bf16` as written: GaussianDiffusion.q_sample multiplies by
    float32 schedule tensors, so x_t reaches the bf16 DiT as float32 and F.linear raises (action_model.py:57-66; the
    shipped scripts use --precision fp32).  For the bf16-cast DEVIATION records only (tolerance floors, never golden
    values) x_t is cast to the noise's dtype -- the one-line fix any bf16 run of the reference would need."""
    am = getattr(m, "action_model", None)
    if am is None or not hasattr(am, "diffusion"):
        return
    orig = am.diffusion.q_sample
    am.diffusion.q_sample = lambda x, t, noise=None: orig(x, t, noise).to(x.dtype)
    if type(am).__name__ == "ActionModelFM":
        # same defect in the flow-matching loss (action_model.py:127-131): the float32 `timestep` promotes x_t to float32
        net_fwd = am.net.forward
        am.net.forward = lambda x, t, z: net_fwd(x.to(next(am.net.parameters()).dtype), t, z)


def _out_rel(a, b):
    a, b = a.detach().float().flatten(), b.detach().float().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


def amp_deviation(name, runs=2):
    """The REAL reference's own bf16 path (train.py --precision amp_bf16 = torch.autocast) against its own fp32 outputs on
    the fixture's inputs, per output: rel-L2 and max-abs / absmax.  That deviation is the floor any bf16 implementation of
    the model sits on; tests/model_checks.py sets each output's tolerance to max(1e-3, 2 x this).  Added to the existing
    fixture file (the stored fp32 outputs are not touched)."""
    path = os.path.join(GOLD, f"dreamvla_{name}.pt")
    fx = torch.load(path, map_location="cpu")
    cfg = fx["cfg"]
    m = build_reference_model(cfg)
    B, S = fx["B"], fx["S"]
    b = weights.synthetic_batch(B, S, window=fx["window"], seed=fx["seed"])
    for k in ("image_primary", "image_wrist", "state", "text_token"):
        b[k] = b[k][:, :S]
    if "state" in fx:
        b["state"] = fx["state"]
    real = {k: getattr(torch, k) for k in ("randn_like", "randint", "randn")}
    if cfg["use_dit_head"]:
        noise, tstep = fx["dit_noise"], fx["dit_timestep"]
        torch.randn_like = lambda x, **k: noise.clone()
        torch.randint = lambda *a, **k: tstep.clone()
    try:
        with torch.no_grad():
            ref = m(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], action=None,
                    action_label=fx["action_label"], mode="train")
            devs = []
            for _ in range(runs):
                with torch.autocast("cpu", dtype=BF):
                    o16 = m(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], action=None,
                            action_label=fx["action_label"], mode="train")
                devs.append([None if r is None else
                             dict(rel_l2=_out_rel(o, r), max_abs=float((o.float() - r).abs().max()), absmax=float(r.abs().max()))
                             for o, r in zip(o16, ref)])
    finally:
        for k, v in real.items():
            setattr(torch, k, v)
    worst = []
    for i in range(len(ref)):
        if ref[i] is None:
            worst.append(None)
        else:
            worst.append(dict(rel_l2=max(d[i]["rel_l2"] for d in devs), max_abs=max(d[i]["max_abs"] for d in devs),
                              absmax=devs[0][i]["absmax"]))
    fx["ref_amp_bf16_deviation"] = worst
    # the reference's `--precision bf16` mode proper (train.py:122-123: model.bfloat16(), inputs cast by the loop): every
    # parameter, the residual stream and the normalisations in bf16 -- the mode the HIP path implements
    real = {k: getattr(torch, k) for k in ("randn_like", "randint", "randn")}
    if cfg["use_dit_head"]:
        noise, tstep = fx["dit_noise"], fx["dit_timestep"]
        torch.randn_like = lambda x, **k: noise.clone().to(x.dtype)
        torch.randint = lambda *a, **k: tstep.clone()
    try:
        m16 = m.bfloat16()
        m16._init_model_type()
        _patch_q_sample(m16)
        with torch.no_grad():
            o16 = m16(b["image_primary"].to(BF), b["image_wrist"].to(BF), b["state"].to(BF), b["text_token"], action=None,
                      action_label=fx["action_label"].to(BF), mode="train")
    finally:
        for k, v in real.items():
            setattr(torch, k, v)
    fx["ref_bf16_cast_deviation"] = [None if r is None else
                                     dict(rel_l2=_out_rel(o, r), max_abs=float((o.float() - r).abs().max()), absmax=float(r.abs().max()))
                                     for o, r in zip(o16, ref)]
    # the evaluation path (mode="test": DDIM-10 + CFG or the flow-matching Euler loop): the reference's own bf16 runs --
    # autocast on the fp32 module, then the bf16-cast module -- against its fp32 samples (fx["test"]) with the same start noise.
    # Ten sampler steps feed the denoiser its own output, so bf16 noise is amplified: this record, not a guess, sets the
    # tolerance of the GPU test-mode checks (tests/model_checks.py)
    if "test" in fx and cfg["use_dit_head"]:
        tn, bs_ = fx["test_noise"], fx["test_noise"].shape[0]

        def fake_randn(*a, **k):
            n0 = a[0][0] if isinstance(a[0], (tuple, list, torch.Size)) else a[0]
            t = tn.clone() if n0 == bs_ else torch.cat([tn, tn], 0)
            return t.to(k["dtype"]) if k.get("dtype") is not None else t
        real_randn = torch.randn
        m32 = build_reference_model(cfg)        # (`m` was cast to bf16 in place above)
        torch.randn = fake_randn
        try:
            devs = []
            with torch.no_grad():
                with torch.autocast("cpu", dtype=BF):
                    o_amp = m32(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], mode="test")
                o_cast = m16(b["image_primary"].to(BF), b["image_wrist"].to(BF), b["state"].to(BF), b["text_token"], mode="test")
            for o16 in (o_amp, o_cast):
                devs.append([None if r is None else dict(rel_l2=_out_rel(o, r), max_abs=float((o.float() - r).abs().max()),
                                                         absmax=float(r.abs().max())) for o, r in zip(o16, fx["test"])])
            fx["ref_test_bf16_deviation"] = [None if devs[0][i] is None else
                                             dict(rel_l2=max(d[i]["rel_l2"] for d in devs), max_abs=max(d[i]["max_abs"] for d in devs),
                                                  absmax=devs[0][i]["absmax"]) for i in range(len(fx["test"]))]
            print(name, "test-mode bf16", [None if w is None else round(w["rel_l2"], 5) for w in fx["ref_test_bf16_deviation"]])
        finally:
            torch.randn = real_randn
    torch.save(fx, path)
    print(name, "autocast", [None if w is None else round(w["rel_l2"], 5) for w in worst])
    print(name, "bf16 cast", [None if w is None else round(w["rel_l2"], 5) for w in fx["ref_bf16_cast_deviation"]])


def grad_fixture(name):
    """Gradients of the REAL reference (its own autograd, fp32) for the scalar  sum_o <o, w_o>  (w_o seeded, one per
    output; tests/model_checks.py::output_weights builds the same) on the fixture's inputs: per trainable tensor its
    norm and a strided sample, plus the reference's own amp-bf16 (autocast) gradient deviation rel-L2 from them.
    tests/golden/grads_<name>.pt pins the oracle's autograd (CPU test) and sets the per-tensor tolerance of the GPU
    gradient check (max(floor, 2 x the reference's own deviation))."""
    path = os.path.join(GOLD, f"dreamvla_{name}.pt")
    fx = torch.load(path, map_location="cpu")
    cfg = fx["cfg"]
    m = build_reference_model(cfg)
    B, S = fx["B"], fx["S"]
    b = weights.synthetic_batch(B, S, window=fx["window"], seed=fx["seed"])
    for k in ("image_primary", "image_wrist", "state", "text_token"):
        b[k] = b[k][:, :S]
    if "state" in fx:
        b["state"] = fx["state"]
    real = {k: getattr(torch, k) for k in ("randn_like", "randint", "randn")}
    if cfg["use_dit_head"]:
        noise, tstep = fx["dit_noise"], fx["dit_timestep"]
        torch.randn_like = lambda x, **k: noise.clone()
        torch.randint = lambda *a, **k: tstep.clone()

    def run(amp):
        m.zero_grad(set_to_none=True)
        if amp:
            with torch.autocast("cpu", dtype=BF):
                out = m(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], action=None,
                        action_label=fx["action_label"], mode="train")
        else:
            out = m(b["image_primary"], b["image_wrist"], b["state"], b["text_token"], action=None,
                    action_label=fx["action_label"], mode="train")
        g = torch.Generator().manual_seed(5)
        seen, loss = set(), 0.0
        for o in out:
            if o is None:
                continue
            w = torch.randn(o.shape, generator=g).to(BF).float()
            if id(o) in seen:          # DiT training returns the same loss tensor in slots 0 and 1: count it once
                continue
            seen.add(id(o))
            loss = loss + (o.float() * w).sum()
        loss.backward()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    try:
        g32 = run(False)
        g16 = run(True)
        # `--precision bf16` proper: the whole module cast (parameters, residual stream, normalisations in bf16)
        if cfg["use_dit_head"]:
            torch.randn_like = lambda x, **k: noise.clone().to(x.dtype)
        m.bfloat16()
        m._init_model_type()
        _patch_q_sample(m)
        for k in ("image_primary", "image_wrist", "state"):
            b[k] = b[k].to(BF)
        fx["action_label"] = fx["action_label"].to(BF)
        gcast = run(False)
    finally:
        for k, v in real.items():
            setattr(torch, k, v)
    entries = {}
    for k, g in g32.items():
        n = float(g.norm())
        if n == 0.0:
            continue
        flat = g.flatten()
        idx = torch.linspace(0, flat.numel() - 1, min(flat.numel(), 512)).long()
        dev = float((g16[k].float() - g).norm() / n) if k in g16 else None
        devc = float((gcast[k].float() - g).norm() / n) if k in gcast else None
        entries[k] = dict(norm=n, idx=idx, vals=flat[idx].clone(), amp_rel_l2=dev, cast_rel_l2=devc, absmax=float(g.abs().max()))
    torch.save(dict(entries=entries, source="real reference autograd (fp32) + its own autocast-bf16 deviation; oracle/make_golden.py grads"),
               os.path.join(GOLD, f"grads_{name}.pt"))
    devs = sorted((e["amp_rel_l2"] for e in entries.values() if e["amp_rel_l2"] is not None))
    print(name, len(entries), "tensors; reference amp-bf16 gradient deviation median %.4f max %.4f" % (devs[len(devs) // 2], devs[-1]))
    devs = sorted((e["cast_rel_l2"] for e in entries.values() if e["cast_rel_l2"] is not None))
    print(name, "reference bf16-cast gradient deviation median %.4f max %.4f" % (devs[len(devs) // 2], devs[-1]))


def main(only=()):
    """`python -m oracle.make_golden E` regenerates only the named full-model fixture(s)"""
    assert ref_loader.available(), "needs /root/reference"
    os.makedirs(GOLD, exist_ok=True)
    src = "generated by oracle/make_golden.py from the REAL reference modules under /root/reference"
    if only and only[0] == "grads":
        for name in only[1:]:
            grad_fixture(name)
        return
    if only and only[0] == "amp":
        for name in only[1:]:
            amp_deviation(name)
        return
    if only:
        for name in only:
            fx = full_fixture(name)
            fx["source"] = src
            torch.save(fx, os.path.join(GOLD, f"dreamvla_{name}.pt"))
            print(name, os.path.getsize(os.path.join(GOLD, f"dreamvla_{name}.pt")))
        return
    # masks
    from tests.test_mask import COMBOS
    ref = ref_loader.ref_module("models.dreamvla_model").generate_attention_mask
    packed = []
    for kw in COMBOS:
        np.random.seed(123)
        packed.append(torch.from_numpy(np.packbits((ref(**kw) == 0).numpy())))
    torch.save({"packed": packed, "source": src}, os.path.join(GOLD, "masks.pt"))
    mf = module_fixtures()
    mf["source"] = src
    torch.save(mf, os.path.join(GOLD, "modules.pt"))
    # state_dict surface (keys + shapes) of the shipped CALVIN finetune configuration at full size
    import json
    cfgC = dict(finetune_type="calvin", sequence_length=7, num_resampler_query=16, num_obs_token_per_image=9,
                action_pred_steps=3, transformer_layers=24, hidden_dim=1024, transformer_heads=16, phase="finetune",
                obs_pred=True, depth_pred=True, sam_feat_pred=True, use_dit_head=True, attn_implementation="sdpa")
    dv = ref_loader.ref_module("models.dreamvla_model")
    mC = dv.DreamVLA(clip_device="cpu", vit_checkpoint_path=fake_mae_ckpt(), **cfgC)
    with open(os.path.join(GOLD, "state_dict_surface_C.json"), "w") as f:
        json.dump({"cfg": cfgC, "source": src, "trainable": sorted(n for n, p in mC.named_parameters() if p.requires_grad),
                   "entries": {k: list(v.shape) for k, v in mC.state_dict().items()}}, f)
    del mC
    for name in FULL_CFGS:
        if only and name not in only:
            continue
        fx = full_fixture(name)
        fx["source"] = src
        torch.save(fx, os.path.join(GOLD, f"dreamvla_{name}.pt"))
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main(tuple(sys.argv[1:]))
