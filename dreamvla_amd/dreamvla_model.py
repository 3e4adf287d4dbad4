"""DreamVLA on the MI355X HIP kernels -- host-side mirror of /root/reference/models/dreamvla_model.py.

Drop-in surface: same constructor keywords and defaults (dreamvla_model.py:123-166),
`forward(image_primary, image_wrist, state, text_token, action=None, track_infos=None, action_label=None,
mode='train')` -> the same 10-tuple (609, 991), `_init_model_type()`, the attributes the callers poke
(`image_processor`, `clip_model`, `vision_encoder`, `perceiver_resampler`, `transformer_backbone`, the projectors,
`sequence_length`) and the same parameter names / shapes (App. B) so reference checkpoints load.

What runs where: every GEMM / LayerNorm / attention / MLP below is a hand-written gfx950 kernel reached through
include/dvla.h (dreamvla_amd/ops.py).  Token assembly (cat / expand / position add / slicing) is still done with
torch tensor ops on the device -- pure data movement, < 1 % of step time, listed in DESIGN.md.

Deliberate deviations from the reference, each parity-neutral:
  * both camera views go through the frozen ViT and the resampler as ONE batch (same weights for both views,
    dreamvla_model.py:672-673,716-717);
  * the ViT does not randomly permute patch tokens (see dreamvla_amd/vit_mae.py);
  * the (B,1,L,L) materialised copy of the mask (769-774) is replaced by bit tables built once per mask;
  * `vit_checkpoint_path=None` skips loading the MAE checkpoint (random init) -- the reference would crash.
"""
import os

import numpy as np
import torch
from torch import nn

from . import clip_text, ops
from .action_model import ActionModel, ActionModelFM
from .gpt2 import GPT2Config, GPT2Model
from .nn import Block, LayerNorm, Linear
from .perceiver_resampler import PerceiverResampler
from .vit_mae import (MaskedAutoencoderViT, get_1d_sincos_pos_embed_from_grid, get_2d_sincos_pos_embed,  # noqa: F401
                      get_2d_sincos_pos_embed_from_grid)

NEG_INF = -float("inf")


def generate_attention_mask(K, num_A, num_B, atten_goal, atten_goal_state, atten_only_obs, attn_robot_proprio_state,
                            mask_l_obs_ratio, num_obs_token, action_pred_steps):
    """Additive (L, L) mask, L = (num_A + num_B) * K, values 0 / -inf -- same semantics, flag handling and
    numpy-RNG consumption as dreamvla_model.py:25-66 (pinned bit-exactly by tests/test_mask.py):
      * timestep block i cannot see blocks after it;
      * the num_B query/prediction tokens of every block are invisible as keys to everybody ...
      * ... except that the action tokens of a block see that block's obs/query tokens;
      * atten_only_obs / attn_robot_proprio_state / mask_l_obs_ratio / atten_goal(+_state): pretrain variants.
    """
    blk = num_A + num_B
    L = blk * K
    m = torch.zeros((L, L))
    for i in range(K):
        s = i * blk
        e = s + blk
        m[s:e, e:] = NEG_INF
        m[:, s + num_A:e] = NEG_INF
        a0 = s + num_A + num_obs_token            # first action token row
        a1 = a0 + action_pred_steps
        o0, o1 = s + num_A, s + num_A + num_obs_token
        if num_obs_token > 0 and action_pred_steps:
            m[a0:a1, o0:o1] = 0.0
        if num_obs_token > 0 and atten_only_obs and action_pred_steps:
            m[a0:a1] = NEG_INF
            m[a0:a1, s + 2:s + num_A] = 0.0
            m[a0:a1, o0:o1] = 0.0
            if attn_robot_proprio_state:
                m[a0:a1, s + 1:s + 2] = 0.0
            if mask_l_obs_ratio > 0:
                count = int(mask_l_obs_ratio * num_obs_token)
                for num in np.random.choice(range(num_obs_token), size=count, replace=False):
                    m[a0:a1, o0 + num] = NEG_INF
        if num_obs_token > 0 and atten_goal:
            if i < K - atten_goal:
                pred_end_index = (i + atten_goal) * blk
                if atten_goal_state:
                    m[o0:o1, pred_end_index + 1:pred_end_index + 2] = 0.0
    return m


def get_1d_sincos_pos_embed(embed_dim, length, scale=1.0):
    pos = np.arange(0, length)[..., None] / scale
    return get_1d_sincos_pos_embed_from_grid(embed_dim, pos)


class SiLogLoss(nn.Module):
    """utils/sigloss.py:6-15 (parameter-free; kept as a module attribute for surface parity)."""

    def __init__(self, lambd=0.5):
        super().__init__()
        self.lambd = lambd

    def forward(self, pred, target):
        diff_log = torch.log(target + 1e-6) - torch.log(pred + 1e-6)
        return torch.sqrt(torch.pow(diff_log, 2).mean() - self.lambd * torch.pow(diff_log.mean(), 2))


def _decoder(dim):
    return nn.Sequential(Block(dim, num_heads=16, mlp_ratio=4, qkv_bias=True, norm_layer=LayerNorm),
                         Block(dim, num_heads=16, mlp_ratio=4, qkv_bias=True, norm_layer=LayerNorm))


class DreamVLA(nn.Module):
    def __init__(self, finetune_type, clip_device, vit_checkpoint_path, sequence_length=10, num_resampler_query=9,
                 num_obs_token_per_image=10, obs_pred=False, atten_only_obs=False, attn_robot_proprio_state=False,
                 atten_goal=False, atten_goal_state=False, mask_l_obs_ratio=0.0, calvin_input_image_size=224,
                 patch_size=16, mask_ratio=0.0, num_token_per_timestep=41, input_self=False, action_pred_steps=1,
                 transformer_layers=12, hidden_dim=384, transformer_heads=12, phase="", gripper_width=False,
                 pred_num=1, depth_pred=False, trajectory_pred=False, use_depth_query=False, use_dpt_head=False,
                 use_trajectory_query=False, track_label_patch_size=4, dino_feat_pred=False, sam_feat_pred=False,
                 use_dinosiglip=False, use_dit_head=False, use_gpt2_pretrained=False, no_pred_gripper_traj=False,
                 no_unshuffle=False, share_query=False, attn_implementation=False, use_fm=False, dit_type="DiT-B"):
        super().__init__()
        if use_dinosiglip:
            raise NotImplementedError("use_dinosiglip needs timm hub checkpoints (dreamvla_model.py:479-509); not on the "
                                      "round-1 hot path")
        if use_dpt_head:
            raise NotImplementedError("use_dpt_head needs the external Depth-Anything-V2 head (dreamvla_model.py:516-537)")
        if use_gpt2_pretrained:
            raise NotImplementedError("use_gpt2_pretrained is broken in the reference (SURVEY.md App. F item 7)")
        self.finetune_type = finetune_type
        self.device = clip_device
        self.sequence_length = sequence_length
        self.action_pred_steps = action_pred_steps
        self.obs_pred, self.depth_pred = obs_pred, depth_pred
        self.dino_feat_pred, self.sam_feat_pred, self.trajectory_pred = dino_feat_pred, sam_feat_pred, trajectory_pred
        self.atten_goal, self.atten_goal_state = atten_goal, atten_goal_state
        self.atten_only_obs, self.attn_robot_proprio_state = atten_only_obs, attn_robot_proprio_state
        self.mask_l_obs_ratio = mask_l_obs_ratio
        self.hidden_dim = hidden_dim
        self.phase = phase
        self.dit_type = dit_type
        assert self.phase in ["pretrain", "finetune", "evaluate"]
        self.share_query = share_query
        self.gripper_width = gripper_width
        self.share_text_over_time = True   # encode_frames: run the text tower once per sample when its S token rows are equal
        self.vit_checkpoint_path = vit_checkpoint_path
        self.pred_num = pred_num
        H = self.hidden_dim

        self.text_projector = Linear(512, H)
        self.arm_state_encoder = Linear(6, H)
        self.gripper_state_encoder = Linear(2, H)
        self.state_projector = Linear(2 * H, H)
        # constructed-but-unused modules of the reference (state_dict surface; dreamvla_model.py:203-205,320-333)
        self.action_pose_encoder = Linear(6, H)
        self.action_gripper_position_encoder = Linear(2, H)
        self.action_projector = Linear(2 * H, H)

        self.use_dinosiglip = use_dinosiglip
        self.vision_encoder = MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12,
                                                   decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16,
                                                   mlp_ratio=4, norm_layer=lambda d: LayerNorm(d, eps=1e-6))
        self.RESAMPLER_hidden_dim = 768
        self.NUM_RESAMPLER_QUERY = num_resampler_query
        self.perceiver_resampler = PerceiverResampler(dim=self.RESAMPLER_hidden_dim, num_latents=self.NUM_RESAMPLER_QUERY,
                                                      depth=3)
        self.image_primary_projector = Linear(self.RESAMPLER_hidden_dim, H)
        self.cls_token_primary_projector = Linear(768, H)
        self.image_wrist_projector = Linear(self.RESAMPLER_hidden_dim, H)
        self.cls_token_wrist_projector = Linear(768, H)

        if self.action_pred_steps > 0:
            self.action_pred_token = nn.Parameter(torch.zeros(1, 1, self.action_pred_steps, H))

        self.NUM_OBS_TOKEN = self.NUM_DEPTH_TOKEN = self.NUM_TRAJ_TOKEN = self.NUM_DINO_TOKEN = self.NUM_SAM_TOKEN = 0
        if self.obs_pred:
            self.NUM_OBS_TOKEN_PER_IMAGE = num_obs_token_per_image
            self.NUM_OBS_TOKEN = num_obs_token_per_image * 2
            self.obs_tokens = nn.Parameter(torch.zeros(1, 1, self.NUM_OBS_TOKEN, H))
        if self.depth_pred:
            self.NUM_OBS_TOKEN_PER_DEPTH = num_obs_token_per_image
            self.NUM_DEPTH_TOKEN = num_obs_token_per_image * 2
        if self.dino_feat_pred:
            self.NUM_OBS_TOKEN_PER_DINO = num_obs_token_per_image
            self.NUM_DINO_TOKEN = num_obs_token_per_image * 2
        if self.sam_feat_pred:
            self.NUM_OBS_TOKEN_PER_SAM = num_obs_token_per_image
            self.NUM_SAM_TOKEN = num_obs_token_per_image * 2
        if self.trajectory_pred:
            self.NUM_OBS_TOKEN_PER_TRAJ = num_obs_token_per_image
            self.NUM_TRAJ_TOKEN = num_obs_token_per_image * (1 if no_pred_gripper_traj else 2)
        if not self.share_query:
            if self.depth_pred:
                self.depth_tokens = nn.Parameter(torch.zeros(1, 1, self.NUM_DEPTH_TOKEN, H))
            if self.dino_feat_pred:
                self.dino_feat_tokens = nn.Parameter(torch.zeros(1, 1, self.NUM_DINO_TOKEN, H))
            if self.sam_feat_pred:
                self.sam_feat_tokens = nn.Parameter(torch.zeros(1, 1, self.NUM_SAM_TOKEN, H))
            if trajectory_pred:
                self.trajectory_tokens = nn.Parameter(torch.zeros(1, 1, self.NUM_TRAJ_TOKEN, H))

        self.embedding_layer_norm = LayerNorm(H)
        self.attention_mask = nn.Parameter(self._make_mask(), requires_grad=False)
        self.transformer_backbone_position_embedding = nn.Parameter(torch.zeros(1, self.sequence_length, 1, H),
                                                                    requires_grad=True)
        config = GPT2Config()
        config.hidden_size = H
        config.n_layer = transformer_layers
        config.vocab_size = 1
        config.n_head = transformer_heads
        self.attn_implementation = config.attn_implementation = attn_implementation
        self.transformer_backbone = GPT2Model(config)

        MLP_hidden_dim = H // 2
        self.recon_state_decoder = nn.Sequential(Linear(H, MLP_hidden_dim), nn.ReLU(), Linear(MLP_hidden_dim, MLP_hidden_dim),
                                                 nn.ReLU())          # not used (reference: same)
        self.recon_arm_state_decoder = nn.Sequential(Linear(MLP_hidden_dim, 6), nn.Tanh())          # not used
        self.recon_gripper_state_decoder = nn.Sequential(Linear(MLP_hidden_dim, 1), nn.Sigmoid())   # not used

        n_patch = int(calvin_input_image_size ** 2 / patch_size / patch_size)
        proj_in = int(H / 4) if self.share_query else H
        if self.obs_pred:
            self.IMAGE_DECODER_hidden_dim = H
            self.NUM_MASK_TOKEN = n_patch * self.pred_num
            self.PATCH_SIZE = patch_size
            self.mask_token = nn.Parameter(torch.zeros(1, 1, H))
            self.image_decoder_obs_pred_projector = Linear(proj_in, H)
            self.image_decoder_position_embedding = nn.Parameter(
                torch.zeros(1, self.NUM_OBS_TOKEN_PER_IMAGE + self.NUM_MASK_TOKEN, H), requires_grad=False)
            self.image_decoder = _decoder(H)
            self.image_decoder_norm = LayerNorm(H)
            self.image_decoder_pred = Linear(H, self.PATCH_SIZE ** 2 * 3)
        if self.depth_pred:
            self.use_dpt_head = use_dpt_head
            self.DEPTH_DECODER_hidden_dim = H
            self.NUM_DEPTH_MASK_TOKEN = n_patch * self.pred_num
            self.PATCH_SIZE = patch_size
            self.depth_decoder_obs_pred_projector = Linear(proj_in, H)
            self.depth_decoder = _decoder(H)
            self.depth_decoder_norm = LayerNorm(H)
            self.depth_decoder_pred = Linear(H, self.PATCH_SIZE ** 2 * 1)
            self.depth_loss_head = SiLogLoss()
            self.depth_mask_token = nn.Parameter(torch.zeros(1, 1, H))
            self.depth_decoder_position_embedding = nn.Parameter(
                torch.zeros(1, self.NUM_OBS_TOKEN_PER_DEPTH + self.NUM_DEPTH_MASK_TOKEN, H), requires_grad=False)
        if self.dino_feat_pred:
            self.DINO_DECODER_hidden_dim = H
            self.NUM_DINO_MASK_TOKEN = 256 * self.pred_num
            self.dino_decoder_obs_pred_projector = Linear(proj_in, H)
            self.dino_feat_decoder = _decoder(H)
            self.dino_decoder_norm = LayerNorm(H)
            self.dino_decoder_pred = Linear(H, 768)
            self.dino_loss_head = SiLogLoss()
            self.dino_mask_token = nn.Parameter(torch.zeros(1, 1, H))
            self.dino_decoder_position_embedding = nn.Parameter(
                torch.zeros(1, self.NUM_OBS_TOKEN_PER_DINO + self.NUM_DINO_MASK_TOKEN, H), requires_grad=False)
        if self.sam_feat_pred:
            self.SAM_DECODER_hidden_dim = H
            self.NUM_SAM_MASK_TOKEN = 256 * self.pred_num
            self.sam_decoder_obs_pred_projector = Linear(proj_in, H)
            self.sam_feat_decoder = _decoder(H)
            self.sam_decoder_norm = LayerNorm(H)
            self.sam_decoder_pred = Linear(H, 256)
            self.sam_mask_token = nn.Parameter(torch.zeros(1, 1, H))
            self.sam_decoder_position_embedding = nn.Parameter(
                torch.zeros(1, self.NUM_OBS_TOKEN_PER_SAM + self.NUM_SAM_MASK_TOKEN, H), requires_grad=False)
        if self.trajectory_pred:
            self.use_traj_query = use_trajectory_query
            self.track_label_patch_size = track_label_patch_size
            self.TRAJ_DECODER_hidden_dim = H
            if no_unshuffle:
                self.NUM_TRAJ_MASK_TOKEN = 784 * self.pred_num
                self.traj_decoder_pred = Linear(H, 2)
            else:
                self.NUM_TRAJ_MASK_TOKEN = n_patch * self.pred_num
                self.traj_decoder_pred = Linear(H, (patch_size // track_label_patch_size) ** 2 * 2)
            self.PATCH_SIZE = patch_size
            self.traj_decoder_obs_pred_projector = Linear(H, H)
            self.traj_decoder = _decoder(H)
            self.traj_decoder_norm = LayerNorm(H)
            self.traj_mask_token = nn.Parameter(torch.zeros(1, 1, H))
            torch.nn.init.normal_(self.traj_mask_token, std=.02)
            self.traj_decoder_position_embedding = nn.Parameter(
                torch.zeros(1, self.NUM_OBS_TOKEN_PER_TRAJ + self.NUM_TRAJ_MASK_TOKEN, H), requires_grad=False)

        self.use_dit_head = use_dit_head
        if self.use_dit_head:
            cls = ActionModel if not use_fm else ActionModelFM
            self.action_model = cls(model_type=self.dit_type, token_size=H, in_channels=7,
                                    future_action_window_size=self.action_pred_steps - 1,
                                    past_action_window_size=0).to(torch.float32)
        else:
            self.action_decoder = nn.Sequential(Linear(H, MLP_hidden_dim), nn.ReLU(), Linear(MLP_hidden_dim, MLP_hidden_dim),
                                                nn.ReLU())
            self.arm_action_decoder = nn.Sequential(Linear(MLP_hidden_dim, 6), nn.Tanh())
            self.gripper_action_decoder = nn.Sequential(Linear(MLP_hidden_dim, 1), nn.Sigmoid())
        self.initialize_weights()

        if self.vit_checkpoint_path is not None:
            vit_checkpoint = torch.load(self.vit_checkpoint_path, map_location='cpu')
            self.vision_encoder.load_state_dict(vit_checkpoint['model'], strict=False)
        clip_path = "checkpoints/clip/ViT-B-32.pt"
        self.clip_model, self.image_processor = clip_text.load(clip_path if os.path.exists(clip_path) else "ViT-B/32",
                                                               device=clip_device)

    # ------------------------------------------------------------------------------------------------
    def _num_query_tokens(self):
        if self.share_query:
            return self.NUM_OBS_TOKEN
        return self.NUM_OBS_TOKEN + self.NUM_DEPTH_TOKEN + self.NUM_TRAJ_TOKEN + self.NUM_DINO_TOKEN + self.NUM_SAM_TOKEN

    def _make_mask(self):
        nq = self._num_query_tokens()
        return generate_attention_mask(
            K=self.sequence_length, num_A=1 + 1 + self.NUM_RESAMPLER_QUERY * 2 + 1 * 2, num_B=nq + self.action_pred_steps,
            atten_goal=self.atten_goal, atten_goal_state=self.atten_goal_state, atten_only_obs=self.atten_only_obs,
            attn_robot_proprio_state=self.attn_robot_proprio_state, mask_l_obs_ratio=self.mask_l_obs_ratio,
            num_obs_token=nq, action_pred_steps=self.action_pred_steps)

    def _mask_rule(self):
        nq = self._num_query_tokens()
        return dict(K=self.sequence_length, num_A=1 + 1 + self.NUM_RESAMPLER_QUERY * 2 + 1 * 2, num_B=nq + self.action_pred_steps,
                    atten_goal=self.atten_goal, atten_goal_state=self.atten_goal_state, atten_only_obs=self.atten_only_obs,
                    attn_robot_proprio_state=self.attn_robot_proprio_state, num_obs_token=nq,
                    action_pred_steps=self.action_pred_steps)

    def _pretrain_mask_tables(self, device):
        """Pretrain phase (dreamvla_model.py:610-628): a new mask every training step.  The random part -- which obs-token
        columns `mask_l_obs_ratio` hides -- is drawn on the host with numpy's RNG exactly as generate_attention_mask draws
        it; the kernels' tables are then computed on the device from the rule (ops.build_mask_tables_device): no (L, L)
        tensor, no upload of one, no device -> host copy.  `self.attention_mask` (a state_dict entry) keeps its value."""
        r = self._mask_rule()
        drop = ops.draw_mask_drop(r["K"], r["num_obs_token"], r["action_pred_steps"], r["atten_only_obs"], self.mask_l_obs_ratio)
        return ops.build_mask_tables_device(device, drop=drop, **r)

    def _fill_decoder_pos(self, param, n_obs, n_mask):
        obs = get_2d_sincos_pos_embed(self.hidden_dim, int(n_obs ** .5), cls_token=False)
        msk = get_2d_sincos_pos_embed(self.hidden_dim, int(n_mask ** .5), cls_token=False)
        param.data.copy_(torch.from_numpy(np.concatenate((obs, msk), axis=0)).float().unsqueeze(0))

    def initialize_weights(self):
        """dreamvla_model.py:543-591 (note: sam_decoder_position_embedding / sam_mask_token stay zero there too)."""
        if self.obs_pred:
            self._fill_decoder_pos(self.image_decoder_position_embedding, self.NUM_OBS_TOKEN_PER_IMAGE, self.NUM_MASK_TOKEN)
            torch.nn.init.normal_(self.mask_token, std=.02)
        if self.depth_pred:
            self._fill_decoder_pos(self.depth_decoder_position_embedding, self.NUM_OBS_TOKEN_PER_DEPTH, self.NUM_DEPTH_MASK_TOKEN)
            torch.nn.init.normal_(self.depth_mask_token, std=.02)
        if self.dino_feat_pred:
            self._fill_decoder_pos(self.dino_decoder_position_embedding, self.NUM_OBS_TOKEN_PER_DINO, self.NUM_DINO_MASK_TOKEN)
            torch.nn.init.normal_(self.dino_mask_token, std=.02)
        if self.trajectory_pred:
            self._fill_decoder_pos(self.traj_decoder_position_embedding, self.NUM_OBS_TOKEN_PER_TRAJ, self.NUM_TRAJ_MASK_TOKEN)
            torch.nn.init.normal_(self.traj_mask_token, std=.02)
        torch.nn.init.normal_(self.transformer_backbone_position_embedding, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            if m.weight is not None:
                nn.init.constant_(m.weight, 1.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def _init_model_type(self):
        self.vision_encoder_type = next(self.vision_encoder.parameters()).type()
        self.perceiver_resampler_type = next(self.perceiver_resampler.parameters()).type()
        self.transformer_backbone_type = next(self.transformer_backbone.parameters()).type()
        if not self.use_dit_head:
            self.action_decoder_type = next(self.action_decoder.parameters()).type()

    # ------------------------------------------------------------------------------------------------
    def _dream_head(self, feat, n2, n_q, n_mask, projector, mask_token, pos, decoder, norm, pred, act="none"):
        """feat: (B, S, n_tok, Hin) slice of the trunk output -> pred (n2*n_mask, out).  dreamvla_model.py:793-911."""
        Hd = self.hidden_dim
        emb = projector(feat.reshape(-1, feat.shape[-1])).view(n2, n_q, Hd)
        pos = pos.to(emb.dtype)
        # the n_mask trailing tokens (mask_token + position) are the same in every sequence: the first decoder block
        # computes their LayerNorm / qkv projection once (nn.Block.forward_shared_suffix)
        x = decoder[0].forward_shared_suffix(emb + pos[:, :n_q], (mask_token.to(emb.dtype) + pos[:, n_q:])[0], n2)
        for blk in list(decoder)[1:]:
            x = blk(x)
        # the prediction layer sees the mask tokens only (dreamvla_model.py:812-816): normalised straight out of the (n2, n_q + n_mask, H)
        # stream -- the strided slice is never copied, and its backward writes the whole stream's gradient (zeros for the query tokens)
        if os.environ.get("DVLA_LN_ROWS") == "0":      # (same-box A/B of the row-group LayerNorm: the copy of the strided slice)
            x = norm(x[:, -n_mask:, :].reshape(-1, Hd))
        else:
            x = norm.last_tokens(x, n_mask)
        return pred(x, act=act)

    def forward(self, image_primary, image_wrist, state, text_token, action=None, track_infos=None, action_label=None,
                mode='train'):
        self._step_mask_tables = None
        if self.training and self.phase == "pretrain":
            self._step_mask_tables = self._pretrain_mask_tables(self.attention_mask.device)
        parts = self.encode_frames(image_primary, image_wrist, state, text_token)
        return self.decode_tokens(parts, action_label=action_label, mode=mode)

    def _text_share_begin(self, text_token):
        """Are the S token rows of every sample equal (utils/train_utils.py:124 repeats the instruction over the window)?
        The answer is needed on the HOST (it changes the text tower's batch size), so it costs a device -> host read.  This
        only LAUNCHES the comparison and the copy of its one-byte verdict into pinned memory; `_text_share_end` waits for it
        after the state / vision path of the SAME forward has been enqueued -- the host blocks until the device reaches the
        comparison, the device then still has the whole ViT + resampler queued behind it and never idles.  The verdict is
        exact for every forward: no assumption carried over from an earlier batch, nothing to raise a step late (round-2
        ADVICE), identical on every rank of a data-parallel job.  Returns the pending (flag, event) or None (decided: not
        shared -- S == 1, switched off, CPU tokens, or inside a stream capture where a host read is illegal)."""
        S = text_token.shape[1]
        if S <= 1 or not self.share_text_over_time or not text_token.is_cuda or torch.cuda.is_current_stream_capturing():
            return None
        flag = torch.empty(1, dtype=torch.bool, pin_memory=True)
        flag.copy_((text_token == text_token[:, :1]).all().reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return flag, ev

    @staticmethod
    def _text_share_end(pending):
        if pending is None:
            return False
        flag, ev = pending
        ev.synchronize()
        return bool(flag.item())

    def encode_text(self, text_token):
        """(B, S, 77) int64 -> the text token of every frame (B, S, 1, H): frozen CLIP text tower + projector (643-653)"""
        B, S = text_token.shape[:2]
        with torch.no_grad():
            text_feature = self.clip_model.encode_text(text_token.flatten(0, 1))
        return self.text_projector(text_feature.to(torch.bfloat16)).view(B, S, -1, self.hidden_dim)

    def encode_frames(self, image_primary, image_wrist, state, text_token, text_embedding=None):
        """Conditioning tokens of every frame, as the list [text (B,S,1,H), state (B,S,1,H), primary image (B,S,nq,H),
        wrist image (B,S,nq,H), cls primary (B,S,1,H), cls wrist (B,S,1,H)]  (dreamvla_model.py:643-737).  Each frame
        is encoded independently of every other frame and of its position in the window (the window position embedding
        is added in `decode_tokens`), which is what lets dreamvla_amd.rollout.RolloutEngine keep the tokens of the
        frames it has already seen and encode only the newest one per control step."""
        B, S, _ = state.shape
        H = self.hidden_dim
        wdt = torch.bfloat16     # compute dtype: fp32 parameters are masters, the kernels run on bf16 shadows (ops.shadow)

        # `text_embedding` (B, S, 1, H), if given, is used instead of running the text tower on `text_token` (the rollout engine
        # keeps it across control steps while the instruction does not change)
        share_pending = None if text_embedding is not None else self._text_share_begin(text_token)   # verdict read below

        # state: arm Linear(6,H) | gripper one-hot(2) -> Linear(2,H) -> cat -> Linear(2H,H)   (656-664)
        st = state.flatten(0, 1).to(wdt)
        arm_state_feature = self.arm_state_encoder(st[:, :6])
        if not self.gripper_width:
            idx = torch.where(st[:, 6:].flatten() < 1, 0, 1)
            gripper_in = torch.nn.functional.one_hot(idx, num_classes=2).to(wdt)
        else:
            gripper_in = st[:, 6:]
        gripper_state_feature = self.gripper_state_encoder(gripper_in)
        state_embedding = self.state_projector(torch.cat((arm_state_feature, gripper_state_feature), dim=1)).view(B, S, -1, H)

        # vision: frozen MAE ViT-B/16 over both views as one batch                        (667-673, 705-713)
        vdt = torch.bfloat16
        n = B * S
        with torch.no_grad():
            if image_primary.dtype == vdt and image_wrist.dtype == vdt and image_primary[0, 0].numel() % 8 == 0:
                # both views as one batch: one 16-byte-vector copy pass (torch.cat moved these 135 MB at ~0.9 TB/s)
                frame = tuple(image_primary.shape[2:])
                imgs = ops.assemble_tokens([image_primary.reshape(1, 1, 1, -1), image_wrist.reshape(1, 1, 1, -1)]).view(2 * n, *frame)
            else:
                imgs = torch.cat((image_primary.flatten(0, 1), image_wrist.flatten(0, 1)), dim=0).to(vdt)
            feats, _, _ = self.vision_encoder.forward_encoder(imgs, mask_ratio=0.0)       # (2n, 197, 768)
        feats = feats.to(torch.bfloat16)
        cls_tok = feats[:, :1, :]
        # one contiguous copy of the patch tokens HERE: the resampler's six norm_media LayerNorms each made their own copy of this
        # strided view (6 x 49 us per step in the torch profile; round 6)
        patches = feats[:, 1:, :].contiguous()
        # perceiver resampler (shared weights, both views batched)                        (716-717)
        lat = self.perceiver_resampler(patches.unsqueeze(1).unsqueeze(1))                 # (2n, 1, nq, 768)
        nq = lat.shape[-2]
        lat = lat.reshape(2, n * nq, self.RESAMPLER_hidden_dim)
        image_primary_embedding = self.image_primary_projector(lat[0]).view(B, S, -1, H)
        image_wrist_embedding = self.image_wrist_projector(lat[1]).view(B, S, -1, H)
        cls2 = cls_tok.reshape(2, n, 768)
        cls_primary = self.cls_token_primary_projector(cls2[0]).view(B, S, -1, H)
        cls_wrist = self.cls_token_wrist_projector(cls2[1]).view(B, S, -1, H)

        # text: frozen CLIP text tower -> Linear(512, H)                                  (643-653)
        # The training loop feeds the SAME instruction to every frame of a window (`text_tokens.unsqueeze(1).repeat(1,
        # window_size, 1)`, utils/train_utils.py:124): when all S rows of every sample are equal the 12-layer tower runs on B
        # sequences instead of B*S and the result is broadcast.  Decided per forward, exactly (self._text_share_begin/_end).
        if text_embedding is not None:
            pass
        elif self._text_share_end(share_pending):
            with torch.no_grad():
                text_feature = self.clip_model.encode_text(text_token[:, 0].contiguous())
            text_embedding = self.text_projector(text_feature.to(wdt)).view(B, 1, -1, H).expand(B, S, -1, H)
        else:
            with torch.no_grad():
                text_feature = self.clip_model.encode_text(text_token.flatten(0, 1))
            text_embedding = self.text_projector(text_feature.to(wdt)).view(B, S, -1, H)

        return [text_embedding, state_embedding, image_primary_embedding, image_wrist_embedding, cls_primary, cls_wrist]

    def decode_tokens(self, parts, action_label=None, mode='train', test_noise=None, test_select=None):
        """Token assembly with the prediction queries, trunk, dream heads (train) and action head
        (dreamvla_model.py:739-991).  `parts`: the list from `encode_frames`, or one (B, S, 36, H) tensor of them.
        `test_noise` (B*S, action_pred_steps, 7), mode='test' with the DiT head only: the sampler's start noise as an INPUT
        (the reference draws it with torch.randn inside forward, dreamvla_model.py:941) -- what lets a hipGraph-captured
        decode take fresh noise per replay and lets a parity test feed the reference's own draw.
        `test_select` (B,) int64 on the device, mode='test' with the DiT head only: sample the action of ONE window position
        per sequence (the position the evaluation wrapper executes, utils/eval_utils_calvin.py:141-146) instead of all S --
        the sampler's batch elements are independent, so the selected position's samples are those of the full call from the
        same noise rows; the action outputs are then (1, B, steps, .) and `test_noise` is (B, steps, 7)."""
        if torch.is_tensor(parts):
            parts = [parts]
        else:
            parts = list(parts)
        B, S = parts[0].shape[:2]
        n = B * S
        H = self.hidden_dim
        wdt = torch.bfloat16     # compute dtype: fp32 parameters are masters, the kernels run on bf16 shadows (ops.shadow)
        image_pred = depth_pred = traj_pred = dino_pred = sam_pred = None
        arm_pred_action = gripper_pred_action = None
        arm_pred_state = gripper_pred_state = None
        loss_arm_action = None

        # token assembly                                                                     (739-759)
        pred_token_start_idx = sum(p.shape[2] for p in parts)
        if self.obs_pred:
            parts.append(self.obs_tokens.to(wdt).expand(B, S, -1, -1))
        if not self.share_query:
            if self.depth_pred:
                parts.append(self.depth_tokens.to(wdt).expand(B, S, -1, -1))
            if self.dino_feat_pred:
                parts.append(self.dino_feat_tokens.to(wdt).expand(B, S, -1, -1))
            if self.sam_feat_pred:
                parts.append(self.sam_feat_tokens.to(wdt).expand(B, S, -1, -1))
            if self.trajectory_pred:
                parts.append(self.trajectory_tokens.to(wdt).expand(B, S, -1, -1))
        if self.action_pred_steps > 0:
            parts.append(self.action_pred_token.to(wdt).expand(B, S, -1, -1))
        # one gather-write kernel: cat along the token axis + the window-position embedding (ops.assemble_tokens)
        transformer_input = ops.assemble_tokens(parts, self.transformer_backbone_position_embedding.to(wdt))
        transformer_input = transformer_input.flatten(1, 2)

        # trunk                                                                              (762-790)
        transformer_input = self.embedding_layer_norm(transformer_input)
        transformer_output = self.transformer_backbone(inputs_embeds=transformer_input, attention_mask=self.attention_mask,
                                                       mask_tables=getattr(self, "_step_mask_tables", None))
        transformer_output = transformer_output.view(B, S, -1, H)

        # dream heads (training only)                                                        (792-911)
        q0 = pred_token_start_idx
        cur = 0
        n2 = n * 2
        if self.obs_pred and mode == 'train':
            if self.share_query:
                feat = transformer_output[:, :, q0:q0 + self.NUM_OBS_TOKEN, :int(H / 4)]
                cur = 0
            else:
                feat = transformer_output[:, :, q0:q0 + self.NUM_OBS_TOKEN, :]
                cur += self.NUM_OBS_TOKEN
            p = self._dream_head(feat, n2, self.NUM_OBS_TOKEN_PER_IMAGE, self.NUM_MASK_TOKEN,
                                 self.image_decoder_obs_pred_projector, self.mask_token,
                                 self.image_decoder_position_embedding, self.image_decoder, self.image_decoder_norm,
                                 self.image_decoder_pred)
            image_pred = p.view(n, self.NUM_OBS_TOKEN // self.NUM_OBS_TOKEN_PER_IMAGE, self.pred_num,
                                self.NUM_MASK_TOKEN // self.pred_num, -1)
        if self.depth_pred and mode == 'train':
            if self.share_query:
                feat = transformer_output[:, :, q0 + cur:q0 + cur + self.NUM_DEPTH_TOKEN, int(H / 4):int(H / 2)]
                cur = 0
            else:
                feat = transformer_output[:, :, q0 + cur:q0 + cur + self.NUM_DEPTH_TOKEN, :]
                cur += self.NUM_DEPTH_TOKEN
            p = self._dream_head(feat, n2, self.NUM_OBS_TOKEN_PER_DEPTH, self.NUM_DEPTH_MASK_TOKEN,
                                 self.depth_decoder_obs_pred_projector, self.depth_mask_token,
                                 self.depth_decoder_position_embedding, self.depth_decoder, self.depth_decoder_norm,
                                 self.depth_decoder_pred, act="relu")
            depth_pred = p.view(n, self.NUM_DEPTH_TOKEN // self.NUM_OBS_TOKEN_PER_DEPTH, self.pred_num,
                                self.NUM_DEPTH_MASK_TOKEN // self.pred_num, -1)
        if self.dino_feat_pred and mode == 'train':
            if self.share_query:
                feat = transformer_output[:, :, q0 + cur:q0 + cur + self.NUM_DINO_TOKEN, int(H / 2):int(H * 3 / 4)]
                cur = 0
            else:
                feat = transformer_output[:, :, q0 + cur:q0 + cur + self.NUM_DINO_TOKEN, :]
                cur += self.NUM_DINO_TOKEN
            p = self._dream_head(feat, n2, self.NUM_OBS_TOKEN_PER_DINO, self.NUM_DINO_MASK_TOKEN,
                                 self.dino_decoder_obs_pred_projector, self.dino_mask_token,
                                 self.dino_decoder_position_embedding, self.dino_feat_decoder, self.dino_decoder_norm,
                                 self.dino_decoder_pred)
            dino_pred = p.view(n, self.NUM_DINO_TOKEN // self.NUM_OBS_TOKEN_PER_DINO, self.pred_num,
                               self.NUM_DINO_MASK_TOKEN // self.pred_num, -1)
        if self.sam_feat_pred and mode == 'train':
            if self.share_query:
                feat = transformer_output[:, :, q0 + cur:q0 + cur + self.NUM_SAM_TOKEN, int(H * 3 / 4):int(H)]
                cur = 0
            else:
                feat = transformer_output[:, :, q0 + cur:q0 + cur + self.NUM_SAM_TOKEN, :]
                cur += self.NUM_SAM_TOKEN
            p = self._dream_head(feat, n2, self.NUM_OBS_TOKEN_PER_SAM, self.NUM_SAM_MASK_TOKEN,
                                 self.sam_decoder_obs_pred_projector, self.sam_mask_token,
                                 self.sam_decoder_position_embedding, self.sam_feat_decoder, self.sam_decoder_norm,
                                 self.sam_decoder_pred)
            sam_pred = p.view(n, self.NUM_SAM_TOKEN // self.NUM_OBS_TOKEN_PER_SAM, self.pred_num,
                              self.NUM_SAM_MASK_TOKEN // self.pred_num, -1)
        if self.trajectory_pred and mode == 'train':
            feat = transformer_output[:, :, q0 + cur:q0 + cur + self.NUM_TRAJ_TOKEN, :]
            ntv = self.NUM_TRAJ_TOKEN // self.NUM_OBS_TOKEN_PER_TRAJ
            p = self._dream_head(feat, n * ntv, self.NUM_OBS_TOKEN_PER_TRAJ, self.NUM_TRAJ_MASK_TOKEN,
                                 self.traj_decoder_obs_pred_projector, self.traj_mask_token,
                                 self.traj_decoder_position_embedding, self.traj_decoder, self.traj_decoder_norm,
                                 self.traj_decoder_pred)
            traj_pred = p.view(n, ntv, self.pred_num, self.NUM_TRAJ_MASK_TOKEN // self.pred_num, -1)
            cur += self.NUM_TRAJ_TOKEN

        # action head                                                                        (915-987)
        if self.action_pred_steps > 0:
            nqt = self._num_query_tokens()
            action_pred_feature = transformer_output[:, :, q0 + nqt:q0 + nqt + self.action_pred_steps, :]
            if not self.use_dit_head:
                h1 = self.action_decoder[0](action_pred_feature, act="relu")
                h2 = self.action_decoder[2](h1, act="relu")
                arm_pred_action = self.arm_action_decoder[0](h2, act="tanh")
                gripper_pred_action = self.gripper_action_decoder[0](h2, act="sigmoid")
            elif mode == 'train':
                feat = action_pred_feature[:, :self.sequence_length - self.atten_goal].flatten(0, 1)
                labels = action_label.flatten(0, 1)
                r = 8   # repeated_diffusion_steps
                arm_pred_action = self.action_model.loss(labels.repeat(r, 1, 1), feat.repeat(r, 1, 1))
                gripper_pred_action = arm_pred_action
            elif mode == 'test':
                if test_select is None:
                    bs = n
                    cond = action_pred_feature.flatten(0, 1)
                else:
                    if tuple(test_select.shape) != (B,):
                        raise ValueError(f"test_select {tuple(test_select.shape)}: expected ({B},)")
                    bs = B
                    cond = action_pred_feature[torch.arange(B, device=test_select.device), test_select]
                cfg_scale = 1.5
                if test_noise is None:
                    noise = torch.randn(bs, self.action_pred_steps, self.action_model.in_channels,
                                        device=cond.device).to(cond.dtype)
                else:
                    if tuple(test_noise.shape) != (bs, self.action_pred_steps, self.action_model.in_channels):
                        raise ValueError(f"test_noise {tuple(test_noise.shape)}: expected "
                                         f"{(bs, self.action_pred_steps, self.action_model.in_channels)}")
                    noise = test_noise.to(cond.device, cond.dtype)
                noise = torch.cat([noise, noise], 0)
                uncondition = self.action_model.net.z_embedder.uncondition.to(cond.dtype)
                uncondition = uncondition.unsqueeze(0).expand(bs, self.action_pred_steps, -1)
                z = torch.cat([cond, uncondition], 0)
                if self.action_model.ddim_diffusion is None:
                    self.action_model.create_ddim(ddim_step=10)
                if getattr(self, "fast_sampler", True) and hasattr(self.action_model, "sample_ddim_cfg"):
                    # the same sampler with the step-invariant work hoisted and the per-step algebra in one kernel
                    # (ActionModel.sample_ddim_cfg; `model.fast_sampler = False` runs the operation-by-operation loop below)
                    samples = self.action_model.sample_ddim_cfg(cond, noise[:bs].float(), cfg_scale)
                else:
                    samples = self.action_model.ddim_diffusion.ddim_sample_loop(
                        self.action_model.net.forward_with_cfg, noise.shape, noise, clip_denoised=False,
                        model_kwargs=dict(z=z, cfg_scale=cfg_scale), progress=False, device=cond.device, eta=0.0,
                        start_noise=None if test_noise is None else noise)
                    samples, _ = samples.chunk(2, dim=0)
                arm_pred_action, gripper_pred_action = samples.unsqueeze(0)[..., :6], samples.unsqueeze(0)[..., 6:]

        return (arm_pred_action, gripper_pred_action, image_pred, arm_pred_state, gripper_pred_state, loss_arm_action,
                depth_pred, traj_pred, dino_pred, sam_pred)
