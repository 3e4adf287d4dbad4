"""Host logic (CPU): attention-mask generator bit-exactness vs the reference + the kernel's mask tables."""
import itertools

import numpy as np
import pytest
import torch

from dreamvla_amd.dreamvla_model import generate_attention_mask
from oracle import ref_loader

COMBOS = [dict(K=K, num_A=36, num_B=nq + aps, atten_goal=ag, atten_goal_state=ags, atten_only_obs=aoo,
               attn_robot_proprio_state=arps, mask_l_obs_ratio=ml, num_obs_token=nq, action_pred_steps=aps)
          for K, nq, aps, ag, ags, aoo, arps, ml in [
              (7, 18, 3, 0, False, False, False, 0.0), (7, 54, 3, 0, False, False, False, 0.0),
              (10, 72, 3, 0, False, False, False, 0.0), (14, 18, 3, 0, False, True, True, 0.5),
              (7, 18, 3, 1, True, True, False, 0.3), (5, 0, 3, 0, False, False, False, 0.0),
              (6, 18, 0, 0, False, False, False, 0.0), (4, 36, 1, 2, True, False, True, 0.0)]]


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
@pytest.mark.parametrize("kw", COMBOS)
def test_mask_bit_exact_vs_reference(kw):
    ref = ref_loader.ref_module("models.dreamvla_model").generate_attention_mask
    np.random.seed(123)
    a = ref(**kw)
    np.random.seed(123)
    b = generate_attention_mask(**kw)
    assert a.shape == b.shape and torch.equal(a, b)


def test_mask_golden():
    import os
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "masks.pt"))
    for kw, packed in zip(COMBOS, g["packed"]):
        np.random.seed(123)
        m = generate_attention_mask(**kw)
        assert torch.equal(torch.from_numpy(np.packbits((m == 0).numpy())), packed)


@pytest.mark.parametrize("compact", [False, True])
def test_mask_tables(compact):
    from dreamvla_amd.ops import build_mask_tables
    m = generate_attention_mask(7, 36, 57, 0, False, False, False, 0.0, 54, 3)
    mt = build_mask_tables(m, device="cpu", compact_keys=compact)
    vis = (m == 0)
    cols = np.arange(m.shape[1]) if mt.key_index is None else mt.key_index.numpy()
    if compact:
        assert mt.Lk < m.shape[1] and mt.Lk == int(vis.any(0).sum())
    bq = mt.bits_q.numpy().view(np.uint32)
    bk = mt.bits_k.numpy().view(np.uint32)
    tm = mt.tile_map.numpy()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        i, j = int(rng.integers(m.shape[0])), int(rng.integers(mt.Lk))
        expect = bool(vis[i, cols[j]])
        assert bool((bq[i, j // 32] >> (j % 32)) & 1) == expect
        assert bool((bk[j, i // 32] >> (i % 32)) & 1) == expect
    V = vis.numpy()[:, cols]
    for qt in range(tm.shape[0]):
        for kt in range(tm.shape[1]):
            blk = V[qt * 32:(qt + 1) * 32, kt * 32:(kt + 1) * 32]
            want = 1 if blk.all() else (0 if not blk.any() else 2)
            assert tm[qt, kt] == want
    assert abs(mt.visible_fraction - V.mean()) < 1e-6
