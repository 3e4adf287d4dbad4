"""Host logic (CPU): attention-mask generator bit-exactness vs the reference + the kernel's mask tables."""

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


def _rule_kw(kw):
    return dict(K=kw["K"], num_A=kw["num_A"], num_B=kw["num_B"], atten_goal=kw["atten_goal"], atten_goal_state=kw["atten_goal_state"],
                atten_only_obs=kw["atten_only_obs"], attn_robot_proprio_state=kw["attn_robot_proprio_state"],
                num_obs_token=kw["num_obs_token"], action_pred_steps=kw["action_pred_steps"])


@pytest.mark.parametrize("kw", COMBOS)
def test_mask_rule_mirror_bit_exact(kw):
    """the closed-form visibility rule that csrc/masks.hip evaluates on the device (its host mirror
    ops.mask_rule_visible) against generate_attention_mask for every flag combination, with the mask_l_obs_ratio columns
    drawn from numpy's stream in the reference's order (ops.draw_mask_drop) -- and the stream must end in the same state."""
    from dreamvla_amd import ops
    np.random.seed(123)
    m = generate_attention_mask(**kw)
    after_ref = np.random.random()
    np.random.seed(123)
    drop = ops.draw_mask_drop(kw["K"], kw["num_obs_token"], kw["action_pred_steps"], kw["atten_only_obs"], kw["mask_l_obs_ratio"])
    after_ours = np.random.random()
    assert after_ref == after_ours
    vis = ops.mask_rule_visible(drop=drop, **_rule_kw(kw))
    assert vis.shape == tuple(m.shape) and np.array_equal(vis, (m == 0).numpy())


def test_key_order_groups_columns_by_audience():
    """the gather list orders kept keys by audience size: for the trunk mask the leading columns of all steps come first
    (block-causal prefix), the obs columns last -- far fewer non-empty and far fewer mixed 32 x 32 tiles"""
    from dreamvla_amd.ops import build_mask_tables
    m = generate_attention_mask(7, 36, 57, 0, False, False, False, 0.0, 54, 3)
    mt = build_mask_tables(m, device="cpu")
    ki = mt.key_index.numpy()
    assert np.array_equal(ki[:7 * 36], (np.arange(7)[:, None] * 93 + np.arange(36)[None, :]).reshape(-1))
    assert np.array_equal(ki[7 * 36:], (np.arange(7)[:, None] * 93 + 36 + np.arange(54)[None, :]).reshape(-1))
    tm = mt.tile_map.numpy()
    asc = build_mask_tables(m, device="cpu", key_order=np.sort(ki)).tile_map.numpy()
    assert (tm != 0).sum() <= 130 and (asc != 0).sum() >= 180 and (tm == 2).sum() <= 45 and (asc == 2).sum() >= 160
    with pytest.raises(ValueError):
        build_mask_tables(m, device="cpu", key_order=ki[:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("kw", COMBOS)
def test_mask_tables_on_device_bit_identical(kw):
    """dvla_mask_tables (device, from the rule) == build_mask_tables(generate_attention_mask(...)) (host, from the (L, L)
    tensor): the device's key list names exactly the visible columns (leading columns first, obs columns last -- for the
    default trunk flags that IS the host's own order), and both bit tables and the tile map agree bit for bit for that order."""
    from dreamvla_amd import ops
    np.random.seed(123)
    mask = generate_attention_mask(**kw)
    own = ops.build_mask_tables(mask, device="cpu")
    np.random.seed(123)
    drop = ops.draw_mask_drop(kw["K"], kw["num_obs_token"], kw["action_pred_steps"], kw["atten_only_obs"], kw["mask_l_obs_ratio"])
    got = ops.build_mask_tables_device("cuda", drop=drop, **_rule_kw(kw))
    torch.cuda.synchronize()
    gk = torch.arange(got.Lk_full, dtype=torch.int32) if got.key_index is None else got.key_index.cpu()
    want = ops.build_mask_tables(mask, device="cpu", key_order=gk.numpy())     # raises unless gk names the visible columns
    assert (got.Lq, got.Lk_full, got.Lk) == (want.Lq, want.Lk_full, want.Lk)
    blk = kw["num_A"] + kw["num_B"]
    lead = gk[:kw["K"] * kw["num_A"]].numpy()
    assert np.array_equal(lead, (np.arange(kw["K"])[:, None] * blk + np.arange(kw["num_A"])[None, :]).reshape(-1))
    if not kw["atten_only_obs"] and not kw["atten_goal"]:
        ok = torch.arange(own.Lk_full, dtype=torch.int32) if own.key_index is None else own.key_index
        assert torch.equal(gk, ok)
    assert torch.equal(got.bits_q.cpu(), want.bits_q) and torch.equal(got.bits_k.cpu(), want.bits_k)
    assert torch.equal(got.tile_map.cpu(), want.tile_map)


def test_mask_table_cache_is_per_tensor_object():
    """round-1 ADVICE: the cache must not serve the tables of a freed mask whose storage address was recycled"""
    from dreamvla_amd import ops
    a = generate_attention_mask(3, 36, 21, 0, False, False, False, 0.0, 18, 3)
    ta = ops.mask_tables_for(a)
    assert ops.mask_tables_for(a) is ta
    b = a.clone()
    b[:, :5] = -float("inf")
    tb = ops.mask_tables_for(b)
    assert tb is not ta and tb.Lk == ta.Lk - 5
    key_a = next(k for k, e in ops._MASK_CACHE.items() if e[4] is ta)
    del a
    ops._MASK_CACHE[key_a] = (lambda: None, 0, 0, torch.device("cpu"), ta)   # a dead entry under a recyclable id
    c = generate_attention_mask(3, 36, 21, 0, False, True, True, 0.0, 18, 3)
    tc = ops.mask_tables_for(c)
    assert tc is not ta


def test_mask_table_cache_hits_for_views_and_misses_after_rebind():
    """round-2 ADVICE: (a) the (B,1,L,L) call form hands the trunk `mask[0][0]`, a NEW view object on every forward -- the
    cache is keyed on the base tensor + view geometry, so it hits; (b) a `.data` rebind (module.to(device), flat re-homing)
    changes the storage address without touching the version counter -- the entry must not be served."""
    from dreamvla_amd import ops
    m = generate_attention_mask(3, 36, 21, 0, False, False, False, 0.0, 18, 3)
    m4 = torch.nn.Parameter(m[None, None].expand(2, 1, -1, -1).contiguous(), requires_grad=False)
    t1 = ops.mask_tables_for(m4[0][0])
    t2 = ops.mask_tables_for(m4[0][0])
    assert t2 is t1
    t3 = ops.mask_tables_for(m4[1][0])              # same base, other offset: its own entry (same content here)
    assert t3 is not t1 and torch.equal(t3.bits_q, t1.bits_q)
    other = m4.data.clone()
    other[..., :5] = -float("inf")
    m4.data = other                                 # rebind: version unchanged, address changed
    t4 = ops.mask_tables_for(m4[0][0])
    assert t4 is not t1 and t4.Lk == t1.Lk - 5
    with torch.no_grad():
        m4[..., 5:7] = -float("inf")                # in-place edit on the parameter: version bump
    t5 = ops.mask_tables_for(m4[0][0])
    assert t5 is not t4 and t5.Lk == t4.Lk - 2
