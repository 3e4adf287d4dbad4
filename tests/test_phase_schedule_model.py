"""CPU: the hazard argument of the phase GEMM kernel's DMA schedule (csrc/gemm_phase.h, file header), checked as a model for every
piece-placement table the header defines -- round 5 moved all eight LDS-DMA pieces of a K-tile to the head of the load segments
and made the placement a compile-time table; the arguments "A pieces are legal anywhere in tile u, B pieces from LOAD2 on" and
"group 1 waits vmcnt(pieces issued so far) in LOAD2" are what keeps a re-placement from racing.

Model (time in barrier intervals; every segment of a group ends with lgkmcnt(0) / the counted vmcnt and one s_barrier):
    group 0 runs segment s of K-tile u in interval 4u + s, group 1 one interval later (s: 0 LOAD1, 1 MFMA1, 2 LOAD2, 3 MFMA2);
    LOAD1(u) reads A image slot u % 3 (a-lo) and B image u & 1 (all B fragments), LOAD2(u) reads A slot u % 3 (a-hi);
    in K-tile u a wave issues A0..A3 of image A(u+2) -> slot (u+2) % 3 and B0..B3 of image B(u+2) -> buffer u & 1, each in the
    segment its table slot belongs to; an issued piece may land at ANY time from its issue to the wave's covering wait;
    covering waits: every wave at the end of MFMA2(u') waits vmcnt(8) -> all pieces of K-tiles < u' of that wave have landed;
    group 1 also in LOAD2(u'), behind the pieces of slots <= 8, waits vmcnt(pieces_up_to(8)) -> the same set;
    a landed piece is visible to other waves from the interval after the barrier that follows the wait.
Checked per table: WAR -- no piece can land in an image before the last read of the image's previous content has retired;
RAW -- every piece is visible when the first read of its image starts; the counted wait of group 1 names exactly the pieces of
the current K-tile that are in flight at that point; pieces of one operand are issued in order (the cursor advances behind the
fourth)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "dreamvla_amd", "csrc", "gemm_phase.h")


def tables():
    src = open(HDR).read()
    body = src[src.index("constexpr PiecePlace piece_place(int pl)"):]
    body = body[:body.index("__host__ __device__ constexpr int pieces_up_to")]
    out = {}
    for m in re.finditer(r"(case\s+(\d+)|default)\s*:\s*return\s*\{\{([0-9,\s]+)\}\}", body):
        key = int(m.group(2)) if m.group(2) else 0
        out[key] = [int(x) for x in m.group(3).split(",")]
    return out


def seg_of_slot(slot):
    return 0 if slot <= 2 else 1 if slot <= 6 else 2 if slot <= 9 else 3


def check(table):
    assert len(table) == 8
    a, b = table[:4], table[4:]
    assert a == sorted(a) and b == sorted(b), "pieces of one operand must be issued in order"
    assert all(0 <= s <= 13 for s in table)
    U = 9                                    # K-tiles modelled (steady state from u = 2 on)
    start = lambda g, u, s: 4 * u + s + g    # interval in which group g runs segment s of K-tile u

    def last_read_A(u):                      # interval of the last read of A(u): group 1's LOAD2(u)
        return max(start(g, u, 2) for g in (0, 1))

    def last_read_B(u):                      # B(u) is read in LOAD1(u) only
        return max(start(g, u, 0) for g in (0, 1))

    def first_read(u):                       # A(u) / B(u): group 0's LOAD1(u)
        return start(0, u, 0)

    for g in (0, 1):
        for u in range(1, U):
            # ---- WAR: the image a piece of K-tile u goes into held A(u-1) (slot (u+2) % 3 == (u-1) % 3) / B(u) (buffer u & 1)
            for i, slot in enumerate(table):
                t_issue = start(g, u, seg_of_slot(slot))
                if i < 4:
                    assert t_issue > last_read_A(u - 1), ("A piece issued before the last read of the slot's previous image", g, u, i)
                else:
                    assert t_issue > last_read_B(u), ("B piece issued before both groups have read B(u)", g, u, i)
            # ---- RAW: visibility of the pieces of K-tile u before LOAD1(u+2) of group 0
            # group 0: wait at the end of MFMA2(u+1), barrier, visible from the next interval
            # group 1: wait in LOAD2(u+1) (its MFMA2(u+1) wait would be one interval late for group 0's read)
            wait_interval = start(0, u + 1, 3) if g == 0 else start(1, u + 1, 2)
            assert wait_interval + 1 <= first_read(u + 2), ("pieces not visible at the first read of their image", g, u)
    # ---- the counted wait of group 1 in LOAD2 sits behind slot 8: the pieces of THIS K-tile issued so far are exactly those with
    # slot <= 8; everything older (the previous K-tile's pieces) must be covered, i.e. N = that count and nothing else is younger
    n_before = sum(1 for s in table if s <= 8)
    assert 0 <= n_before <= 8
    # pieces in slot 9 (behind the wait) and in MFMA2 are issued AFTER the wait: they must not be counted
    assert n_before == len([s for s in table if seg_of_slot(s) < 2 or s in (7, 8)])
    return n_before


def test_every_placement_table_of_the_header_is_hazard_free():
    tabs = tables()
    assert 0 in tabs and len(tabs) >= 3, tabs
    assert tabs[0] == [0, 0, 0, 0, 7, 7, 7, 7]          # production: the head of the two load segments
    counts = {k: check(t) for k, t in tabs.items()}
    assert counts[0] == 8                                # all eight pieces are in flight at group 1's LOAD2 wait: vmcnt(8)
    assert counts[13] == 5                               # the rounds 2-4 placement: A0..A3 and B0 (the old hard-coded vmcnt(5))


def test_the_model_rejects_illegal_placements():
    import pytest
    with pytest.raises(AssertionError):
        check([0, 0, 0, 0, 0, 7, 7, 7])                  # a B piece in LOAD1: group 1 has not read B(u) yet
    with pytest.raises(AssertionError):
        check([0, 0, 0, 0, 4, 7, 7, 7])                  # ... or in MFMA1 (legal for group 1 only: the code is shared)
    with pytest.raises(AssertionError):
        check([2, 0, 0, 0, 7, 7, 7, 7])                  # out of order: the cursor advances behind the fourth piece
