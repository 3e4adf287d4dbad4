"""Host logic of dreamvla_amd.ops.GemmTuner (no GPU: the events are stubs)."""
import pytest

torch = pytest.importorskip("torch")
from dreamvla_amd.ops import GemmTuner  # noqa: E402


class FakeEvent:
    def __init__(self, ms, done=True):
        self.ms, self.done = ms, done

    def query(self):
        return self.done

    def elapsed_time(self, other):
        return other.ms


def drive(key, cost, lag=0):
    """call pick() until the key locks; `cost[v]` is the fake duration of variant v; `lag` trials stay unfinished"""
    calls = []
    for _ in range(64):
        v, trial = GemmTuner.pick(key)
        calls.append(v)
        if trial is None:
            return v, calls
        pend = trial["pending"]
        pend.append((v, FakeEvent(0.0), FakeEvent(cost[v], done=lag == 0)))
        if lag and len(pend) > lag:       # older events complete later, as on a busy stream
            for (_, _, e1) in pend[:-lag]:
                e1.done = True
    raise AssertionError("tuner did not converge")


def test_tuner_locks_fastest_candidate():
    GemmTuner.reset()
    cost = {0: 3.0, 4: 2.5, 5: 1.0, 6: 2.0, 2: 4.0}
    best, calls = drive(("k1",), cost)
    assert best == 5
    # every candidate was tried exactly ROUNDS times before the lock, each real call ran exactly one candidate
    assert sorted(calls[:-1]) == sorted(list(GemmTuner.CANDIDATES) * GemmTuner.ROUNDS)
    assert GemmTuner.pick(("k1",)) == (5, None)


def test_tuner_with_unfinished_events_keeps_cycling():
    GemmTuner.reset()
    cost = {0: 1.0, 4: 2.0, 5: 3.0, 6: 4.0, 2: 5.0}
    best, calls = drive(("k2",), cost, lag=3)
    assert best == 0
    assert len(calls) > len(GemmTuner.CANDIDATES)      # had to wait for timings, trying candidates meanwhile
    assert set(calls) <= set(GemmTuner.CANDIDATES)


def test_tuner_keys_are_independent():
    GemmTuner.reset()
    a, _ = drive(("a",), {0: 1, 4: 2, 5: 3, 6: 4, 2: 5})
    b, _ = drive(("b",), {0: 5, 4: 4, 5: 3, 6: 2, 2: 1})
    assert (a, b) == (0, 2)
    GemmTuner.reset()
    assert GemmTuner.table == {} and GemmTuner.trials == {}


def drive_plain(key, cost):
    calls = []
    for _ in range(64):
        v, trial = GemmTuner.pick(key, plain=True)
        calls.append(v)
        if trial is None:
            return v, calls
        trial["pending"].append((v, FakeEvent(0.0), FakeEvent(cost[v])))
    raise AssertionError("tuner did not converge")


def test_library_candidate_only_for_plain_keys():
    GemmTuner.reset()
    assert GemmTuner.LIBRARY not in GemmTuner.candidates(("fused",), plain=False)
    assert GemmTuner.LIBRARY in GemmTuner.candidates(("plain",), plain=True)
    cost = {0: 3.0, 4: 2.5, 5: 2.0, 6: 2.0, 2: 4.0, GemmTuner.LIBRARY: 1.0}
    best, calls = drive_plain(("plain",), cost)
    assert best == GemmTuner.LIBRARY and calls.count(GemmTuner.LIBRARY) == 2   # one trial + the locked call
    assert GemmTuner.summary() == {"library": 1}
    best, calls = drive(("fused",), cost)
    assert best in (5, 6) and GemmTuner.LIBRARY not in calls


def test_library_ban_removes_the_candidate():
    GemmTuner.reset()
    key = ("plain2",)
    seen = []
    for _ in range(32):
        v, trial = GemmTuner.pick(key, plain=True)
        if trial is None:
            break
        if v == GemmTuner.LIBRARY:
            GemmTuner.ban_library(key)       # what ops.gemm does when dvla_gemm_library_bf16 returns an error
            continue
        seen.append(v)
        trial["pending"].append((v, FakeEvent(0.0), FakeEvent(float(v + 1))))
    assert GemmTuner.table[key] == 0 and GemmTuner.LIBRARY not in seen
    assert GemmTuner.LIBRARY not in GemmTuner.candidates(key, plain=True)
    GemmTuner.reset()
