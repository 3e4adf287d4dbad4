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
    cost = {0: 3.0, 4: 2.5, 8: 1.0, 6: 2.0, 7: 2.2, 9: 1.5, 10: 1.6, 2: 4.0}
    best, calls = drive(("k1",), cost)
    assert best == 8
    # every candidate was tried exactly ROUNDS times before the lock, each real call ran exactly one candidate
    assert sorted(calls[:-1]) == sorted(list(GemmTuner.CANDIDATES) * GemmTuner.ROUNDS)
    assert GemmTuner.pick(("k1",)) == (8, None)


def test_tuner_with_unfinished_events_keeps_cycling():
    GemmTuner.reset()
    cost = {0: 1.0, 4: 2.0, 8: 3.0, 6: 4.0, 7: 4.5, 9: 4.7, 10: 4.8, 2: 5.0}
    best, calls = drive(("k2",), cost, lag=3)
    assert best == 0
    assert len(calls) > len(GemmTuner.CANDIDATES)      # had to wait for timings, trying candidates meanwhile
    assert set(calls) <= set(GemmTuner.CANDIDATES)


def test_tuner_keys_are_independent():
    GemmTuner.reset()
    a, _ = drive(("a",), {0: 1, 4: 2, 8: 3, 6: 4, 7: 4.5, 9: 4.7, 10: 4.8, 2: 5})
    b, _ = drive(("b",), {0: 5, 4: 4, 8: 3, 6: 2, 7: 2.5, 9: 2.7, 10: 2.8, 2: 1})
    assert (a, b) == (0, 2)
    GemmTuner.reset()
    assert GemmTuner.table == {} and GemmTuner.trials == {}


def test_plan_round_trip(tmp_path):
    """save_plan / load_plan carry the locked choices to another process (profiler passes of the tuned step): keys survive JSON
    (tuples of ints and bools), unknown keys take the cost model (variant 0) once the plan is frozen, no trials are started"""
    GemmTuner.reset()
    key = (20832, 4096, 1024, 0, 1, 1, 2, 0, True, True, False, False, False, False, False, 0, 0)
    assert len(key) == GemmTuner.KEY_LEN
    old_key = (1024, 4096, 20832, 1, 0, 4, 0, 0, False, False, False, False, False, True, False, 2)   # a round-3 plan entry
    GemmTuner.table[key] = 10
    GemmTuner.table[old_key] = 4
    path = str(tmp_path / "plan.json")
    GemmTuner.save_plan(path)
    GemmTuner.reset()
    try:
        GemmTuner.load_plan(path)
        assert GemmTuner.pick(key) == (10, None)
        assert GemmTuner.pick(old_key + (0,)) == (4, None)        # pre-schedule-tag keys load as default-schedule keys
        assert GemmTuner.pick((1, 2, 3)) == (0, None) and GemmTuner.trials == {}
    finally:
        GemmTuner.frozen = False
        GemmTuner.reset()


def test_no_vendor_library_in_the_product_path():
    """round-1 offered hipBLASLt as a tuner candidate; the product path now runs hand-written kernels only: the tuner has
    no such candidate, ops.gemm never touches the comparator library and libdvla_hip.so does not link it"""
    import inspect
    import os
    import subprocess
    from dreamvla_amd import _lib, ops
    assert not hasattr(GemmTuner, "LIBRARY") and not hasattr(GemmTuner, "library")
    assert all(c in (0, 2, 4, 6, 7, 8, 9, 10) for c in GemmTuner.CANDIDATES)
    assert "load_comparator" not in inspect.getsource(ops) and "dvla_gemm_library" not in inspect.getsource(ops)
    if os.path.exists(_lib.LIB_PATH):
        out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
        assert "hipblaslt" not in out.lower() and "rocblas" not in out.lower()


def test_median_of_rounds_decides():
    """ROUNDS timings per candidate, the median decides (one noisy trial cannot lock a slow configuration)"""
    GemmTuner.reset()
    old = GemmTuner.ROUNDS
    GemmTuner.ROUNDS = 3
    try:
        seq = {0: [1.0, 9.0, 1.0], 4: [2.0, 2.0, 2.0], 8: [3.0, 0.1, 3.0], 6: [4.0, 4.0, 4.0], 7: [4.0, 4.0, 4.0], 9: [4.5, 4.5, 0.2], 10: [4.6, 4.6, 4.6], 2: [5.0, 5.0, 5.0]}
        key = ("med",)
        for _ in range(64):
            v, trial = GemmTuner.pick(key)
            if trial is None:
                break
            n = len(trial["times"][v]) + sum(1 for (pv, _, _) in trial["pending"] if pv == v)
            trial["pending"].append((v, FakeEvent(0.0), FakeEvent(seq[v][min(n, 2)])))
        assert GemmTuner.table[key] == 0
    finally:
        GemmTuner.ROUNDS = old
        GemmTuner.reset()
