"""GPU parity tests proper: every HIP kernel, through the C ABI (ctypes), against oracle/torch_ref.py on the
same seeded inputs.  Tolerances are defined in tests/gpu_checks.py."""
import pytest

from tests import gpu_checks

CASES = gpu_checks.all_checks()


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(CASES)), ids=[f"{i}-{c[0].__name__}" for i, c in enumerate(CASES)])
def test_kernel_parity(idx):
    fn, kw = CASES[idx]
    results = fn(**kw)
    from tests.model_checks import report
    report(results)
    for m in results:
        assert m["ok"], f"{m['name']}: rel_l2={m.get('rel_l2')} max_abs={m.get('max_abs')} tol={m.get('tol')}"


PLAN_FILE, PLAN = gpu_checks.load_gemm_plan()


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(PLAN)), ids=[f"{i}-{'x'.join(str(v) for v in k[:3])}-v{v}" for i, (k, v) in enumerate(PLAN)])
def test_gemm_plan_entry(idx):
    """every problem key of the timed training step (profiles/r0N_gemm_plan.json, written by `bench.py --save-plan` on the GPU)
    at full size under the configuration the tuner locked for it: whole output + sampled block + `dvla_last_gemm_variant`"""
    key, variant = PLAN[idx]
    results = gpu_checks.check_gemm_plan_entry(key, variant)
    from tests.model_checks import report
    report(results)
    for m in results:
        assert m["ok"], f"{PLAN_FILE}: {m['name']}: rel_l2={m.get('rel_l2')} max_abs={m.get('max_abs')} tol={m.get('tol')} ran={m.get('ran')}"


def test_gemm_plan_is_committed():
    """the plan the cases above are generated from exists and has the step's ~120 problem keys (CPU: collection-time guard)"""
    assert PLAN_FILE is not None and len(PLAN) >= 100, (PLAN_FILE, len(PLAN))
