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
