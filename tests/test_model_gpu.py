"""GPU: the HIP modules / whole DreamVLA module against the golden fixtures generated from the real reference and
against the oracle's autograd (gradients).  Tolerances in tests/model_checks.py."""
import pytest

from tests import model_checks as C


def _assert_all(results):
    bad = [r for r in results if not r["ok"]]
    assert not bad, bad


@pytest.mark.gpu
def test_hip_modules_vs_golden():
    _assert_all(C.hip_module_checks())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["A", "B", "E"])
def test_hip_full_model_vs_golden(name):
    _assert_all(C.hip_full_model_checks(name))


@pytest.mark.gpu
def test_hip_whole_model_gradients_vs_oracle():
    _assert_all(C.hip_grad_checks())
