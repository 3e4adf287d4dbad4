"""CPU: the C-ABI shared library loads and exports every symbol include/dvla.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dvla.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvla_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from dreamvla_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dvla.h but not exported"
    assert set(syms) == set(_lib.SYMBOLS), (set(syms) ^ set(_lib.SYMBOLS))
    assert lib.dvla_abi_version() == _lib.ABI_VERSION == 4


def test_structs_match_header_layout():
    """ctypes mirrors of the parameter structs: field order == header order (sizes are implied by the C types)."""
    from dreamvla_amd import _lib
    txt = open(os.path.join(ROOT, "include", "dvla.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for cname, pystruct in (("dvla_gemm_params", _lib.GemmParams), ("dvla_attn_params", _lib.AttnParams)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), txt, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(",")
            for i, part in enumerate(parts):
                names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
        assert names == [f[0] for f in pystruct._fields_], cname


def test_no_cpu_fallback():
    """the product path must fail loudly on CPU tensors (no eager fallback)."""
    import torch
    from dreamvla_amd import ops
    from dreamvla_amd._lib import DvlaError
    x = torch.zeros(4, 8, dtype=torch.bfloat16)
    w = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(DvlaError):
        ops.linear(x, w)
    with pytest.raises(DvlaError):
        ops.layer_norm(x, None, None, 1e-5)
