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
    assert lib.dvla_abi_version() == _lib.ABI_VERSION == 8


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


def test_struct_offsets_match_the_c_compiler(tmp_path):
    """Every ctypes mirror against the header as gcc lays it out: offsetof() of each field and sizeof() of each struct (a field of
    the wrong width, or a missing pad, shifts everything behind it -- the name-order test above cannot see that)."""
    import shutil
    import subprocess
    from dreamvla_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    structs = {"dvla_gemm_params": _lib.GemmParams, "dvla_attn_params": _lib.AttnParams, "dvla_mask_rule": _lib.MaskRule,
               "dvla_token_src": _lib.TokenSrc, "dvla_frame_view": _lib.FrameView, "dvla_dit_sample_params": _lib.DitSampleParams}
    txt = open(os.path.join(ROOT, "include", "dvla.h")).read()
    structs = {c: p for c, p in structs.items() if re.search(r"\}\s*%s\s*;" % c, txt)}
    assert "dvla_gemm_params" in structs and "dvla_attn_params" in structs
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dvla.h"', 'int main(void) {']
    for cname, py in structs.items():
        lines.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for fname, _ in py._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    for line in out.strip().splitlines():
        cname, field, val = line.split()
        py = structs[cname]
        want = ctypes.sizeof(py) if field == "sizeof" else getattr(py, field).offset
        assert int(val) == want, f"{cname}.{field}: C {val} vs ctypes {want}"


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
