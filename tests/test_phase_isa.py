"""CPU: the phase and ring GEMM kernels' LDS-DMA issue leaves M0 behind (csrc/gemm_impl.h glds16s_lean / glds16s_m0: one scalar write
of M0 + nop + DMA, no save / restore -- round 5, worth ~9 % of the phase kernel's main loop).  That is only legal while NO other
instruction of these kernels reads M0: every gemm_phase_kernel / gemm_ring_kernel in the built library is disassembled and
checked -- M0 appears only as the destination of the scalar instruction in front of a DMA, and every `global_load_lds_dwordx4` is
preceded by such a write and the hazard nop."""
import os
import re
import subprocess
import tempfile

import pytest

from tests import kernel_resources as K

LIB = os.path.join(K.ROOT, "dreamvla_amd", "libdvla_hip.so")
OBJDUMP = os.path.join(K.LLVM, "llvm-objdump")


def _phase_functions():
    """{symbol: [instruction text, ...]} of every gemm_phase_kernel in the library's gfx950 code objects"""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for k, co in enumerate(K.code_objects(LIB)):
            if b"gemm_phase_kernel" not in co and b"gemm_ring_kernel" not in co:
                continue
            path = os.path.join(d, f"co{k}.elf")
            open(path, "wb").write(co)
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1) if ("gemm_phase_kernel" in m.group(1) or "gemm_ring_kernel" in m.group(1)) else None
                    if cur:
                        out[cur] = []
                    continue
                if cur and line.strip():
                    ins = line.split("//")[0].strip()
                    if ins:
                        out[cur].append(ins)
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library / llvm-objdump not available")
def test_no_instruction_of_the_phase_kernels_reads_m0():
    fns = _phase_functions()
    n_phase = sum("gemm_phase_kernel" in n for n in fns)
    n_ring = sum("gemm_ring_kernel" in n for n in fns)
    assert n_phase >= 33 and n_ring >= 96, (n_phase, n_ring)   # phase: 4 layouts x 8 classes + partial-K-tile build (+ measurement
    lean = 0                                                    # variants); ring: 3 configurations x 4 layouts x 8 classes
    for name, ins in fns.items():
        dma = [i for i, t in enumerate(ins) if t.startswith("global_load_lds_dwordx4")]
        assert len(dma) >= (16 if "phase" in name else 4), (name, len(dma))  # phase: prologue 16 + 8 per copy of the K loop
        legacy = any(re.match(r"s_mov_b32 s\d+, m0", t) for t in ins)      # the round-4 issue code saves and restores M0 (variants 40 / 41)
        if legacy:
            continue
        lean += 1
        for i, t in enumerate(ins):
            if "m0" not in re.split(r"\s+", t, 1)[-1] and not t.startswith("s_add_u32 m0") and not t.startswith("s_mov_b32 m0"):
                continue
            ops = re.split(r"\s+", t, 1)
            assert ops[0] in ("s_add_u32", "s_mov_b32"), (name, t)
            dst, srcs = ops[1].split(",", 1)
            assert dst.strip() == "m0" and "m0" not in srcs, (name, t)                 # written, never read
        for i in dma:                                                                   # every DMA: M0 write, hazard nop, DMA
            assert ins[i - 1].startswith("s_nop") and (ins[i - 2].startswith("s_add_u32 m0") or ins[i - 2].startswith("s_mov_b32 m0")), \
                (name, ins[i - 3:i + 1])
    assert lean >= 33 + 96, lean
