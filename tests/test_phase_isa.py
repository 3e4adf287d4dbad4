"""CPU: the phase and ring GEMM kernels' LDS-DMA issue leaves M0 behind (csrc/gemm_impl.h glds16s_lean / glds16s_m0: one scalar write
of M0 + nop + DMA, no save / restore -- round 5, worth ~9 % of the phase kernel's main loop).  That is only legal while NO other
instruction of these kernels reads M0: every gemm_phase_kernel / gemm_ring_kernel in the built library is disassembled and
checked -- M0 appears only as the destination of the scalar instruction in front of a DMA, and every `global_load_lds_dwordx4` is
preceded by such a write and the hazard nop.

Round 6 adds the VALU -> VMEM scalar hazard audit (test_no_asm_vmem_reads_a_freshly_reloaded_sgpr): hipcc reloads spilled SGPRs with
v_readlane_b32 and pads the five wait states a VMEM instruction needs before it may read such a register -- but not for the
instructions INSIDE an asm statement, which it cannot see.  The first deferred-epilogue build read its bias through a stale base
register that way (memory faults in every erf-GELU launch).  tests/probes/asm_hazard_audit.py is the same check on a -save-temps file."""
import os
import re
import subprocess
import tempfile

import pytest

from tests import kernel_resources as K

LIB = os.path.join(K.ROOT, "dreamvla_amd", "libdvla_hip.so")
OBJDUMP = os.path.join(K.LLVM, "llvm-objdump")


def _is_dma(t):
    """an LDS-DMA instruction: global_load_lds_dwordx4, or (round 6, the measurement builds on the buffer-descriptor path) buffer_load_dwordx4 ... lds"""
    return t.startswith("global_load_lds_dwordx4") or (t.startswith("buffer_load_dwordx4") and t.split()[-1] == "lds")


def _functions(*needles):
    """{symbol: [instruction text, ...]} of every kernel of the library's gfx950 code objects whose name contains one of `needles`"""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for k, co in enumerate(K.code_objects(LIB)):
            if not any(n.encode() in co for n in needles):
                continue
            path = os.path.join(d, f"co{k}.elf")
            open(path, "wb").write(co)
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1) if any(n in m.group(1) for n in needles) else None
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
    fns = _functions("gemm_phase_kernel", "gemm_ring_kernel")
    n_phase = sum("gemm_phase_kernel" in n for n in fns)
    n_ring = sum("gemm_ring_kernel" in n for n in fns)
    assert n_phase >= 33 and n_ring >= 96, (n_phase, n_ring)   # phase: 4 layouts x 8 classes + partial-K-tile build (+ measurement
    lean = 0                                                    # variants); ring: 3 configurations x 4 layouts x 8 classes
    for name, ins in fns.items():
        dma = [i for i, t in enumerate(ins) if _is_dma(t)]
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


ATTN_DMA = ("attn_fwd_ring_kernel", "attn_bwd_dq_ring_kernel", "attn_bwd_dkv_ring_kernel", "attn_fwd_short_kernel", "attn_bwd_short_kernel")


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library / llvm-objdump not available")
def test_no_instruction_of_the_attention_dma_kernels_reads_m0():
    """csrc/attention.hip fa_glds16 sets M0 and does not restore it (round 5): legal while nothing in these kernels reads M0"""
    fns = _functions(*ATTN_DMA)
    assert all(any(k in n for n in fns) for k in ATTN_DMA), sorted(fns)
    for name, ins in fns.items():
        dma = [i for i, t in enumerate(ins) if t.startswith("global_load_lds_dwordx4")]
        assert len(dma) >= 2, (name, len(dma))
        for t in ins:
            ops = re.split(r"\s+", t, 1)
            if len(ops) < 2 or "m0" not in ops[1]:
                continue
            assert ops[0] == "s_mov_b32", (name, t)
            dst, srcs = ops[1].split(",", 1)
            assert dst.strip() == "m0" and "m0" not in srcs, (name, t)                 # written, never read
        for i in dma:
            assert ins[i - 1].startswith("s_nop") and ins[i - 2].startswith("s_mov_b32 m0"), (name, ins[i - 3:i + 1])


def _mfma_runs(ins):
    """for every v_mfma in the instruction list: how many `s_waitcnt lgkmcnt` instructions stand between it and the MFMA before it
    (with no branch in between) -- a fragment read waited for in front of EACH multiply shows up as a run of ones"""
    waits, out, seen = 0, [], False
    for t in ins:
        if t.startswith("v_mfma"):
            if seen:
                out.append(waits)
            seen, waits = True, 0
        elif t.startswith("s_waitcnt") and "lgkmcnt" in t:
            waits += 1
        elif t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_barrier"):
            seen, waits = False, 0
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library / llvm-objdump not available")
def test_attention_fragment_reads_are_grouped_in_front_of_the_multiplies():
    """Round 5: hipcc, left alone, recycles ONE register quad for the LDS fragments of a tile -- ds_read, s_waitcnt lgkmcnt(0),
    v_mfma, repeated -- so a wave sat through an LDS round trip per multiply (8 per tile in the forward kernels, 12-16 in the
    backward ones).  attention.hip now requests a tile's fragments as a group (fa_group4) before the first multiply.  The check:
    inside straight-line code, at most a QUARTER of the multiplies of these kernels have an LDS wait directly in front of them
    (before the change: all of them).  attn_bwd_dq_ring_kernel's FIRST block (eight multiplies: S and dP) is exempt: the kernel
    sits at its 128 registers (four waves per SIMD), a group of four fragments there spills 48 registers inside the tile loop and
    even pairs spill 12 (scratch traffic drains the DMA ring) -- only its dQ block is grouped, so seven waits remain there."""
    fns = _functions(*ATTN_DMA)
    for name, ins in fns.items():
        if "attn_fwd_ring_kernel" in name and "ILi0E" not in name:
            continue                               # ablation builds of the forward ring kernel (timing only)
        runs = _mfma_runs(ins)
        assert len(runs) >= 6, (name, runs)
        frac = sum(1 for w in runs if w > 0) / len(runs)
        assert frac <= (0.75 if "dq_ring" in name else 0.3), (name, frac, runs)


def _sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _valu_sgpr_hazards(ins):
    """(index, instruction, states, registers) of every VMEM instruction that reads an SGPR written by v_readlane_b32 /
    v_readfirstlane_b32 fewer than five wait states earlier (straight-line code: a branch empties the window)"""
    bad, recent = [], []
    for i, t in enumerate(ins):
        parts = t.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_endpgm"):
            recent = []
            continue
        if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            used = set()
            for a in args:
                used |= _sregs(a)
            for age, regs in recent:
                if age < 5 and (used & regs):
                    bad.append((i, t, age, sorted(used & regs)))
        n = int(args[0], 0) + 1 if op == "s_nop" else 1
        recent = [(age + n, regs) for age, regs in recent if age + n < 8]
        if op in ("v_readlane_b32", "v_readfirstlane_b32"):
            recent.append((0, _sregs(args[0])))
    return bad


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library / llvm-objdump not available")
def test_no_asm_vmem_reads_a_freshly_reloaded_sgpr():
    """every kernel that issues memory instructions from inline asm (the GEMM DMA kernels, the attention DMA kernels): no VMEM
    instruction reads an SGPR within five wait states of a VALU write of it"""
    fns = _functions("gemm_phase_kernel", "gemm_ring_kernel", *ATTN_DMA)
    assert len(fns) > 130, len(fns)
    found = {name: h for name, ins in fns.items() for h in [_valu_sgpr_hazards(ins)] if h}
    assert not found, {k: v[:2] for k, v in list(found.items())[:4]}


def test_the_hazard_scan_sees_a_planted_hazard():
    ins = ["v_readlane_b32 s26, v212, 37", "v_readlane_b32 s27, v212, 38", "global_load_ushort v1, v1, s[26:27]"]
    assert len(_valu_sgpr_hazards(ins)) == 2
    ins = ["v_readlane_b32 s26, v212, 37", "v_readlane_b32 s27, v212, 38", "s_nop 4", "global_load_ushort v1, v1, s[26:27]"]
    assert not _valu_sgpr_hazards(ins)
    ins = ["v_readfirstlane_b32 s4, v0", "s_add_u32 m0, s17, 0", "s_nop 0", "buffer_load_dwordx4 v199, s[4:7], s46 offen lds"]
    assert len(_valu_sgpr_hazards(ins)) == 1
