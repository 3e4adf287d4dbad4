"""Inline-asm hazard audit of a hipcc -save-temps .s file (gfx950): an SGPR written by a VALU instruction (v_readlane_b32 /
v_readfirstlane_b32 -- the compiler's SGPR-spill reloads) needs FIVE wait states before a VMEM instruction reads it as base,
descriptor or scalar offset.  hipcc pads that hazard for its own instructions but not for the instructions inside an asm statement
(cdna_hip_programming.md section 5.7): a reload right in front of `;;#ASMSTART` feeds the DMA / load a stale register.
    python tests/probes/asm_hazard_audit.py file.s   -> prints every asm VMEM instruction with fewer than 5 states behind such a write"""
import re
import sys


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def states(op, args):
    if op == "s_nop":
        return int(args[0], 0) + 1
    return 1


def audit(path):
    lines = [l.split(";")[0].strip() if not l.strip().startswith(";;#") else l.strip() for l in open(path)]
    bad = []
    recent = []          # (states since, sgprs written by VALU)
    in_asm = False
    for ln, l in enumerate(lines, 1):
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith(".") or l.endswith(":"):
            if l.endswith(":"):
                recent = []          # (a label: unknown predecessors -- the compiler's own padding covers fall-through paths only partly; keep it simple)
            continue
        parts = l.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if in_asm and (op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_")):
            used = set()
            for a in args:
                used |= sregs(a)
            for age, regs in recent:
                if age < 5 and (used & regs):
                    bad.append((ln, l, age, sorted(used & regs)))
        n = states(op, args)
        recent = [(age + n, regs) for age, regs in recent if age + n < 8]
        if op in ("v_readlane_b32", "v_readfirstlane_b32"):
            recent.append((0, sregs(args[0])))
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for ln, l, age, regs in audit(p):
            print(f"{p}:{ln}: {l}   <- s{regs} written by a VALU instruction {age} state(s) earlier")
            rc = 1
    sys.exit(rc)
