"""Does any instruction of a gfx950 function touch a VGPR whose load has not been waited for?

csrc/dit_team.hip issues its weight / activation loads by hand (inline asm): the compiler does not know that they are in flight,
so nothing it emits waits for them -- which is the point (a phase leaves the next phase's weight requests outstanding across its
barrier), and the danger: a register copy, a spill or an address computation that reads such a register before the load lands
reads garbage, silently.  This tool walks the control-flow graph of one function in the compiler's ISA listing (hipcc -save-temps)
with the list of vector-memory operations in flight -- `s_waitcnt vmcnt(n)` retires all but the n youngest, branches fork the
walk -- and reports every instruction that reads or writes a register of a load still in flight.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c dreamvla_amd/csrc/dit_team.hip -o /tmp/x.o -save-temps=obj
    python tests/probes/isa_inflight_audit.py /tmp/x-hip-amdgcn-amd-amdhsa-gfx950.s dit_team_kernel_ahead

tests/test_dit_team_isa.py runs it on every build of the look-ahead kernel (CPU: hipcc cross-compiles)."""
import re
import sys


def regs_of(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def function_lines(listing, name_part):
    """the lines of the first function whose mangled name contains `name_part`"""
    lines = listing.split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(name_part) + r'\w*:', l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
    return lines[start:end]


def audit(lines, max_in_flight=63):
    """-> (flags, blocks visited, instructions walked).  vmcnt is a 6-bit counter: at most 63 operations are tracked."""
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB[0-9_]+):', l)
        if m:
            labels[m.group(1)] = i
    flags, seen, steps = set(), set(), 0
    work = [(0, ())]
    while work:
        i, inflight = work.pop()
        inflight = list(inflight)
        while i < len(lines):
            inflight = inflight[-max_in_flight:]
            t = lines[i].split(';')[0].strip()
            if not t or t.endswith(':') or t.startswith('.'):
                if t.endswith(':'):
                    key = (i, frozenset().union(*inflight) if inflight else frozenset(), len(inflight))
                    if key in seen:
                        break
                    seen.add(key)
                i += 1
                continue
            steps += 1
            op = t.split()[0]
            if op.startswith('s_waitcnt'):
                m = re.search(r'vmcnt\((\d+)\)', t)
                if m:
                    n = int(m.group(1))
                    inflight = inflight[len(inflight) - n:] if n else []
                i += 1
                continue
            if op.startswith('s_cbranch'):
                tgt = t.split()[-1]
                if tgt in labels:
                    work.append((labels[tgt], tuple(inflight)))
                i += 1
                continue
            if op == 's_branch':
                i = labels[t.split()[-1]]
                continue
            if op in ('s_endpgm', 's_setpc_b64'):
                break
            used = regs_of(t)
            cur = set().union(*inflight) if inflight else set()
            if op.startswith(('global_', 'flat_', 'scratch_', 'buffer_')):
                bad = used & cur
                if bad:
                    flags.add(f"{i + 1}: memory operation touches a register in flight {sorted(bad)[:6]} :: {t[:90]}")
                is_load = 'load' in op and 'atomic' not in op
                inflight.append(frozenset(regs_of(t.split(',')[0])) if is_load else frozenset())
            else:
                bad = used & cur
                if bad:
                    flags.add(f"{i + 1}: register in flight touched {sorted(bad)[:6]} :: {t[:90]}")
            i += 1
    return sorted(flags, key=lambda x: int(x.split(':')[0])), len(seen), steps


def main():
    listing = open(sys.argv[1]).read()
    lines = function_lines(listing, sys.argv[2]) if len(sys.argv) > 2 else listing.split('\n')
    flags, blocks, steps = audit(lines)
    for f in flags[:40]:
        print(f)
    print("flags:", len(flags), "blocks visited:", blocks, "instructions walked:", steps)
    return 1 if flags else 0


if __name__ == "__main__":
    sys.exit(main())
