"""Walk the control-flow graph of one function's ISA with the set of vector-memory operations in flight (s_waitcnt vmcnt(n)
retires all but the n youngest) and flag every instruction that reads or writes a VGPR whose load has not been waited for."""
import re, sys
fn = sys.argv[1]
lines = open(fn).read().split('\n')
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB[0-9_]+):', l)
    if m: labels[m.group(1)] = i
def regs_of(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out
flags = set()
seen = set()
work = [(0, ())]
steps = 0
while work:
    i, inflight = work.pop()
    inflight = list(inflight)
    while i < len(lines):
        inflight = inflight[-63:]
        key = (i, frozenset().union(*inflight) if inflight else frozenset(), len(inflight))
        t = lines[i].split(';')[0].strip()
        if not t or t.endswith(':') or t.startswith('.'):
            if t.endswith(':'):
                if key in seen: break
                seen.add(key)
            i += 1; continue
        steps += 1
        op = t.split()[0]
        if op.startswith('s_waitcnt'):
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                inflight = inflight[len(inflight) - n:] if n else []
            i += 1; continue
        if op.startswith('s_cbranch'):
            tgt = t.split()[-1]
            if tgt in labels: work.append((labels[tgt], tuple(frozenset(r) for r in inflight)))
            i += 1; continue
        if op == 's_branch':
            tgt = t.split()[-1]
            i = labels[tgt]; continue
        if op in ('s_endpgm', 's_setpc_b64'): break
        is_vmem = op.startswith(('global_', 'flat_', 'scratch_', 'buffer_'))
        used = regs_of(t)
        cur = set().union(*inflight) if inflight else set()
        if is_vmem:
            if 'load' in op and 'atomic' not in op:
                dst = regs_of(t.split(',')[0])
                bad = used & cur
                if bad: flags.add(f"{i+1}: vmem touches in-flight {sorted(bad)[:6]} :: {t[:80]}")
                inflight.append(frozenset(dst))
            else:
                bad = used & cur
                if bad: flags.add(f"{i+1}: store/atomic reads in-flight {sorted(bad)[:6]} :: {t[:80]}")
                inflight.append(frozenset())
        else:
            bad = used & cur
            if bad: flags.add(f"{i+1}: in-flight register touched {sorted(bad)[:6]} :: {t[:80]}")
        i += 1
for f in sorted(flags, key=lambda x: int(x.split(':')[0]))[:40]: print(f)
print("flags:", len(flags), "blocks visited:", len(seen), "instructions walked:", steps)
