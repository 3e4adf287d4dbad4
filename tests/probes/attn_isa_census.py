"""Static census of the attention kernels' tile loops, from the compiler's assembly (no GPU):

    python tests/probes/attn_isa_census.py  [old_attention.hip]  > profiles/r05_attn_trunk_isa.txt

compiles dreamvla_amd/csrc/attention.hip (and, if given, an older copy of it, e.g. `git show <rev>:dreamvla_amd/csrc/attention.hip >
/tmp/old/attention.hip` next to copies of the headers it includes) for gfx950 with the build's flags and prints, per kernel, the
basic blocks of the innermost loop that holds the multiplies: instructions by class, and -- the number round 5 was about -- how
many `s_waitcnt lgkmcnt` stand between consecutive multiplies of a block (an LDS round trip the wave sits through)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNELS = ["attn_fwd_ring_kernelILi0E", "attn_bwd_dq_ring_kernel", "attn_bwd_dkv_ring_kernel", "attn_fwd_short_kernel", "attn_bwd_short_kernel"]


def compile_to_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                    "--cuda-device-only", "-I", os.path.join(ROOT, "dreamvla_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    "-o", out, src], check=True, capture_output=True)
    return open(out).read().split("\n")


def cls(i):
    op = i.split()[0]
    if "mfma" in op: return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load_lds"): return "dma"
    if op.startswith(("global_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_exp"): return "v_exp"
    return "valu"


def blocks_of(lines, kernel):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kernel in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
    blocks, cur = [], {"name": "entry", "loop": None, "depth": 0, "ins": []}
    for l in lines[start + 1:end]:
        s = l.strip()
        m = re.match(r"(\.LBB\S+):\s*(?:;\s*(.*))?", s)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "loop": None, "depth": 0, "ins": []}
            c = m.group(2) or ""
        elif s.startswith("; %bb."):
            blocks.append(cur)                       # an unlabeled fall-through block
            cur = {"name": cur["name"] + "+", "loop": cur["loop"], "depth": cur["depth"], "ins": []}
            c = s
        else:
            c = s[1:] if s.startswith(";") else ""
            if s and not s.startswith((";", ".")):
                cur["ins"].append(s.split(";")[0].strip())
        for mm in re.finditer(r"(?:in Loop: Header=(BB\S+)|This Inner Loop Header:|=>\s*This Inner Loop Header:) ?(?:Depth=(\d+))?", c):
            if mm.group(1):
                cur["loop"], cur["depth"] = mm.group(1), int(mm.group(2) or 0)
            else:
                cur["loop"], cur["depth"] = cur["name"].lstrip(".L"), int(mm.group(2) or 0)
    blocks.append(cur)
    return blocks


def waits_between_mfmas(ins):
    """per multiply of the block: the number of LDS waits since the multiply before it (or the top of the block)"""
    w, out = 0, []
    for t in ins:
        if t.startswith("v_mfma"):
            out.append(w)
            w = 0
        elif t.startswith("s_waitcnt") and "lgkmcnt" in t:
            w += 1
    return out


def census(lines, kernel):
    bl = blocks_of(lines, kernel)
    mf = [b for b in bl if any("mfma" in i for i in b["ins"])]
    depth = max(b["depth"] for b in mf)
    headers = {b["loop"] for b in mf if b["depth"] == depth}
    loop = [b for b in bl if b["loop"] in headers and b["depth"] >= depth]
    rows, tot = [], collections.Counter()
    for b in loop:
        c = collections.Counter(cls(i) for i in b["ins"])
        tot.update(c)
        if len(b["ins"]) >= 8 or c["mfma"]:
            w = waits_between_mfmas(b["ins"])
            rows.append(f"    {b['name']:12s} {len(b['ins']):4d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) +
                        (f"   multiplies that wait for LDS: {sum(1 for x in w if x)}/{len(w)}" if c["mfma"] else ""))
    allw = [x for b in loop for x in waits_between_mfmas(b["ins"])]
    head = (f"  tile loop (static, all paths): {sum(tot.values())} instructions in {len(loop)} blocks: " +
            " ".join(f"{k}={v}" for k, v in sorted(tot.items())) +
            f"\n  multiplies that wait for LDS (a wait between them and the multiply before): {sum(1 for x in allw if x)} of {len(allw)}")
    return head, rows


def main():
    srcs = [("this tree", os.path.join(ROOT, "dreamvla_amd", "csrc", "attention.hip"))]
    if len(sys.argv) > 1:
        srcs.insert(0, ("before (" + sys.argv[1] + ")", sys.argv[1]))
    asm = [(tag, compile_to_asm(p)) for tag, p in srcs]
    for k in KERNELS:
        print("=" * 110)
        print(k)
        for tag, lines in asm:
            head, rows = census(lines, k)
            print(f" {tag}")
            print(head)
            print("\n".join(rows))


if __name__ == "__main__":
    main()
