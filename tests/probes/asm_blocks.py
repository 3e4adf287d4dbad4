"""Per-basic-block census of a hipcc -save-temps .s file: MFMAs, scratch traffic, barriers, stores, DMA pieces, VALU.
    python tests/probes/asm_blocks.py file.s [kernel-substring]
Used to see WHERE the register allocator spills (inside the K loop or on a boundary path) when the phase kernel changes."""
import re
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else None
    kern = None
    blocks = []
    cur = None
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
            cur = None
            continue
        if want and (kern is None or want not in kern):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m or cur is None:
            cur = {"label": m.group(1) if m else "entry", "line": ln, "n": 0, "mfma": 0, "scr_ld": 0, "scr_st": 0, "bar": 0, "st": 0,
                   "dma": 0, "valu": 0, "ds": 0, "wait": 0, "kern": kern}
            blocks.append(cur)
            if m:
                continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        cur["n"] += 1
        if op.startswith("v_mfma"):
            cur["mfma"] += 1
        elif op.startswith("scratch_load"):
            cur["scr_ld"] += 1
        elif op.startswith("scratch_store"):
            cur["scr_st"] += 1
        elif op == "s_barrier":
            cur["bar"] += 1
        elif op.startswith("buffer_store") or op.startswith("global_store"):
            cur["st"] += 1
        elif op.startswith("global_load_lds"):
            cur["dma"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op == "s_waitcnt":
            cur["wait"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
    for b in blocks:
        if b["mfma"] or b["scr_ld"] or b["scr_st"] or b["st"] or b["bar"]:
            print(f'{b["label"]:>12} @{b["line"]:<6} n={b["n"]:<5} mfma={b["mfma"]:<3} bar={b["bar"]} ds={b["ds"]:<3} dma={b["dma"]} valu={b["valu"]:<4} '
                  f'st={b["st"]:<3} scratch ld/st={b["scr_ld"]}/{b["scr_st"]} wait={b["wait"]}')


if __name__ == "__main__":
    main()
