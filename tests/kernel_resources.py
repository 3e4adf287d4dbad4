"""Register / scratch / LDS budget of every kernel in the built library, read from the code objects' metadata (no GPU).

    python tests/kernel_resources.py [dreamvla_amd/libdvla_hip.so] [out.json]

The HIP fat binary of the shared library (section .hip_fatbin) is a sequence of clang offload bundles; each gfx950 entry is an
ELF code object whose NT_AMDGPU_METADATA note lists, per kernel, the allocated VGPRs / AGPRs / SGPRs, spilled registers, scratch
(`.private_segment_fixed_size`) and static LDS.  tests/test_kernel_resources.py holds the hot kernels to their budgets."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        b = open(fat, "rb").read()
    pos = 0
    while True:
        i = b.find(MAGIC, pos)
        if i < 0:
            return
        p = i + len(MAGIC)
        cnt, = struct.unpack_from("<Q", b, p)
        p += 8
        for _ in range(cnt):
            off, size, tl = struct.unpack_from("<QQQ", b, p)
            p += 24
            triple = b[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                yield b[i + off:i + off + size]
        pos = i + len(MAGIC)


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.split("\n")[:len(names)]
    except Exception:  # noqa: BLE001
        return names


def kernels(lib):
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for k, co in enumerate(code_objects(lib)):
            path = os.path.join(d, f"co{k}.elf")
            open(path, "wb").write(co)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n  - (?=\.agpr_count|\.args)", notes)[1:]:
                def field(name, default=0):
                    m = re.search(r"\." + name + r":\s+(\S+)", blk)
                    return m.group(1) if m else default
                if not re.search(r"\.name:", blk):
                    continue
                rows.append({"name": field("name"), "vgpr": int(field("vgpr_count")), "agpr": int(field("agpr_count")),
                             "sgpr": int(field("sgpr_count")), "vgpr_spill": int(field("vgpr_spill_count")),
                             "sgpr_spill": int(field("sgpr_spill_count")), "scratch_bytes": int(field("private_segment_fixed_size")),
                             "static_lds_bytes": int(field("group_segment_fixed_size")), "max_threads": int(field("max_flat_workgroup_size")),
                             "wavefront": int(field("wavefront_size")), "dynamic_stack": field("uses_dynamic_stack", "false") == "true"})
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        r["demangled"] = n
    return rows


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "dreamvla_amd", "libdvla_hip.so")
    rows = kernels(lib)
    rows.sort(key=lambda r: (-r["scratch_bytes"], -r["vgpr"], r["name"]))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(rows, f, indent=0)
    print(f"{len(rows)} kernels; with scratch: {sum(r['scratch_bytes'] > 0 for r in rows)}; with spilled VGPRs: {sum(r['vgpr_spill'] > 0 for r in rows)}")
    for r in rows[:40]:
        print(f"{r['scratch_bytes']:6d} B scratch  {r['vgpr']:3d}+{r['agpr']:3d} regs  spills v{r['vgpr_spill']} s{r['sgpr_spill']}  lds {r['static_lds_bytes']:6d}  {r['demangled'][:110]}")


if __name__ == "__main__":
    main()
