"""CPU: the look-ahead sampler kernel (csrc/dit_team.hip, dit_team_kernel_ahead) is cross-compiled and its ISA is audited: no
register spills (a spilled register with a hand-issued load in flight would be saved as garbage) and no instruction that touches
a register whose load has not been waited for (tests/probes/isa_inflight_audit.py walks the control-flow graph)."""
import os
import re
import shutil
import subprocess

import pytest

from tests.probes import isa_inflight_audit as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_audit_flags_a_read_before_the_wait():
    bad = """
_Z1fv:
	global_load_dwordx4 v[4:7], v[0:1], off
	v_add_f32_e32 v8, v4, v9
	s_waitcnt vmcnt(0)
	s_endpgm
.Lfunc_end0:
""".split("\n")
    flags, _, _ = A.audit(bad)
    assert len(flags) == 1 and "v_add_f32" in flags[0]
    ok = """
_Z1fv:
	global_load_dwordx4 v[4:7], v[0:1], off
	global_load_dwordx4 v[10:13], v[0:1], off offset:64
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v8, v4, v9
	s_cbranch_scc1 .LBB0_2
	v_mov_b32_e32 v20, v21
.LBB0_2:
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v8, v10, v9
	s_endpgm
.Lfunc_end0:
""".split("\n")
    assert A.audit(ok)[0] == []
    # the same kind of use on a path that skips the wait
    # (.LBB0_3 sits after the wait in this listing: reached by the branch with one load still in flight)
    lines = ["_Z1fv:", "\tglobal_load_dwordx4 v[10:13], v[0:1], off", "\ts_cbranch_scc1 .LBB0_3", "\ts_waitcnt vmcnt(0)", ".LBB0_3:",
             "\tv_add_f32_e32 v8, v11, v9", "\ts_endpgm", ".Lfunc_end0:"]
    assert len(A.audit(lines)[0]) == 1


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_lookahead_kernel_has_no_spills_and_touches_no_register_in_flight(tmp_path):
    out = tmp_path / "dit_team.o"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                        os.path.join(ROOT, "dreamvla_amd", "csrc", "dit_team.hip"), "-o", str(out), "-save-temps=obj"],
                       capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    listing = open(tmp_path / "dit_team-hip-amdgcn-amd-amdhsa-gfx950.s").read()
    # kernel metadata: the look-ahead kernel must not spill
    meta = listing[listing.index("amdhsa.kernels"):]
    blocks = re.split(r"\n  - ", meta)
    mine = [b for b in blocks if "dit_team_kernel_ahead" in b and ".vgpr_spill_count" in b]
    assert mine, "kernel metadata not found"
    assert re.search(r"\.vgpr_spill_count:\s+0\b", mine[0]), mine[0][-600:]
    lines = A.function_lines(listing, "dit_team_kernel_ahead")
    flags, nblocks, steps = A.audit(lines)
    assert steps > 5000 and nblocks > 100, (nblocks, steps)          # the walk covered the function
    assert flags == [], flags[:5]
    # the loads the audit is about are really there: hand-issued 16-byte loads, counted waits that leave requests in flight
    body = "\n".join(lines)
    assert body.count("global_load_dwordx4") > 50 and re.search(r"s_waitcnt vmcnt\((?!0\))\d+\)", body)
