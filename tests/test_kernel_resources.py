"""CPU: register / scratch budget of every kernel in the built library (tests/kernel_resources.py reads the code objects'
metadata).  A kernel that starts spilling or touching scratch shows up here at build time, not as a slower step on the GPU:
everything is spill-free except a short, named list."""
import os
import re

import pytest

from tests import kernel_resources as K

LIB = os.path.join(K.ROOT, "dreamvla_amd", "libdvla_hip.so")

# (pattern of the mangled name, scratch bytes allowed, spilled VGPRs allowed)
ALLOWED = [
    # the GENERIC epilogue class (EPI_GEN = 7: ReLU / SiLU heads, activation + residual) of the two 256 x 256 kernels keeps its
    # runtime-selected activation / act' code next to the full accumulator set: a few dozen spills outside the K-tile body (round 5,
    # two copies of the K loop per tile: up to 57, four reloads of them in the once-per-tile cursor re-open block).  The step's
    # time is in classes 0-6 (profiles/r04_gemm_breakdown.json), which must be clean.
    (r"gemm_phase_kernelILb[01]ELb[01]ELi7ELi0E", 160, 60),
    (r"gemm_ring_kernelINS_4RCfgILi2ELi4ELi4ELi2ELi4ELi2ELi4ELi32EEELb[01]ELb[01]ELi7E", 140, 36),
    # round 5 (scalar K bases of the lean DMA issue): the 256 x 256 ring kernel's fp32 class in the TN layout -- no problem key of the
    # step uses it (every weight gradient is TT, which is clean) -- spills 4 registers outside its stage loop
    (r"gemm_ring_kernelINS_4RCfgILi2ELi4ELi4ELi2ELi4ELi2ELi4ELi32EEELb1ELb0ELi6E", 24, 4),
    # round 5: the two k-sum builds of the phase kernel's fp32 class (TT layout: DBG 8192, and 8320 with a partial last K-tile) save
    # ONE 8-byte value in front of the K loops and reload it once behind them (for the partial-row store); nothing inside a loop
    (r"gemm_phase_kernelILb1ELb1ELi6ELi(8192|8320)E", 16, 2),
    # dQ ring kernel with the dropout generator: one spilled pair outside the loop
    (r"attn_bwd_dq_ring_kernel", 16, 2),
    # the one-XCD sampler kernels: the frame of their one real call (step_boundary) and of the per-phase calls; no VGPR spills --
    # a spilled register with a hand-issued load in flight would be garbage (tests/test_dit_team_isa.py audits the ISA as well)
    (r"dit_team_kernel", 512, 0),
]


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_every_kernel_within_its_budget():
    rows = K.kernels(LIB)
    assert len(rows) >= 150, len(rows)
    over = []
    for r in rows:
        assert r["wavefront"] == 64 and not r["dynamic_stack"], r["name"]
        assert r["vgpr"] + r["agpr"] <= 512 and r["static_lds_bytes"] <= 160 * 1024, r["name"]
        scratch_ok, spill_ok = 0, 0
        for pat, s, v in ALLOWED:
            if re.search(pat, r["name"]):
                scratch_ok, spill_ok = s, v
        if r["scratch_bytes"] > scratch_ok or r["vgpr_spill"] > spill_ok:
            over.append((r["name"], r["scratch_bytes"], r["vgpr_spill"]))
    assert not over, over
    # the kernels the step's time is in are there and clean
    hot = [r for r in rows if re.search(r"gemm_phase_kernelILb[01]ELb[01]ELi[0-5]ELi0E|gemm_phase_kernelILb[01]ELb[01]ELi6ELi(0|128)E|gemm_skinny_kernel|attn_(fwd|bwd)_short_kernel|"
                                        r"attn_fwd_ring_kernel|attn_bwd_dkv_ring_kernel|ln_(fwd|bwd)_kernel", r["name"])]
    assert len(hot) >= 30 and all(r["scratch_bytes"] == 0 and r["vgpr_spill"] == 0 for r in hot)
