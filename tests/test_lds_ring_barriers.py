"""A property of the COMPILED kernels (not of their source): in the single-barrier LDS rings (conv_gemm_glds_kernel, conv_gemm_rs_kernel and
its flat twin, conv1x1_stream_kernel) the barrier of a step hands the slot the previous step read back to the LDS-DMA, so this wave's reads
of that slot must have COMPLETED -- lgkmcnt(0) -- when it arrives at the barrier.  Source order does not give that: the scheduler sinks a
step's last MFMAs, and the lgkmcnt wait attached to them, below the s_barrier (r06: one wave tile of stale weight rows in ~15 % of the
launches of conv_gemm_rs_kernel<128, 64> with buffer-descriptor pieces on 160-pixel-wide maps, profiles/r06_lds_ring_war_race.txt).
The check reads the disassembly of the built library: walking back from every s_barrier, an s_waitcnt with lgkmcnt(0) must come before
any ds_read.  The ping-pong kernels (conv_gemm_pp_kernel, conv_gemm_pprs_kernel) are exempt by design: their barriers separate a phase's
fragment reads from its MFMAs, and a slot is rewritten no earlier than one barrier after the MFMAs that consumed it (their headers)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import codeobj  # noqa: E402

LIB = os.path.join(ROOT, "efficientteacher_amd", "libet_hip.so")
EXEMPT = ("conv_gemm_pp_kernel", "conv_gemm_pprs_kernel", "conv_gemm_pprs_flat_kernel")
RINGS = ("conv_gemm_glds_kernel", "conv_gemm_rs_kernel", "conv_gemm_rs_flat_kernel", "conv1x1_stream_kernel", "conv1x1_stream_flat_kernel")


def _violations(ins):
    bad = []
    for i, t in enumerate(ins):
        if not t.startswith("s_barrier"):
            continue
        j = i - 1
        while j >= 0:
            u = ins[j]
            if u.startswith("s_waitcnt") and "lgkmcnt(0)" in u:
                break
            if u.startswith("ds_read") or u.startswith("ds_load"):
                bad.append((i, j, u))
                break
            j -= 1
    return bad


@pytest.mark.skipif(not os.path.exists(codeobj.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
@pytest.mark.skipif(not os.path.exists(LIB), reason="libet_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')")
def test_no_lds_read_is_outstanding_at_a_slot_reuse_barrier():
    ks = codeobj.kernels(LIB, match="conv_gemm_rs_kernel")
    seen = {r: 0 for r in RINGS}
    for name, ins in ks.items():
        fam = re.match(r"_Z\d+([A-Za-z0-9_]+?)I", name)
        fam = fam.group(1) if fam else name
        if fam in EXEMPT or not any(t.startswith("s_barrier") for t in ins):
            continue
        bad = _violations(ins)
        assert not bad, (name, [(i, u) for i, _, u in bad][:4])
        if fam in seen:
            seen[fam] += 1
    assert all(v >= 2 for v in seen.values()), seen      # bf16 and fp16 instantiations of every ring kernel were looked at


def test_the_check_itself_flags_a_sunk_wait():
    """the r05/r06 shape of the bug, as text: the last fragment reads of a step, the barrier, and only then their wait"""
    before = ["ds_read_b128 v[84:87], v79", "v_mfma_f32_32x32x16_bf16 v[2:17], v[96:99], v[100:103], v[2:17]", "s_waitcnt vmcnt(0)", "s_barrier",
              "s_waitcnt lgkmcnt(0)", "v_mfma_f32_32x32x16_bf16 v[18:33], v[84:87], v[92:95], v[18:33]"]
    after = [before[0], before[1], "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier", before[5]]
    assert _violations(before) and not _violations(after)
