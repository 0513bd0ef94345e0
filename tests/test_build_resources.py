"""Build-time facts the kernels rely on, read from the compiler's own report (dpr_scale_amd/resource_usage.txt, written by the library's
Makefile with -Rpass-analysis=kernel-resource-usage).  The few-rows kernels of csrc/skinny.h count their `s_waitcnt vmcnt(N)` by hand
(LDS-DMAs in flight across barriers): a register spill puts scratch loads / stores -- which ARE vmcnt traffic -- inside those loops and
silently breaks the count (wrong tiles, not a crash).  So: no scratch in any of them, ever."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "dpr_scale_amd", "resource_usage.txt")
HAND_COUNTED = ("sk_sim_kernel", "sk_simp_kernel", "sk_bwd_kernel", "sk_bwdf_kernel", "sk_bwdp_kernel")


def _kernels():
    cur, rows = None, {}
    for ln in open(REPORT, errors="replace"):
        m = re.search(r" Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", ln)
        if m and cur:
            rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    return rows


@pytest.mark.skipif(not os.path.isfile(REPORT), reason="no resource report next to the library (built without the Makefile)")
def test_hand_counted_kernels_never_spill():
    rows = _kernels()
    seen = 0
    for name, r in rows.items():
        if any(f"dprhot{len(k)}{k}" in name or f"{len(k)}{k}" in name for k in HAND_COUNTED):
            seen += 1
            assert r.get("ScratchSize", 0) == 0 and r.get("VGPRs Spill", 0) == 0, f"{name}: {r} -- a spill inside a hand-counted vmcnt loop returns wrong tiles"
    assert seen >= 20, f"only {seen} skinny.h kernels found in the report: has the mangling or the report format changed?"
