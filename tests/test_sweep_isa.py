"""CPU: the compiled gfx950 code of the score-matrix sweeps (``gtsfm_amd/csrc/sweep_kernels.hip``) keeps what round 5 found and fixed.

Every row kernel (Sinkhorn, LightGlue's double log-softmax, the match extraction; one wave per row and the 4- / 8-wave tiers, every chunk count,
plain and nontemporal reads: 72 instantiations) prefetches the next row into a second register buffer while it works on the current one. Until
late in round 5 each of those loads sat inside a branch (``if (col < n)``, ``if (i + 1 < rend)``), the compiler's wait-count pass lost count of
the loads in flight and every wait in the row loop was ``s_waitcnt vmcnt(0)`` -- the wave waited for the row it had just prefetched and nothing
overlapped. The kernels now load through a buffer resource with no branch around a load. This test compiles the file to
assembly (hipcc cross-compiles without a GPU, ~20 s) and checks, inside the loops of every row kernel:

* no ``s_waitcnt vmcnt(0)`` (waits are exact counts: the prefetched row stays in flight),
* no scratch traffic (a spilled register is reloaded through the same counter and forces ``vmcnt(0)`` as well),
* the loads are ``buffer_load_dwordx4`` (one 32-bit offset register per chunk), never flat / global loads.
"""

import re
import subprocess

import pytest

from conftest import REPO

CSRC = REPO / "gtsfm_amd" / "csrc"


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    from gtsfm_amd.csrc import build

    out = tmp_path_factory.mktemp("isa") / "sweep_kernels.s"
    flags = [f for f in build.FLAGS if f not in ("-Wall",)]
    cmd = [build.HIPCC, *flags, "-I", str(REPO / "include"), "-S", "--cuda-device-only", str(CSRC / "sweep_kernels.hip"), "-o", str(out)]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-2000:]
    return out.read_text()


def _loop_statistics(assembly: str) -> dict:
    """Per kernel: what stands inside basic blocks the assembler comments mark as part of a loop."""
    stats, name, in_loop = {}, None, False
    for line in assembly.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, in_loop = m.group(1), False
            stats[name] = {"buffer_loads": 0, "other_loads": 0, "vmcnt0": 0, "vmcnt_exact": 0, "scratch": 0, "barriers": 0}
            continue
        if name is None:
            continue
        if ".Lfunc_end" in line:
            name = None
            continue
        if re.match(r"^\.LBB|^; %bb\.", line):
            in_loop = "Loop" in line  # "=>This Inner Loop Header" / "in Loop: Header=..."
            continue
        if not in_loop:
            continue
        text = line.strip()
        if text.startswith("buffer_load_dwordx4"):
            stats[name]["buffer_loads"] += 1
        elif text.startswith(("global_load_dwordx4", "flat_load_dwordx4")):
            stats[name]["other_loads"] += 1
        elif text.startswith("scratch_"):
            stats[name]["scratch"] += 1
        elif text.startswith("s_barrier"):
            stats[name]["barriers"] += 1
        wait = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", text)
        if wait:
            stats[name]["vmcnt0" if wait.group(1) == "0" else "vmcnt_exact"] += 1
    return stats


def test_row_kernels_wait_with_exact_counts(assembly):
    stats = _loop_statistics(assembly)
    rows = {k: v for k, v in stats.items() if re.search(r"(sinkhorn|lg|extract)_rows(_wide)?_kernel", k)}
    # 8 + 12 Sinkhorn (NT x chunks / tiers), 8 + 12 double log-softmax, 8 + 24 extraction (SuperGlue / LightGlue terms)
    assert len(rows) == 72, sorted(rows)
    for name, s in rows.items():
        assert s["buffer_loads"] >= 2, (name, s)  # both register buffers are filled inside the row loop
        assert s["other_loads"] == 0, (name, s)
        assert s["vmcnt0"] == 0, f"{name}: s_waitcnt vmcnt(0) inside the row loop -- the prefetched row is waited for: {s}"
        assert s["vmcnt_exact"] >= 2, (name, s)
        assert s["scratch"] == 0, (name, s)


def test_only_sinkhorn_meets_at_a_barrier_per_row(assembly):
    """The waves of a wide-tier workgroup share a row. Sinkhorn needs the row's log-sum-exp over all slices before its column accumulation: one
    barrier per row (two in the loop body, which handles two rows). The double log-softmax and the extraction do not: their slices' per-row
    results meet once per 32-row block, after the loop (a barrier per row made every wave wait for the slowest wave's loads of every row)."""
    stats = _loop_statistics(assembly)
    for name, s in stats.items():
        if re.search(r"(lg|extract)_rows_wide_kernel", name):
            assert s["barriers"] == 0, (name, s)
        elif re.search(r"sinkhorn_rows_wide_kernel", name):
            assert s["barriers"] == 2, (name, s)
        elif re.search(r"(sinkhorn|lg|extract)_rows_kernel", name):  # one wave per row: nothing to meet for
            assert s["barriers"] == 0, (name, s)


def test_cap_kernels_fit_three_workgroups_per_cu(assembly):
    """The instantiations GTSfM's 5000-keypoint cap runs (four waves x five chunks): at most 168 VGPRs = three workgroups per CU, nothing
    spilled (the extraction needed 234 with its column terms in registers)."""
    for pattern in (r"_Z24extract_rows_wide_kernelILb[01]ELi4ELi5ELb[01]E", r"_Z19lg_rows_wide_kernelILi4ELi5ELb[01]E", r"_Z25sinkhorn_rows_wide_kernelILi4ELi5ELb[01]E"):
        found = 0
        for m in re.finditer(r"\.amdhsa_kernel (" + pattern + r"\w*)(.*?)\.end_amdhsa_kernel", assembly, re.S):
            found += 1
            body = m.group(2)
            vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
            scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
            assert vgprs <= 168 and scratch == 0, (m.group(1), vgprs, scratch)
        assert found >= 2, pattern
