"""CPU: the compiled gfx950 code of the attention kernels (``gtsfm_amd/csrc/attention_kernels.hip``; hipcc cross-compiles without a GPU, ~6 s).

* no instantiation uses scratch memory;
* every arithmetic multiplies on its own matrix instruction only: exact fp32 on ``v_mfma_f32_32x32x2_f32``, bf16x3 on ``v_mfma_f32_32x32x16_bf16``,
  f16x2 on ``v_mfma_f32_32x32x16_f16`` (and f16x2 converts with the packed round-to-nearest instruction and takes its residuals with ``v_fma_mix_f32``);
* in the split-arithmetic kernels the DMA issue of the next K / V tile is a block of its own behind its barrier -- no matrix or transcendental
  instruction between the barrier and the last ``global_load_lds`` of the block. A build in which the scheduler had interleaved them returned scores that
  differed from run to run beside a second stream (profiles/r06_x3_two_stream_bisect.txt); the ``sched_barrier`` fences in ``attention_x3_kernel`` keep
  the block intact and this test keeps the fences honest. (The GPU-side guard is tests/test_attention_bf16x3_gpu.py::test_bf16x3_two_stream_pipeline_is_deterministic.)"""

import re
import subprocess

import pytest

from conftest import REPO

CSRC = REPO / "gtsfm_amd" / "csrc"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    from gtsfm_amd.csrc import build

    out = tmp_path_factory.mktemp("isa") / "attention_kernels.s"
    flags = [f for f in build.FLAGS if f not in ("-Wall",)] + build.PER_FILE_FLAGS.get("attention_kernels.hip", [])
    cmd = [build.HIPCC, *flags, "-I", str(REPO / "include"), "-S", "--cuda-device-only", str(CSRC / "attention_kernels.hip"), "-o", str(out)]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-2000:]
    text = out.read_text()
    bodies = {}
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm", text, re.S):
        bodies[m.group(1)] = [ln.strip() for ln in m.group(2).splitlines() if ln.strip() and not ln.strip().startswith((";", "."))]
    scratch = {m.group(1): int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", m.group(2)).group(1))
               for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)}
    return bodies, scratch


def test_no_attention_kernel_uses_scratch(kernels):
    _, scratch = kernels
    mine = {k: v for k, v in scratch.items() if "attention" in k}
    assert len(mine) == 10, sorted(mine)  # combine, split<2|3>, x3<fused|split, 2|3>, dma<fused 4|8, split 4>
    assert all(v == 0 for v in mine.values()), mine


def test_each_arithmetic_uses_its_own_matrix_instruction(kernels):
    bodies, _ = kernels
    seen = set()
    for name, body in bodies.items():
        text = "\n".join(body)
        counts = {"f32": text.count("v_mfma_f32_32x32x2_f32"), "bf16": text.count("v_mfma_f32_32x32x16_bf16"), "f16": text.count("v_mfma_f32_32x32x16_f16")}
        if "attention_dma_kernel" in name:
            want = "f32"
        elif "attention_x3_kernel" in name:
            want = "f16" if name.endswith("ELi2EEv10AttnParams") else "bf16"
            if want == "f16":
                assert "v_cvt_pk_f16_f32" in text and "v_fma_mix_f32" in text and "v_cvt_pkrtz" not in text, name
        else:
            assert sum(counts.values()) == 0, (name, counts)
            continue
        seen.add(want)
        assert counts[want] > 0 and sum(counts.values()) == counts[want], (name, counts)
    assert seen == {"f32", "bf16", "f16"}


def test_the_dma_issue_of_the_split_arithmetic_kernels_is_a_block_of_its_own(kernels):
    bodies, _ = kernels
    checked = 0
    for name, body in bodies.items():
        if "attention_x3_kernel" not in name:
            continue
        pieces = 4 if name.endswith("ELi2EEv10AttnParams") else 6  # LDS-DMA instructions per wave and tile: 2 x the number of 16-bit pieces
        i = 0
        while i < len(body):
            if not body[i].startswith("s_barrier"):
                i += 1
                continue
            # the block behind this barrier: up to the `pieces`-th global_load_lds, if the first one comes before any LDS read / branch / barrier
            j, found, block = i + 1, 0, []
            while j < len(body) and found < pieces:
                op = body[j].split()[0]
                if op.startswith("global_load_lds"):
                    found += 1
                elif op in ("s_barrier", "s_endpgm") or op.startswith(("s_cbranch", "ds_read")):
                    break
                block.append(op)
                j += 1
            if found == pieces:
                bad = [op for op in block if op.startswith(("v_mfma", "v_exp", "v_permlane"))]
                assert not bad, (name, i, bad)
                checked += 1
            i = j
    assert checked >= 8, checked  # K and V blocks of the tile loop in four instantiations (the prologue's loads sit in front of the first barrier)
