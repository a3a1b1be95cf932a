"""CPU: the compiled gfx950 code of the matchers' LDS-DMA GEMMs (``gtsfm_amd/csrc/gemm_dma_kernels.hip``) uses no scratch memory.

Until round 6 the residual and the rotary epilogue variants of ``gemm_dma_walk_kernel`` sat at the 256-VGPR ceiling and spilled (20 / 28 bytes of
scratch; 7.3 % of the headline step ran in them): the bias of the lane's 64 columns was held in 32 registers through the whole column block. It now
joins in the transposed epilogue (8 registers, same operation order per element: same bits), and the epilogue reads its lane coordinates where it
uses them. This test compiles the file to assembly (hipcc cross-compiles without a GPU, ~5 s) and checks every instantiation's kernel descriptor."""

import re
import subprocess

import pytest

from conftest import REPO

CSRC = REPO / "gtsfm_amd" / "csrc"


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    from gtsfm_amd.csrc import build

    out = tmp_path_factory.mktemp("isa") / "gemm_dma_kernels.s"
    flags = [f for f in build.FLAGS if f not in ("-Wall",)]
    cmd = [build.HIPCC, *flags, "-I", str(REPO / "include"), "-S", "--cuda-device-only", str(CSRC / "gemm_dma_kernels.hip"), "-o", str(out)]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-2000:]
    return out.read_text()


def _descriptors(assembly: str) -> dict:
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", assembly, re.S):
        field = lambda key: int(re.search(key + r"\s+(\d+)", m.group(2)).group(1))  # noqa: E731
        out[m.group(1)] = {"vgprs": field(r"\.amdhsa_next_free_vgpr"), "scratch": field(r"\.amdhsa_private_segment_fixed_size")}
    return out


def test_no_gemm_instantiation_uses_scratch(assembly):
    kernels = {k: v for k, v in _descriptors(assembly).items() if "gemm_dma_" in k}
    # {plain, residual, rotary} x {fp32, bf16x3, f16x2} x {128 x 128 walk tiles, 64 x 64 small tiles}
    assert len(kernels) == 18, sorted(kernels)
    for name, d in kernels.items():
        assert d["scratch"] == 0, f"{name}: private_segment_fixed_size = {d['scratch']} (a spill inside an MFMA kernel)"
        assert d["vgprs"] <= 256, (name, d)
    assert "scratch_" not in "".join(line for line in assembly.splitlines() if line.strip().startswith("scratch_"))


def test_matrix_instructions_are_where_they_should_be(assembly):
    """fp32 variants multiply on v_mfma_f32_32x32x2_f32 only, bf16x3 variants on v_mfma_f32_32x32x16_bf16 only, f16x2 variants on
    v_mfma_f32_32x32x16_f16 only -- and the f16x2 split converts with the packed round-to-nearest instruction."""
    seen = set()
    for block in re.split(r"\n(?=_Z\w+:)", assembly):
        name = block.split(":")[0]
        if "gemm_dma_" not in name or not name.startswith("_Z"):
            continue
        math = int(re.search(r"ELi(\d)EEv", name).group(1))  # the third template argument: 0 fp32, 1 bf16x3, 2 f16x2
        seen.add(math)
        counts = [block.count("v_mfma_f32_32x32x2_f32"), block.count("v_mfma_f32_32x32x16_bf16"), block.count("v_mfma_f32_32x32x16_f16")]
        assert counts[math] > 0 and sum(counts) == counts[math], (name, counts)
        if math == 2:
            assert "v_cvt_pk_f16_f32" in block and "v_cvt_pkrtz" not in block, name
    assert seen == {0, 1, 2}
