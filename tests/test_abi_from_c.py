"""The drop-in boundary is a C ABI: ``include/gtsfm_amd.h`` has to be valid plain C and ``libgtsfm_amd.so`` has to link and
run from a C program. gcc compiles ``tests/abi/abi_from_c.c`` with ``-std=c99 -pedantic -Werror`` against the header and the
in-tree library and runs it (host-only entry points and argument checks: no GPU needed); on the GPU box a second C program
(``tests/abi/abi_device_from_c.c``) allocates device memory with the HIP runtime's C API and drives a device entry point."""

import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_is_plain_c_and_library_links_from_c(built_library, tmp_path):
    exe = tmp_path / "abi_from_c"
    lib_dir = Path(built_library).parent
    subprocess.run(
        ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{REPO / 'include'}", str(REPO / "tests" / "abi" / "abi_from_c.c"), f"-L{lib_dir}",
         "-lgtsfm_amd", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
        check=True, capture_output=True, text=True,
    )
    env = dict(os.environ, LD_LIBRARY_PATH=f"{lib_dir}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "abi_from_c OK" in run.stdout


def _build_c(tmp_path, built_library, source: str, extra=()):
    exe = tmp_path / Path(source).stem
    lib_dir = Path(built_library).parent
    subprocess.run(
        ["gcc", "-std=c99", "-ffp-contract=off", "-Wall", "-Werror", f"-I{REPO / 'include'}", *extra, str(REPO / "tests" / "abi" / source), f"-L{lib_dir}", "-lgtsfm_amd",
         f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-o", str(exe)],
        check=True, capture_output=True, text=True,
    )
    return exe, dict(os.environ, LD_LIBRARY_PATH=f"{lib_dir}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_device_program_compiles_as_c(built_library, tmp_path):
    """CPU half of the device test: the C program that drives a device entry point compiles and links here (it runs under -m gpu)."""
    exe, _ = _build_c(tmp_path, built_library, "abi_device_from_c.c", ["-isystem", "/opt/rocm/include"])
    assert exe.exists()


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_model_program_compiles_as_c_and_its_header_is_current(built_library, tmp_path):
    """CPU half of the model-level test: the program compiles and links here, and the expectation header it includes is what
    oracle/make_abi_model_expectation.py writes today (same hash, same oracle) -- checked on its cheap fields."""
    exe, _ = _build_c(tmp_path, built_library, "abi_model_from_c.c", ["-isystem", "/opt/rocm/include", f"-I{REPO / 'tests' / 'abi'}"])
    assert exe.exists()
    import re
    import sys

    sys.path.insert(0, str(REPO))
    import numpy as np
    import torch

    from oracle import make_abi_model_expectation as gen
    from oracle import superpoint_oracle as spo

    header = (REPO / "tests" / "abi" / "abi_model_expected.h").read_text()
    seed = int(re.search(r"#define ABI_MODEL_SEED (\d+)u", header).group(1))
    k = int(re.search(r"#define ABI_MODEL_K (\d+)", header).group(1))
    xy = np.array(re.findall(r"\{(\d+), (\d+)\}", header.split("abi_model_xy")[1].split(";")[0]), dtype=np.int64)
    with torch.no_grad():
        out = spo.superpoint_forward(gen.tensors(seed), spo.gray_u8_to_tensor(gen.image(seed)))
    assert out["keypoints"].shape[0] == k == len(xy)
    np.testing.assert_array_equal(out["keypoints"].numpy().astype(np.int64), xy)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_model_level_entry_points_driven_from_c(built_library, tmp_path):
    """gtsfm_sp_pack_weights -> gtsfm_sp_workspace_bytes -> gtsfm_sp_forward from a C program that owns its device memory and stream (no Python
    in the process): keypoint count and every coordinate identical to the oracle's numbers in abi_model_expected.h, scores / descriptors 1e-4."""
    exe, env = _build_c(tmp_path, built_library, "abi_model_from_c.c", ["-isystem", "/opt/rocm/include", f"-I{REPO / 'tests' / 'abi'}"])
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "abi_model_from_c OK" in run.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_lightglue_program_compiles_as_c_and_its_header_is_current(built_library, tmp_path):
    """CPU half of the matcher-level test: the program compiles and links here; the upstream-layout state_dict of the generator still prepares
    (matcher_engine.lightglue_entries) into exactly the logical entries the C program builds, and the oracle still gives the header's matches."""
    exe, _ = _build_c(tmp_path, built_library, "abi_lightglue_from_c.c", ["-isystem", "/opt/rocm/include", f"-I{REPO / 'tests' / 'abi'}"])
    assert exe.exists()
    import re
    import sys

    sys.path.insert(0, str(REPO))
    import numpy as np

    from oracle import make_abi_lightglue_expectation as gen

    header = (REPO / "tests" / "abi" / "abi_lightglue_expected.h").read_text()
    seed = int(re.search(r"#define ABI_LG_SEED (\d+)u", header).group(1))
    want0 = np.array(header.split("abi_lg_matches0[ABI_LG_N0] = {")[1].split("}")[0].split(","), dtype=np.int64)
    gen.check_preparation(seed)
    out = gen.run_oracle(seed)
    np.testing.assert_array_equal(out["matches0"][0].numpy(), want0)
    assert int((want0 > -1).sum()) >= 100 and len(want0) == gen.N0


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_lightglue_entry_points_driven_from_c(built_library, tmp_path):
    """gtsfm_pack_blob -> gtsfm_match_build_desc -> gtsfm_lg_workspace_bytes -> gtsfm_lg_forward from a C program that owns its device memory and
    stream: all 580 match indices identical to the oracle's numbers in abi_lightglue_expected.h, matching scores within 1e-4."""
    exe, env = _build_c(tmp_path, built_library, "abi_lightglue_from_c.c", ["-isystem", "/opt/rocm/include", f"-I{REPO / 'tests' / 'abi'}"])
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "abi_lightglue_from_c OK" in run.stdout


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_device_entry_point_driven_from_c(built_library, tmp_path):
    """hipMalloc / hipMemcpy / hipStreamCreate from C, gtsfm_sp_softmax_d2s through the C ABI on that stream, result checked in C."""
    exe, env = _build_c(tmp_path, built_library, "abi_device_from_c.c", ["-isystem", "/opt/rocm/include"])
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "abi_device_from_c OK" in run.stdout
