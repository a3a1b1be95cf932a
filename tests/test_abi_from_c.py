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
        ["gcc", "-std=c99", "-Wall", "-Werror", f"-I{REPO / 'include'}", *extra, str(REPO / "tests" / "abi" / source), f"-L{lib_dir}", "-lgtsfm_amd",
         f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-o", str(exe)],
        check=True, capture_output=True, text=True,
    )
    return exe, dict(os.environ, LD_LIBRARY_PATH=f"{lib_dir}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_device_program_compiles_as_c(built_library, tmp_path):
    """CPU half of the device test: the C program that drives a device entry point compiles and links here (it runs under -m gpu)."""
    exe, _ = _build_c(tmp_path, built_library, "abi_device_from_c.c", ["-isystem", "/opt/rocm/include"])
    assert exe.exists()


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_device_entry_point_driven_from_c(built_library, tmp_path):
    """hipMalloc / hipMemcpy / hipStreamCreate from C, gtsfm_sp_softmax_d2s through the C ABI on that stream, result checked in C."""
    exe, env = _build_c(tmp_path, built_library, "abi_device_from_c.c", ["-isystem", "/opt/rocm/include"])
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "abi_device_from_c OK" in run.stdout
