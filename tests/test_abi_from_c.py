"""The drop-in boundary is a C ABI: ``include/gtsfm_amd.h`` has to be valid plain C and ``libgtsfm_amd.so`` has to link and
run from a C program. gcc compiles ``tests/abi/abi_from_c.c`` with ``-std=c99 -pedantic -Werror`` against the header and the
in-tree library and runs it (host-only entry points and argument checks: no GPU needed)."""

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
