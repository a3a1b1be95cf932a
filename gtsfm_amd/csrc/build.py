"""Build libgtsfm_amd.so (hand-written HIP kernels + C ABI) for gfx950, in-tree next to the package.

hipcc cross-compiles without a GPU. The shared library is git-ignored but travels to the GPU box with the snapshot.
Usage: python -m gtsfm_amd.csrc.build [--force]
"""

from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
LIB = PKG / "libgtsfm_amd.so"
OBJ_DIR = CSRC / "build"
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]
# attention: the softmax row maximum is 32 fmaxf per lane and tile; without NaN semantics to preserve, the compiler
# drops the operand canonicalisation (v_max x, x) and pairs them into v_max3 (55 -> 24 instructions per tile).
# Masked scores are -inf, not NaN; infinities keep their meaning.
PER_FILE_FLAGS = {"attention_kernels.hip": ["-fno-honor-nans"]}


def sources():
    return sorted(CSRC.glob("*.hip"))


def _included_headers(src: Path, known: dict) -> list:
    """Project headers `src` includes, transitively (by file name; sorted)."""
    import re

    seen, todo = {}, [src]
    while todo:
        f = todo.pop()
        for name in re.findall(r'#include\s+"(?:[^"]*/)?([^"/]+)"', f.read_text()):
            if name in known and name not in seen:
                seen[name] = known[name]
                todo.append(known[name])
    return [seen[k] for k in sorted(seen)]


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [PKG.parent / "include" / "gtsfm_amd.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    stamp = OBJ_DIR / "digest.txt"
    digest = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == digest:
        return LIB
    OBJ_DIR.mkdir(exist_ok=True)

    headers = sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "gtsfm_amd.h"]

    def compile_one(src: Path) -> Path:
        """One object per source, rebuilt only when the source, a header or the flags changed (dense_kernels.hip alone
        takes minutes)."""
        obj = OBJ_DIR / (src.stem + ".o")
        flags = [*FLAGS, *PER_FILE_FLAGS.get(src.name, [])]
        h = hashlib.sha256(" ".join(flags).encode())
        for f in [src, *_included_headers(src, {x.name: x for x in headers})]:
            h.update(f.name.encode())
            h.update(f.read_bytes())
        obj_stamp = OBJ_DIR / (src.stem + ".digest")
        if not force and obj.exists() and obj_stamp.exists() and obj_stamp.read_text() == h.hexdigest():
            return obj
        cmd = [HIPCC, *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        obj_stamp.write_text(h.hexdigest())
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(digest)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
