"""Compact per-kernel resource table (VGPRs, spills, occupancy, LDS) of one HIP source, as the compiler reports it.

    python tools/kernel_resources.py gtsfm_amd/csrc/sweep_kernels.hip [filter-substring]
"""
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from gtsfm_amd.csrc.build import FLAGS, HIPCC, PER_FILE_FLAGS  # noqa: E402

src = Path(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [HIPCC, *FLAGS, *PER_FILE_FLAGS.get(src.name, []), *sys.argv[3:], "-c", str(src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
    if "error" in line or "warning" in line:
        print(line)
print(f"{'kernel':70s} vgpr agpr spill scratch occ    lds")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:70]:70s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('spill', 0):5d} {r.get('scratch', 0):7d} {r.get('occ', 0):3d} {r.get('lds', 0):6d}")
