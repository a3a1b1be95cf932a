"""The projection GEMMs at the batched launch shapes (fraction of the fp32 MFMA peak), e.g. under GTSFM_LIB=<variant>."""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import bench  # noqa: E402
from gtsfm_amd.runtime import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
for rows in (32768, 163840):
    for k, n in ((256, 768), (512, 512), (512, 256), (256, 512), (256, 256)):
        for _ in range(3):
            bench.measure_gemm_roofline(lib, dev, rows, k, n, reps=10)
        v = [bench.measure_gemm_roofline(lib, dev, rows, k, n, reps=10)["frac"] for _ in range(3)]
        print(f"M={rows:6d} {k}->{n}: " + " ".join(f"{x:.3f}" for x in v), flush=True)
