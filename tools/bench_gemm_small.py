"""Projection GEMMs at single-pair sizes: 128 x 128 tiles vs 64 x 64 tiles (GTSFM_GEMM_SMALL_BELOW)."""
import os
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import bench  # noqa: E402
from gtsfm_amd.runtime import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
for rows in (4096, 10240, 20480, 40960):
    for k, n in ((256, 768), (512, 512), (512, 256), (256, 512), (256, 256)):
        out = []
        for below in ("0", str(1 << 40)):
            os.environ["GTSFM_GEMM_SMALL_BELOW"] = below
            r = bench.measure_gemm_roofline(lib, dev, rows, k, n, reps=20)
            out.append(f"{r['avg_launch_ms'] * 1e3:7.1f} us ({r['frac']:.3f})")
        print(f"M={rows:6d} {k}->{n}: 128x128 {out[0]}   64x64 {out[1]}   tiles128={-(-rows // 128) * -(-n // 128)}", flush=True)
