"""The score product of one pair (N x 256 -> N, image 1's descriptor rows as the weights) and of wider / narrower shapes, fraction of the fp32 MFMA peak.
GTSFM_GEMM_SUPERTILE=0 selects the row order of rounds 2-5 (every row tile streams all of W through its XCD's L2)."""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import bench  # noqa: E402
from gtsfm_amd.runtime import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [2048, 5000, 8192, 16384]:
    for _ in range(2):
        bench.measure_gemm_roofline(lib, dev, n, 256, n, reps=10)
    r = [bench.measure_gemm_roofline(lib, dev, n, 256, n, reps=20) for _ in range(3)]
    print(f"N={n:6d}: frac " + " ".join(f"{x['frac']:.3f}" for x in r) + "  ms " + " ".join(f"{x['avg_launch_ms']:.4f}" for x in r), flush=True)
