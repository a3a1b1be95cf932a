"""Sinkhorn row/column sweeps on their own at a given width: HBM GB/s per iteration (bench.py's roofline helper).

    python tools/bench_sweeps.py [N ...]          (GTSFM_SWEEP=lds selects the LDS-staged kernels for an A/B)
"""
import json
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import bench  # noqa: E402
from gtsfm_amd.runtime import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [2048, 5000]:
    pairs = max(1, min(32, int(2.2e9 / (4.0 * (n + 1) * (n + 4)))))
    print(json.dumps(bench.measure_sinkhorn_roofline(lib, dev, n, pairs)), flush=True)
