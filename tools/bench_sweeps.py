"""Sinkhorn row/column sweeps on their own at a given width: HBM GB/s per iteration (bench.py's roofline helper).

    python tools/bench_sweeps.py [N | N:pairs ...]   (GTSFM_SWEEP=lds selects the LDS-staged kernels, GTSFM_SWEEP_NT_MB=<MiB> moves the
                                                     size above which the score matrices are read nontemporally; default 256)
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
for arg in sys.argv[1:] or ["2048", "5000"]:
    n, _, given = arg.partition(":")
    n = int(n)
    pairs = int(given) if given else max(1, min(32, int(2.2e9 / (4.0 * (n + 1) * (n + 4)))))
    print(json.dumps(bench.measure_sinkhorn_roofline(lib, dev, n, pairs)), flush=True)
