"""LightGlue's assignment stage alone (gtsfm_lg_assignment_f32) at the benchmark's sizes: the double log-softmax sweeps and the match extraction as
GB/s of algorithmic bytes (4 N^2 B per pair and pass), the extraction under both wave tiers (GTSFM_EXTRACT_WAVES=4 / 8).

    python tools/bench_assign.py"""
import os
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import bench  # noqa: E402
from gtsfm_amd.runtime import lib as L  # noqa: E402

lib, dev = L.load(), torch.device("cuda:0")
for n, pairs in ((5000, 16), (5000, 1), (2048, 32), (1024, 32)):
    for waves in ("8", "4"):
        os.environ["GTSFM_EXTRACT_WAVES"] = waves
        for r in bench.measure_lg_assignment_roofline(lib, dev, n, pairs, reps=10):
            print(f"N={n} pairs={pairs} extract_waves={waves} {r['kernel'][:40]:40s} {r['avg_launch_ms']:.4f} ms  {r['achieved']:.0f} GB/s  frac {r['frac']:.3f}", flush=True)
os.environ.pop("GTSFM_EXTRACT_WAVES", None)
r = bench.measure_layernorm_roofline(lib, dev, 163840, reps=10)
print("layernorm_gelu 163840 rows", r["avg_launch_ms"], "ms", r["achieved"], "GB/s")
