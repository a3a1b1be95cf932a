"""Verifier-stage launches at workload shapes for rocprofv3 (tools/prof_r02.sh): 1000 pairs x 163 matches (the headline
step's match lists), 256 pairs x 718 and x 2048 matches with outliers (up to 1280 hypotheses per pair), both modes."""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from gtsfm_amd.runtime.verifier_engine import VerifierEngine  # noqa: E402
from gtsfm_amd.utils import synthetic  # noqa: E402
from test_verifier_gpu import _batch  # noqa: E402

dev = torch.device("cuda:0")
eng = VerifierEngine(dev)
for pairs, m, outliers in ((1000, 163, 0.02), (256, 718, 0.5), (256, 2048, 0.6)):
    scenes = [synthetic.synthetic_two_view_matches(m, outliers, 0.5, seed=k % 64) for k in range(pairs)]
    batch = _batch(scenes, dev)
    for use_intrinsics in (True, False):
        for _ in range(3):
            out = eng.verify_batch(*batch, 4.0, list(range(pairs)), use_intrinsics=use_intrinsics)
        torch.cuda.synchronize()
        st = out["stats"].float().mean(0).tolist()
        print(f"pairs {pairs} matches {m} outliers {outliers} essential {use_intrinsics}: mean inliers {st[0]:.1f}, hypotheses {st[1]:.0f}")
