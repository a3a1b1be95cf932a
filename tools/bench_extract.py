"""Match extraction at the cap: one LightGlue / SuperGlue layer over 16 pairs of 5000 keypoints, eight waves per row (default) against the
four-wave tier (GTSFM_EXTRACT_WAVES=4); the difference between the two timings is the extract_rows kernel's.

    python tools/bench_extract.py [lightglue|superglue]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from gtsfm_amd.runtime import matcher_engine as ME  # noqa: E402
from gtsfm_amd.utils import synthetic  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "lightglue"
dev = torch.device("cuda:0")
n, pairs = 5000, 16
rng = np.random.default_rng(0)
t = 2 * n * pairs
kp = torch.from_numpy(rng.uniform(0, 1024, (t, 2)).astype(np.float32)).to(dev)
sc = torch.from_numpy(rng.uniform(0, 1, t).astype(np.float32)).to(dev)
de = torch.nn.functional.normalize(torch.randn((t, 256), device=dev), dim=1)
hw = [[1024, 1024, 1024, 1024]] * pairs
if which == "lightglue":
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=1), dev)
    call = lambda: eng.match_batch(kp, de, [n] * pairs, [n] * pairs, hw, pruning_threshold=None)  # noqa: E731
else:
    eng = ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(num_layers=2), dev)
    call = lambda: eng.match_batch(kp, sc, de, [n] * pairs, [n] * pairs, hw, sinkhorn_iterations=0)  # noqa: E731
for rnd in range(2):
    for waves in ("8", "4"):
        os.environ["GTSFM_EXTRACT_WAVES"] = waves
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        e1.synchronize()
        print(which, "waves", waves, round(e0.elapsed_time(e1) / 5, 3), "ms per 16-pair call", flush=True)
