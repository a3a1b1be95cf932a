"""ONE thread, resident images: `LightGlueMatcher.match` / `SuperGlueMatcher.match` call after call at a given keypoint count -- the part of bench.py's
plugin_api leg that `pairs_per_s_match_only_resident` reports, on its own so that a rocprofv3 --kernel-trace --stats of it holds a single pair's kernels
only (tools/bench_plugin.py also runs 2 / 3 worker threads and the bf16x3 calls: their overlapping launches inflate the per-kernel averages).

    python tools/bench_plugin_resident.py [--keypoints 5000] [--matcher lightglue] [--calls 24]"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from gtsfm_amd import parallel  # noqa: E402
from gtsfm_amd.common.image import Image  # noqa: E402
from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor  # noqa: E402
from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher  # noqa: E402
from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher  # noqa: E402
from gtsfm_amd.utils import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--keypoints", type=int, default=5000)
ap.add_argument("--matcher", default="lightglue")
ap.add_argument("--calls", type=int, default=24)
a = ap.parse_args()
views = synthetic.synthetic_overlapping_views(4, 1024, 1024, 1000)
with tempfile.TemporaryDirectory() as tmp:
    torch.save(synthetic.synthetic_superpoint_state_dict(), f"{tmp}/sp.pth")
    det = SuperPointDetectorDescriptor(max_keypoints=a.keypoints, weights_path=f"{tmp}/sp.pth")
    if a.matcher == "superglue":
        torch.save(synthetic.synthetic_superglue_state_dict(), f"{tmp}/sg.pth")
        mt = SuperGlueMatcher(weights_path=f"{tmp}/sg.pth")
    else:
        torch.save(synthetic.synthetic_lightglue_state_dict(), f"{tmp}/lg.pth")
        mt = LightGlueMatcher("superpoint", weights_path=f"{tmp}/lg.pth")
    feats = [det.detect_and_describe(Image(value_array=v)) for v in views]
    pairs = parallel.exhaustive_pairs(4)
    shape = (1024, 1024, 1)
    for i, j in pairs:  # every image resident afterwards
        mt.match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], shape, shape)
    torch.cuda.synchronize()
    each = []
    for c in range(a.calls):
        i, j = pairs[c % len(pairs)]
        t0 = time.perf_counter()
        out = mt.match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], shape, shape)
        each.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"keypoints": a.keypoints, "matcher": a.matcher, "calls": a.calls, "match_ms_per_pair_resident": round(float(np.median(each)), 3),
                      "min_max_ms": [round(min(each), 3), round(max(each), 3)], "pairs_per_s_match_only_resident": round(1e3 / float(np.median(each)), 1),
                      "matches_last_pair": int(len(out))}), flush=True)
