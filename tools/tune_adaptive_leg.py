"""Which synthetic LightGlue heads make adaptive depth AND width fire with a spread over pairs (bench.py's `lightglue_adaptive_realistic` leg)?

    python tools/tune_adaptive_leg.py [keypoints] [pairs]

For a small grid of (delta_gain, conf_bias, conf_gain, match_bias, match_gain) runs the matcher over the first pairs of a mixed scene
(synthetic.synthetic_mixed_scene: three canvases of different texture scale, overlaps 100 % .. 30 % and unrelated pairs) and prints the
histogram of stop layers and the share of keypoints alive at the assignment."""
import itertools
import os
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from gtsfm_amd import parallel  # noqa: E402
from gtsfm_amd.runtime import matcher_engine as ME  # noqa: E402
from gtsfm_amd.runtime.pipeline import FrontEndPipeline  # noqa: E402
from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine  # noqa: E402
from gtsfm_amd.utils import synthetic  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 66
dev = torch.device("cuda:0")
det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), dev)
n = int(os.environ.get("VIEWS", "12"))
import os
CANVASES = int(os.environ.get("CANVASES", "6"))
views = torch.from_numpy(synthetic.synthetic_mixed_scene(n, 1024, 1024, canvases=CANVASES)).to(dev)
pairs = parallel.exhaustive_pairs(n)[:npairs]
shapes = [(1024, 1024)] * n
feats = None
KEYS = ("delta_gain", "conf_bias", "conf_gain", "conf_ramp", "match_bias", "match_gain")
grid = [dict(zip(KEYS, v)) for v in itertools.product((0.25,), (-2.0, -1.0, 0.0, 1.0), (5.0, 10.0, 20.0, 40.0), (0.4, 0.8), (-4.6,), (40.0,))]
if len(sys.argv) > 3:
    grid = [dict(zip(KEYS, map(float, a.split(",")))) for a in sys.argv[3:]]
same = np.array([(i % CANVASES) == (j % CANVASES) for i, j in pairs])  # pairs of one canvas (they overlap) against pairs across canvases (nothing shared)
for g in grid:
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(conf_shared_direction=True, **g), dev)
    pipe = FrontEndPipeline(det, eng, max_keypoints=k, pair_chunk=16 if k > 2560 else 32)
    if feats is None:
        feats = pipe.detect(views)
        print("keypoints per view", feats["count"].tolist(), flush=True)
    res = pipe.match(feats, pairs, shapes)
    stop = torch.cat([r["stop"] for r in res]).cpu().numpy()
    kept = torch.cat([r["kept"] for r in res]).cpu().numpy().reshape(-1)
    nm = np.array([int((r["matches"] > -1).sum()) // 2 for r in res]).sum() / len(pairs)
    hist = np.bincount(stop, minlength=10)[1:].tolist()
    hs = np.bincount(stop[same], minlength=10)[1:].tolist()
    print(",".join(str(g[key]) for key in KEYS), "| stop 1..9:", hist, "same-canvas:", hs, "| kept min/med/max %.2f %.2f %.2f" % (kept.min() / k, np.median(kept) / k, kept.max() / k),
          "| matches/pair %.0f" % nm, flush=True)
