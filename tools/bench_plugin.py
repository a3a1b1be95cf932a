"""The per-call plugin API on its own (bench.py's secondary.plugin_api leg): detect_and_describe per image, match per pair, numpy in / out.

    python tools/bench_plugin.py [--keypoints 5000] [--matcher lightglue|superglue]"""
import argparse
import json
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import bench  # noqa: E402
from gtsfm_amd.runtime import matcher_engine as ME  # noqa: E402
from gtsfm_amd.runtime.pipeline import FrontEndPipeline  # noqa: E402
from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine  # noqa: E402
from gtsfm_amd.utils import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--keypoints", type=int, nargs="+", default=[5000, 2048])
ap.add_argument("--matcher", default="lightglue")
a = ap.parse_args()
dev = torch.device("cuda:0")
det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), dev)
mt = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(), dev) if a.matcher == "lightglue" else ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(), dev)
views = synthetic.synthetic_overlapping_views(6, 1024, 1024, 1000)
for k in a.keypoints:
    args = argparse.Namespace(keypoints=k, matcher=a.matcher, sinkhorn=20, images=bench.fewest_images_for(250), pairs=250)
    pipe = FrontEndPipeline(det, mt, max_keypoints=k, pair_chunk=bench.default_pair_chunk(k), num_streams=1)
    print(json.dumps(bench.plugin_api_rate(args, pipe, views, dev, 1024, 1024)), flush=True)
    # the same single-pair launch sequence back to back without host synchronisation: GPU time per pair when launches overlap execution
    feats = pipe.detect(torch.from_numpy(views[:2]).to(dev))
    kp = feats["xy"][:2].reshape(-1, 2).contiguous()
    de = feats["descriptors"][:2].reshape(-1, 256).contiguous()
    sc = feats["scores"][:2].reshape(-1).contiguous()
    call = (lambda: mt.match_batch(kp, de, [k], [k], [[1024, 1024, 1024, 1024]])) if a.matcher == "lightglue" else (lambda: mt.match_batch(kp, sc, de, [k], [k], [[1024, 1024, 1024, 1024]]))
    import time
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    t_enq = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        call(); torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / 10
    print(json.dumps({"keypoints": k, "enqueue_ms_per_pair": round(t_enq * 1e3, 3), "pipelined_ms_per_pair": round(t_all * 1e3, 3), "synchronous_ms_per_pair": round(t_sync * 1e3, 3)}), flush=True)
