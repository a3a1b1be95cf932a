"""Probe: how much of the chip a single-pair launch sequence leaves idle -- T host threads, each with its own matcher engine (own
workspace) and HIP stream, matching single pairs back to back; aggregate pairs/s against one thread.

    python tools/bench_plugin_threads.py [--keypoints 5000 2048] [--threads 1 2 3]"""
import argparse
import json
import sys
import threading
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from gtsfm_amd.runtime import matcher_engine as ME  # noqa: E402
from gtsfm_amd.runtime.pipeline import FrontEndPipeline  # noqa: E402
from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine  # noqa: E402
from gtsfm_amd.utils import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--keypoints", type=int, nargs="+", default=[5000, 2048])
ap.add_argument("--threads", type=int, nargs="+", default=[1, 2, 3])
ap.add_argument("--calls", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), dev)
sd = synthetic.synthetic_lightglue_state_dict()
engines = [ME.LightGlueEngine(sd, dev) for _ in range(max(a.threads))]
views = synthetic.synthetic_overlapping_views(2, 1024, 1024, 1000)
for k in a.keypoints:
    pipe = FrontEndPipeline(det, engines[0], max_keypoints=k, pair_chunk=1, num_streams=1)
    feats = pipe.detect(torch.from_numpy(views[:2]).to(dev))
    kp = feats["xy"][:2].reshape(-1, 2).contiguous()
    de = feats["descriptors"][:2].reshape(-1, 256).contiguous()
    torch.cuda.synchronize()

    def worker(engine, stream, n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                engine.match_batch(kp, de, [k], [k], [[1024, 1024, 1024, 1024]])
            stream.synchronize()

    for nt in a.threads:
        streams = [torch.cuda.Stream(dev) for _ in range(nt)]
        for e, s in zip(engines, streams):  # warm-up (workspaces, descriptor blocks)
            worker(e, s, 2)
        torch.cuda.synchronize()
        ths = [threading.Thread(target=worker, args=(engines[i], streams[i], a.calls)) for i in range(nt)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"keypoints": k, "threads": nt, "pairs_per_s": round(nt * a.calls / dt, 1), "ms_per_pair_per_thread": round(dt / a.calls * 1e3, 2)}), flush=True)
