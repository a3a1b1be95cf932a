"""CPU baseline per SURVEY.md section 8(d): the REFERENCE'S OWN model files (thirdparty/SuperGluePretrainedNetwork/models/
superpoint.py, superglue.py, imported by path from /root/reference) through the restated wrapper path of
gtsfm/frontend/detector_descriptor/superpoint.py:73-91 (uint8 gray -> /255 -> SuperPoint.forward -> numpy -> top-k) and
gtsfm/frontend/matcher/superglue_matcher.py:75-113 (SuperGlue.forward at 20 and 100 Sinkhorn iterations), fp32,
torch.set_num_threads(cores), 1 warm-up + 5 timed repetitions, MEDIAN, at N in {1024, 2048, 5000}. LightGlue has no reference
source in the snapshot: its line times oracle/lightglue_oracle.py and is labelled "port".

Runs only where /root/reference is mounted (the build container); writes profiles/r03_cpu_baseline.json, which bench.py quotes
as ``cpu_baseline.reference_run``.   python tools/cpu_baseline.py [--reps 5] [--sizes 1024 2048 5000]"""

from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

from gtsfm_amd.utils import synthetic  # noqa: E402
from oracle import lightglue_oracle  # noqa: E402
from oracle import validate_against_reference as ref  # noqa: E402


def timed(fn, reps: int):
    fn()  # warm-up
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--sizes", type=int, nargs="+", default=[1024, 2048, 5000])
    ap.add_argument("--out", default=str(REPO / "profiles" / "r03_cpu_baseline.json"))
    args = ap.parse_args()
    if not ref.MODELS.exists():
        raise SystemExit(f"{ref.MODELS} not found: this tool needs the reference mounted")
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    h = w = 1024
    out = {
        "protocol": f"SURVEY.md section 8(d): reference model files, fp32, torch {torch.__version__} CPU, torch.set_num_threads({cores}), 1 warm-up + {args.reps} reps, median",
        "host": {"cores": cores, "machine": "the build container (the GPU box's host is timed by bench.py's cpu_baseline on a single sample)"},
        "image": f"{h}x{w} uint8 gray (gtsfm_amd.utils.synthetic.synthetic_overlapping_views, seed 1000)", "weights": "seeded synthetic",
    }
    sp_sd = synthetic.synthetic_superpoint_state_dict()
    sp = ref.reference_superpoint(sp_sd)
    view = synthetic.synthetic_overlapping_views(2, h, w, 1000)

    def detect(gray):
        # gtsfm/frontend/detector_descriptor/superpoint.py:73-91 restated: /255, model, numpy, top-k by response
        with torch.no_grad(), ref._force_align_corners():
            res = sp({"image": torch.from_numpy(gray.astype(np.float32) / 255.0)[None, None]})
        kp, sc, de = res["keypoints"][0].numpy(), res["scores"][0].numpy(), res["descriptors"][0].numpy().T
        sel = np.argpartition(-sc, min(5000, len(sc) - 1))[:5000] if len(sc) > 5000 else np.arange(len(sc))
        return kp[sel], sc[sel], de[sel]

    med, ts = timed(lambda: detect(view[0]), args.reps)
    out["superpoint"] = {"kind": "reference", "s_per_image": round(med, 4), "images_per_s": round(1.0 / med, 3), "samples_s": [round(t, 4) for t in ts]}
    print("superpoint", out["superpoint"], flush=True)
    f0, f1 = detect(view[0]), detect(view[1])
    sg_sd = synthetic.synthetic_superglue_state_dict()
    lg_sd = synthetic.synthetic_lightglue_state_dict()
    out["superglue"], out["lightglue"] = {}, {}
    T = torch.from_numpy
    for n in args.sizes:
        order0, order1 = np.sort(np.argsort(-f0[1])[:n]), np.sort(np.argsort(-f1[1])[:n])
        (k0, s0, d0), (k1, s1, d1) = [a[order0] for a in f0], [a[order1] for a in f1]
        for iters in (20, 100):
            model = ref.reference_superglue(sg_sd, iters)
            data = {  # gtsfm/frontend/matcher/superglue_matcher.py:75-102 restated
                "keypoints0": T(k0)[None], "keypoints1": T(k1)[None], "scores0": T(s0)[None], "scores1": T(s1)[None],
                "descriptors0": T(d0).T[None].contiguous(), "descriptors1": T(d1).T[None].contiguous(),
                "image0": torch.empty((1, 1, h, w)), "image1": torch.empty((1, 1, h, w)),
            }

            def run(model=model, data=data):
                with torch.no_grad():
                    return model(data)

            reps = args.reps
            med, ts = timed(run, reps)
            key = f"n{n}_sinkhorn{iters}"
            out["superglue"][key] = {"kind": "reference", "s_per_pair": round(med, 4), "pairs_per_s": round(1.0 / med, 4), "reps": reps,
                                     "independent_pairs_per_s": round(1.0 / (med + 2 * out["superpoint"]["s_per_image"]), 4), "samples_s": [round(t, 4) for t in ts]}
            print("superglue", key, out["superglue"][key], flush=True)

        def run_lg():
            with torch.no_grad():
                return lightglue_oracle.lightglue_forward(lg_sd, T(k0)[None], T(k1)[None], T(d0)[None], T(d1)[None], (h, w), (h, w))

        reps = args.reps
        med, ts = timed(run_lg, reps)
        out["lightglue"][f"n{n}"] = {"kind": "port", "s_per_pair": round(med, 4), "pairs_per_s": round(1.0 / med, 4), "reps": reps,
                                     "independent_pairs_per_s": round(1.0 / (med + 2 * out["superpoint"]["s_per_image"]), 4), "samples_s": [round(t, 4) for t in ts],
                                     "note": "oracle/lightglue_oracle.py (no LightGlue source in the reference snapshot); synthetic confidence heads never stop early: 9 of 9 layers"}
        print("lightglue", n, out["lightglue"][f"n{n}"], flush=True)
        Path(args.out).write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
