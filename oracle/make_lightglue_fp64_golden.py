"""FLOAT64 arbiter for the LightGlue path -- TEST INFRASTRUCTURE (never imported by the product path).

LightGlue's source is absent from the reference snapshot (``/root/reference/.gitmodules:1-3``), so its parity is pinned only towards
``oracle/lightglue_oracle.py`` (a restatement of published upstream) and the HuggingFace port. What CAN be settled offline is the
arithmetic question VERDICT round 3 raised: at the 5000-keypoint cap the HIP path's scores sit 5e-5 from the fp32 oracle's against a
1e-4 contract -- whose error is that? This script runs the restatement in float64 (same code, ``desc.dtype`` decides) next to its
float32 form and stores both, so that ``tests/test_lightglue_fp64_arbiter_gpu.py`` can report HIP-vs-fp64 and oracle-fp32-vs-fp64 side
by side: at the benchmark's N = 2048 after 1, 3, 5, 7 and 9 layers (how the error grows with depth), at 5000 x 4800 at full depth, and
at 5000 x 4800 with a peaked assignment (final projection gain 48 instead of 16: near-one-hot score rows, similarities up to +-100). Early stopping and point pruning are
off in every case so that all runs execute the same layers. Inputs are regenerated from seeds; outputs are stored.

Run (build container, ~5 min of CPU):  python oracle/make_lightglue_fp64_golden.py"""

import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from gtsfm_amd.utils import synthetic  # noqa: E402
from oracle import lightglue_oracle as lgo  # noqa: E402

SHAPE = (1024, 1024)
# name -> (weight kwargs, layers, n0, n1, feature seed)
CASES = {
    **{f"n2048_depth{depth}": ({}, depth, 2048, 2048, 61) for depth in (1, 3, 5, 7, 9)},
    "cap5000x4800_depth9": ({}, 9, 5000, 4800, 62),
    "cap5000x4800_peaked": ({"final_gain": 48.0, "match_gain": 12.0}, 9, 5000, 4800, 63),  # similarities three times as steep: near-one-hot score rows
}


def run(sd, k0, d0, k1, d1, dtype):
    t = lambda a: torch.from_numpy(a).to(dtype)[None]  # noqa: E731
    with torch.no_grad():
        return lgo.lightglue_forward(sd, t(k0), t(k1), t(d0), t(d1), SHAPE, SHAPE, depth_confidence=-1.0, width_confidence=-1.0, pruning_threshold=None)


def main() -> None:
    torch.set_num_threads(8)
    out = {"cases": json.dumps({k: {"weight_kwargs": v[0], "layers": v[1], "n0": v[2], "n1": v[3], "seed": v[4]} for k, v in CASES.items()}),
           "shape": np.array(SHAPE)}
    only = sys.argv[1:]  # optional: regenerate the named cases only (the others are kept from the existing file)
    path = REPO / "tests" / "golden" / "lightglue_fp64_arbiter.npz"
    if only and path.exists():
        old = np.load(path)
        out.update({k: old[k] for k in old.files if k not in ("cases", "shape") and not any(k.startswith(n + "_") for n in only)})
    for name, (kwargs, layers, n0, n1, seed) in CASES.items():
        if only and name not in only:
            continue
        sd = synthetic.synthetic_lightglue_state_dict(num_layers=layers, **kwargs)
        k0, _, d0, k1, _, d1, _ = synthetic.synthetic_pair_features(n0, n1, SHAPE, SHAPE, seed=seed)
        t0 = time.time()
        r64 = run(sd, k0, d0, k1, d1, torch.float64)
        r32 = run(sd, k0, d0, k1, d1, torch.float32)
        assert int(r64["stop"]) == layers == int(r32["stop"])
        for side in (0, 1):
            out[f"{name}_matches{side}_f64"] = r64[f"matches{side}"][0].numpy().astype(np.int16)
            out[f"{name}_scores{side}_f64"] = r64[f"matching_scores{side}"][0].numpy().astype(np.float64)
            out[f"{name}_matches{side}_f32"] = r32[f"matches{side}"][0].numpy().astype(np.int16)
            out[f"{name}_scores{side}_f32"] = r32[f"matching_scores{side}"][0].numpy().astype(np.float32)
        same = all(np.array_equal(out[f"{name}_matches{s}_f64"], out[f"{name}_matches{s}_f32"]) for s in (0, 1))
        err = max(float(np.abs(out[f"{name}_scores{s}_f64"] - out[f"{name}_scores{s}_f32"]).max()) for s in (0, 1))
        print(f"{name}: {int((r64['matches0'][0] > -1).sum())} matches; fp32 oracle vs fp64: matches {'equal' if same else 'DIFFER'}, max |dscore| {err:.3e} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
