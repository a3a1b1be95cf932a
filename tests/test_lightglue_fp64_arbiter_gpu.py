"""-m gpu: LightGlue on the HIP path against a FLOAT64 run of the restatement (``oracle/make_lightglue_fp64_golden.py``), with the
float32 run of the same restatement beside it. LightGlue cannot be pinned to the reference (its source is not in the snapshot); this is
the arithmetic half of the question: wherever the HIP path and the fp32 oracle disagree in the fifth digit, the float64 result says whose
round-off it is. Cases: N = 2048 after 1 / 3 / 5 / 7 / 9 layers (error growth with depth), 5000 x 4800 at full depth (GTSfM's cap), and
5000 x 4800 with a peaked assignment. Both arithmetics of the attention kernel are held to it: exact fp32 (the default) and the opt-in
bf16x3. Requirements: matches equal to the float64 matches except where the float64 score sits within 1e-4 of the 0.1 filter threshold;
scores within 1e-4 of float64 (within twice the fp32 oracle's own distance where that alone exceeds 1e-4: the peaked case); and the
HIP error no larger than 3 x the fp32 oracle's own error + 2e-5. The table it prints is quoted
in DESIGN.md section 5 (profiles/r04_lightglue_fp64_arbiter.txt)."""

import json

import numpy as np
import pytest

from conftest import GOLDEN
from gtsfm_amd.utils import synthetic

pytestmark = pytest.mark.gpu

TOL = 1e-4
FILTER_THRESHOLD = 0.1  # upstream LightGlue default_conf["filter_threshold"]


@pytest.fixture(scope="module")
def arbiter():
    path = GOLDEN / "lightglue_fp64_arbiter.npz"
    if not path.exists():
        pytest.fail("tests/golden/lightglue_fp64_arbiter.npz is missing: run oracle/make_lightglue_fp64_golden.py")
    g = np.load(path)
    return g, json.loads(str(g["cases"])), tuple(int(v) for v in g["shape"])


def _case_names():
    path = GOLDEN / "lightglue_fp64_arbiter.npz"
    return sorted(json.loads(str(np.load(path)["cases"]))) if path.exists() else ["missing"]


@pytest.mark.parametrize("math", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("name", _case_names())
def test_hip_lightglue_against_float64(gpu_device, arbiter, monkeypatch, name, math):
    from gtsfm_amd.runtime.matcher_engine import LightGlueEngine

    g, cases, shape = arbiter
    c = cases[name]
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", math)
    sd = synthetic.synthetic_lightglue_state_dict(num_layers=c["layers"], **c["weight_kwargs"])
    k0, _, d0, k1, _, d1, _ = synthetic.synthetic_pair_features(c["n0"], c["n1"], shape, shape, seed=c["seed"])
    eng = LightGlueEngine(sd, gpu_device)
    eng.image_cache_capacity = 0
    res = eng.match_pair(k0, d0, k1, d1, shape, shape, depth_confidence=-1.0, width_confidence=-1.0, pruning_threshold=None)
    assert res["stop"] == c["layers"]
    err_hip = err_f32 = 0.0
    flips = 0
    for side in (0, 1):
        m64, s64 = g[f"{name}_matches{side}_f64"].astype(np.int64), g[f"{name}_scores{side}_f64"]
        s32 = g[f"{name}_scores{side}_f32"].astype(np.float64)
        got_m, got_s = res[f"matches{side}"].astype(np.int64), res[f"matching_scores{side}"].astype(np.float64)
        differ = got_m != m64
        # a match may fall on the other side of the filter threshold only when float64 itself puts its score within 1e-4 of it
        assert np.all(np.abs(np.maximum(s64[differ], got_s[differ]) - FILTER_THRESHOLD) < TOL), (name, side, int(differ.sum()))
        flips += int(differ.sum())
        same = ~differ
        err_hip = max(err_hip, float(np.abs(got_s[same] - s64[same]).max()))
        err_f32 = max(err_f32, float(np.abs(s32 - s64).max()))
    nm = int((g[f"{name}_matches0_f64"] > -1).sum())
    print(f"ARBITER {name:24s} {math:7s} matches {nm:5d}  threshold flips {flips}  max |score - float64|: HIP {err_hip:.2e}   fp32 oracle {err_f32:.2e}")
    assert nm > 50
    # the contract's 1e-4 -- unless the fp32 restatement ITSELF is further than that from float64 (the peaked case: 1.4e-4 on the
    # CPU), where no fp32 implementation can be held to it: then twice the restatement's own error
    assert err_hip < max(TOL, 2.0 * err_f32), (err_hip, err_f32)
    assert err_hip <= 3.0 * err_f32 + 2e-5, (err_hip, err_f32)
