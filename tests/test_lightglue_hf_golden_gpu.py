"""LightGlue on the HIP path against golden vectors produced by an INDEPENDENT implementation: the HuggingFace
``transformers`` port of upstream cvg/LightGlue (``oracle/make_lightglue_hf_golden.py``; upstream's source is not in the
reference snapshot, so no reference-made fixture can exist -- PARITY UNPINNED towards the reference, pinned towards this third
party). Matches identical, matching scores within 1e-4, for plain / early-stop / pruning / early-stop + pruning cases and the
benchmark's keypoint count (N = 2048, full depth). The CPU test keeps the restatement on the same fixtures."""

import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gtsfm_amd.utils import synthetic
from oracle import lightglue_oracle as lgo

CASES = ["plain", "early_stop", "pruning", "early_stop_pruning", "n2048_full_depth"]
TOL = 1e-4


def _load(name):
    g = np.load(GOLDEN / f"lightglue_hf_{name}.npz")
    return g, synthetic.synthetic_lightglue_state_dict(**json.loads(str(g["weight_kwargs"])))


@pytest.mark.parametrize("name", CASES)
def test_restatement_equals_the_hf_fixture(name):
    g, sd = _load(name)
    t = torch.from_numpy
    hw = tuple(int(v) for v in g["image_hw"])
    with torch.no_grad():
        ora = lgo.lightglue_forward(sd, t(g["k0"])[None], t(g["k1"])[None], t(g["d0"])[None], t(g["d1"])[None], hw, hw, pruning_threshold=-1)
    for side in (0, 1):
        np.testing.assert_array_equal(ora[f"matches{side}"][0].numpy(), g["matches"][side])
        np.testing.assert_allclose(ora[f"matching_scores{side}"][0].numpy(), g["mscores"][side], rtol=0, atol=TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_path_equals_the_hf_fixture(gpu_device, name):
    from gtsfm_amd.runtime.matcher_engine import LightGlueEngine

    g, sd = _load(name)
    hw = tuple(int(v) for v in g["image_hw"])
    res = LightGlueEngine(sd, gpu_device).match_pair(g["k0"], g["d0"], g["k1"], g["d1"], hw, hw, pruning_threshold=-1)  # the port always prunes
    assert int((g["matches"][0] > -1).sum()) > 20
    for side in (0, 1):
        np.testing.assert_array_equal(res[f"matches{side}"], g["matches"][side])
        np.testing.assert_allclose(res[f"matching_scores{side}"], g["mscores"][side], rtol=0, atol=TOL)
