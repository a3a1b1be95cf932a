"""CPU, build container only: the oracle restatements against the REFERENCE'S OWN model files, run live on inputs no committed fixture holds.

``tests/golden/*.npz`` were written by ``oracle/validate_against_reference.py`` from the same model files; this test repeats the comparison on fresh
seeds, odd sizes and ragged keypoint counts each time the CPU suite runs where ``/root/reference`` is mounted, so a change to ``oracle/`` cannot
drift from the reference between two regenerations of the fixtures. It is skipped on the GPU box (no reference there; nothing under ``-m gpu``,
``smoke()`` or ``bench.py`` reads it). Bit-exact: same torch, same operations, same order (thirdparty/SuperGluePretrainedNetwork/models/
superpoint.py:145-202, superglue.py:228-283)."""

import os
from pathlib import Path

import numpy as np
import pytest
import torch

REFERENCE = Path(os.environ.get("GTSFM_REFERENCE", "/root/reference"))
MODELS = REFERENCE / "thirdparty" / "SuperGluePretrainedNetwork" / "models"
pytestmark = pytest.mark.skipif(not (MODELS / "superpoint.py").exists(), reason="the reference tree is only mounted in the build container")


@pytest.fixture(scope="module")
def ref():
    from oracle import validate_against_reference as V

    return V


@pytest.mark.parametrize("h,w,seed", [(97, 131, 101), (64, 200, 102), (136, 120, 103)])
def test_superpoint_restatement_equals_the_reference_model_on_fresh_images(ref, h, w, seed):
    from gtsfm_amd.utils import synthetic
    from oracle import superpoint_oracle

    sd = synthetic.synthetic_superpoint_state_dict()
    model = ref.reference_superpoint(sd)
    img = superpoint_oracle.gray_u8_to_tensor(synthetic.synthetic_gray_image(h, w, seed))
    with torch.no_grad(), ref._force_align_corners():
        out = model({"image": img})
        ora = superpoint_oracle.superpoint_forward(sd, img)
    assert out["keypoints"][0].shape[0] > 20
    assert torch.equal(out["keypoints"][0], ora["keypoints"])
    assert torch.equal(out["scores"][0], ora["scores"])
    assert torch.equal(out["descriptors"][0], ora["descriptors"])


@pytest.mark.parametrize("n0,n1,iters,seed", [(150, 97, 20, 201), (33, 260, 100, 202), (1, 50, 20, 203)])
def test_superglue_restatement_equals_the_reference_model_on_fresh_pairs(ref, n0, n1, iters, seed):
    from gtsfm_amd.utils import synthetic
    from oracle import superglue_oracle

    sd = synthetic.synthetic_superglue_state_dict(num_layers=ref.SUPERGLUE_LAYERS_GOLDEN)
    model = ref.reference_superglue(sd, iters)
    shp0, shp1 = (240, 320), (200, 304)
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(n0, n1, shp0, shp1, seed=seed)
    data = {
        "keypoints0": torch.from_numpy(k0)[None], "keypoints1": torch.from_numpy(k1)[None],
        "descriptors0": torch.from_numpy(d0).T[None].contiguous(), "descriptors1": torch.from_numpy(d1).T[None].contiguous(),
        "scores0": torch.from_numpy(s0)[None], "scores1": torch.from_numpy(s1)[None],
        "image0": torch.empty((1, 1) + shp0), "image1": torch.empty((1, 1) + shp1),
    }
    with torch.no_grad():
        out = model(data)
        ora = superglue_oracle.superglue_forward(sd, data["keypoints0"], data["keypoints1"], data["scores0"], data["scores1"],
                                                 data["descriptors0"], data["descriptors1"], shp0, shp1, sinkhorn_iterations=iters)
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
        assert out[key].dtype == ora[key].dtype and torch.equal(out[key], ora[key]), key
