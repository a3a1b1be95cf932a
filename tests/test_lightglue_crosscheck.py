"""CPU: the LightGlue oracle against an INDEPENDENT implementation.

LightGlue's source is absent from the reference (``/root/reference/.gitmodules:1-3``: un-vendored ``cvg/LightGlue``
submodule; call sites ``gtsfm/frontend/matcher/lightglue_matcher.py:37-110``), so ``oracle/lightglue_oracle.py`` cannot be
pinned on the reference: PARITY UNPINNED, every LightGlue claim reads "== restatement". As hard a pin as this snapshot
allows: the HuggingFace ``transformers`` port of upstream LightGlue (``transformers/models/lightglue/modeling_lightglue.py``,
transformers 5.15.0 in this image -- a converted copy of upstream, NOT the reference; different weight names, q/k/v
un-fused, always-on pruning) must produce the same matches, matching scores and per-keypoint prune counters as the
restatement for the same weights: plain, early-stopping and pruning-active cases. Skipped where transformers is absent.
"""

import numpy as np
import pytest
import torch

from gtsfm_amd.utils import synthetic
from oracle import lightglue_oracle
from oracle.crosscheck_lightglue_hf import to_hf_state_dict

lightglue_hf = pytest.importorskip("transformers.models.lightglue.modeling_lightglue")

CASES = [
    # (weight kwargs, keypoints per image, expects early stop, expects pruning)
    ({}, 200, False, False),
    ({"conf_bias": 3.0, "conf_gain": 6.0}, 200, True, False),
    ({"conf_bias": 1.0, "conf_gain": 6.0, "match_bias": 2.0, "match_gain": 12.0}, 320, False, True),
    ({"conf_bias": 2.0, "conf_gain": 6.0, "match_bias": 1.0, "match_gain": 12.0}, 320, True, None),
]


@pytest.fixture(scope="module")
def hf_model():
    from transformers.models.lightglue.configuration_lightglue import LightGlueConfig

    torch.manual_seed(0)
    return lightglue_hf.LightGlueForKeypointMatching(LightGlueConfig()).eval()


@pytest.mark.parametrize("kwargs,n,early,pruned", CASES, ids=["plain", "early_stop", "pruning", "early_stop_2"])
def test_oracle_agrees_with_huggingface_port(hf_model, kwargs, n, early, pruned):
    import transformers

    sd = synthetic.synthetic_lightglue_state_dict(**kwargs)
    missing, unexpected = hf_model.load_state_dict(to_hf_state_dict(sd, 9), strict=False)
    assert not unexpected and all(k.startswith("keypoint_detector") for k in missing), (transformers.__version__, missing, unexpected)
    k0, _, d0, k1, _, d1, _ = synthetic.synthetic_pair_features(n, n, (480, 640), (480, 640), seed=5)  # HF pads to one length
    T = torch.from_numpy
    with torch.no_grad():
        ora = lightglue_oracle.lightglue_forward(sd, T(k0)[None], T(k1)[None], T(d0)[None], T(d1)[None], (480, 640), (480, 640),
                                                 pruning_threshold=-1, return_intermediates=True)
        kp = torch.stack([T(k0), T(k1)])[None]
        de = torch.stack([T(d0), T(d1)])[None]
        mask = torch.ones((1, 2, n), dtype=torch.int)
        matches, mscores, prune, _, _ = hf_model._match_image_pair(kp, de, 480, 640, mask=mask)
    assert (ora["stop"] < 9) == early
    if pruned is not None:
        assert ((ora["ind0"].shape[1] < n) or (ora["ind1"].shape[1] < n)) == pruned
    assert int((ora["matches0"][0] > -1).sum()) > 20
    for side in (0, 1):
        np.testing.assert_array_equal(matches[0, side].long().numpy(), ora[f"matches{side}"][0].long().numpy())
        np.testing.assert_allclose(mscores[0, side].numpy(), ora[f"matching_scores{side}"][0].numpy(), rtol=0, atol=2e-5)
        np.testing.assert_array_equal(prune[0, side].long().numpy(), ora[f"prune{side}"][0].long().numpy())
