"""Golden vectors for the LightGlue path from an INDEPENDENT implementation -- TEST INFRASTRUCTURE.

LightGlue's source is absent from the reference snapshot (un-vendored submodule, ``/root/reference/.gitmodules:1-3``), so no
fixture can come from the reference itself. The next best anchor available offline is the HuggingFace ``transformers`` port of
upstream cvg/LightGlue (``transformers/models/lightglue/modeling_lightglue.py``; transformers 5.15.0 in this image): a third
party's conversion of upstream, with different weight names, un-fused q/k/v and always-on pruning. This script feeds it the
seeded synthetic weights (``oracle/crosscheck_lightglue_hf.py:to_hf_state_dict``) and seeded synthetic features and stores its
inputs and outputs under ``tests/golden/lightglue_hf_*.npz``; ``tests/test_lightglue_hf_golden_gpu.py`` holds the HIP path to
them directly (matches identical, scores within 1e-4), ``tests/test_lightglue_crosscheck.py`` keeps the restatement equal to
the same model. Run:  python oracle/make_lightglue_hf_golden.py"""

import json
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from gtsfm_amd.utils import synthetic  # noqa: E402
from oracle.crosscheck_lightglue_hf import to_hf_state_dict  # noqa: E402

CASES = {
    "plain": ({}, 200),
    "early_stop": ({"conf_bias": 3.0, "conf_gain": 6.0}, 200),
    "pruning": ({"conf_bias": 1.0, "conf_gain": 6.0, "match_bias": 2.0, "match_gain": 12.0}, 320),
    "early_stop_pruning": ({"conf_bias": 2.0, "conf_gain": 6.0, "match_bias": 1.0, "match_gain": 12.0}, 320),
    "n2048_full_depth": ({}, 2048),  # the benchmark's keypoint count
}


def main() -> None:
    import transformers
    from transformers.models.lightglue.configuration_lightglue import LightGlueConfig
    from transformers.models.lightglue.modeling_lightglue import LightGlueForKeypointMatching

    torch.manual_seed(0)
    model = LightGlueForKeypointMatching(LightGlueConfig()).eval()
    out_dir = REPO / "tests" / "golden"
    for name, (kwargs, n) in CASES.items():
        sd = synthetic.synthetic_lightglue_state_dict(**kwargs)
        missing, unexpected = model.load_state_dict(to_hf_state_dict(sd, 9), strict=False)
        assert not unexpected and all(k.startswith("keypoint_detector") for k in missing)
        k0, _, d0, k1, _, d1, _ = synthetic.synthetic_pair_features(n, n, (480, 640), (480, 640), seed=5)
        t = torch.from_numpy
        with torch.no_grad():
            matches, mscores, prune, _, _ = model._match_image_pair(torch.stack([t(k0), t(k1)])[None], torch.stack([t(d0), t(d1)])[None], 480, 640,
                                                                    mask=torch.ones((1, 2, n), dtype=torch.int))
        m = matches[0].long().numpy()
        np.savez_compressed(
            out_dir / f"lightglue_hf_{name}.npz", weight_kwargs=json.dumps(kwargs), k0=k0, k1=k1, d0=d0, d1=d1, matches=m.astype(np.int32),
            mscores=mscores[0].numpy().astype(np.float32), prune=prune[0].long().numpy().astype(np.int32), image_hw=np.array([480, 640]),
            source=f"transformers {transformers.__version__} LightGlueForKeypointMatching._match_image_pair",
        )
        print(f"{name}: n = {n}, matches0 = {int((m[0] > -1).sum())}, layers pruned to {int(prune[0].max())} max counter")


if __name__ == "__main__":
    main()
