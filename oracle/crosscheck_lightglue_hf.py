"""ORACLE tooling: cross-check ``oracle/lightglue_oracle.py`` against an independent implementation.

LightGlue's source is absent from the reference, so the restatement cannot be pinned on the
reference. The HuggingFace ``transformers`` port (``transformers/models/lightglue/modeling_lightglue.py``, a converted
copy of upstream ``cvg/LightGlue``; NOT the reference, different weight names) happens to be installed in this image.
This script maps a synthetic upstream-named ``state_dict`` onto the HF module and checks that both produce the same
matches / matching scores for one pair (HF always prunes, i.e. upstream's CPU setting ``pruning threshold = -1``).
It is an orientation aid, run in the build container only; parity for LightGlue stays "unpinned".

Usage: python oracle/crosscheck_lightglue_hf.py
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

from gtsfm_amd.utils import synthetic  # noqa: E402
from oracle import lightglue_oracle  # noqa: E402


def to_hf_state_dict(sd, n_layers: int):
    out = {}
    out["positional_encoder.projector.weight"] = sd["posenc.Wr.weight"]
    for l in range(n_layers):
        u, h = f"transformers.{l}", f"transformer_layers.{l}"
        wqkv = sd[f"{u}.self_attn.Wqkv.weight"].unflatten(0, (4, 64, 3))  # channel = h*192 + d*3 + j
        bqkv = sd[f"{u}.self_attn.Wqkv.bias"].unflatten(0, (4, 64, 3))
        for j, name in enumerate(("q_proj", "k_proj", "v_proj")):
            out[f"{h}.self_attention.{name}.weight"] = wqkv[:, :, j].reshape(256, 256)
            out[f"{h}.self_attention.{name}.bias"] = bqkv[:, :, j].reshape(256)
        out[f"{h}.self_attention.o_proj.weight"] = sd[f"{u}.self_attn.out_proj.weight"]
        out[f"{h}.self_attention.o_proj.bias"] = sd[f"{u}.self_attn.out_proj.bias"]
        for name in ("q_proj", "k_proj"):
            out[f"{h}.cross_attention.{name}.weight"] = sd[f"{u}.cross_attn.to_qk.weight"]
            out[f"{h}.cross_attention.{name}.bias"] = sd[f"{u}.cross_attn.to_qk.bias"]
        out[f"{h}.cross_attention.v_proj.weight"] = sd[f"{u}.cross_attn.to_v.weight"]
        out[f"{h}.cross_attention.v_proj.bias"] = sd[f"{u}.cross_attn.to_v.bias"]
        out[f"{h}.cross_attention.o_proj.weight"] = sd[f"{u}.cross_attn.to_out.weight"]
        out[f"{h}.cross_attention.o_proj.bias"] = sd[f"{u}.cross_attn.to_out.bias"]
        for blk, mlp in (("self_attn", "self_mlp"), ("cross_attn", "cross_mlp")):
            out[f"{h}.{mlp}.fc1.weight"] = sd[f"{u}.{blk}.ffn.0.weight"]
            out[f"{h}.{mlp}.fc1.bias"] = sd[f"{u}.{blk}.ffn.0.bias"]
            out[f"{h}.{mlp}.layer_norm.weight"] = sd[f"{u}.{blk}.ffn.1.weight"]
            out[f"{h}.{mlp}.layer_norm.bias"] = sd[f"{u}.{blk}.ffn.1.bias"]
            out[f"{h}.{mlp}.fc2.weight"] = sd[f"{u}.{blk}.ffn.3.weight"]
            out[f"{h}.{mlp}.fc2.bias"] = sd[f"{u}.{blk}.ffn.3.bias"]
        out[f"match_assignment_layers.{l}.final_projection.weight"] = sd[f"log_assignment.{l}.final_proj.weight"]
        out[f"match_assignment_layers.{l}.final_projection.bias"] = sd[f"log_assignment.{l}.final_proj.bias"]
        out[f"match_assignment_layers.{l}.matchability.weight"] = sd[f"log_assignment.{l}.matchability.weight"]
        out[f"match_assignment_layers.{l}.matchability.bias"] = sd[f"log_assignment.{l}.matchability.bias"]
        if l < n_layers - 1:
            out[f"token_confidence.{l}.token.weight"] = sd[f"token_confidence.{l}.token.0.weight"]
            out[f"token_confidence.{l}.token.bias"] = sd[f"token_confidence.{l}.token.0.bias"]
    return out


def main() -> None:
    from transformers.models.lightglue.configuration_lightglue import LightGlueConfig
    from transformers.models.lightglue.modeling_lightglue import LightGlueForKeypointMatching

    torch.manual_seed(0)
    hf = LightGlueForKeypointMatching(LightGlueConfig()).eval()
    for kwargs in ({}, {"conf_bias": 3.0, "conf_gain": 6.0}):
        sd = synthetic.synthetic_lightglue_state_dict(**kwargs)
        missing, unexpected = hf.load_state_dict(to_hf_state_dict(sd, 9), strict=False)
        assert not unexpected and all(k.startswith("keypoint_detector") for k in missing), (missing, unexpected)
        n = 200  # HF pads both images to one length; use equal counts
        k0, s0, d0, k1, s1, d1, gt = synthetic.synthetic_pair_features(n, n, (480, 640), (480, 640), seed=5)
        T = torch.from_numpy
        with torch.no_grad():
            ora = lightglue_oracle.lightglue_forward(
                sd, T(k0)[None], T(k1)[None], T(d0)[None], T(d1)[None], (480, 640), (480, 640), pruning_threshold=-1
            )
            kp = torch.stack([T(k0), T(k1)])[None]
            de = torch.stack([T(d0), T(d1)])[None]
            mask = torch.ones((1, 2, n), dtype=torch.int)
            matches, mscores, prune, _, _ = hf._match_image_pair(kp, de, 480, 640, mask=mask)
        m0_hf, ms0_hf = matches[0, 0], mscores[0, 0]
        m0, ms0 = ora["matches0"][0], ora["matching_scores0"][0]
        same = bool(torch.equal(m0_hf.long(), m0.long()))
        err = float((ms0_hf - ms0).abs().max())
        print(
            f"conf {kwargs}: layers run (oracle) = {ora['stop']}, matches = {int((m0 > -1).sum())}, "
            f"kept0 = {ora['ind0'].shape[1] if 'ind0' in ora else 'n/a'}, "
            f"matches0 identical = {same}, max |d score| = {err:.2e}, "
            f"prune0 identical = {bool(torch.equal(prune[0, 0].long(), ora['prune0'][0].long()))}"
        )
        assert same and err < 1e-5
    print("OK")


if __name__ == "__main__":
    main()
