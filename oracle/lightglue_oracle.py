"""ORACLE (test infrastructure, not product code): CPU restatement of LightGlue(features="superpoint").

PARITY UNPINNED. The arithmetic of this matcher is NOT in the reference tree: ``thirdparty/LightGlue`` is an empty,
un-vendored git submodule (``.gitmodules:1-3`` -> ``cvg/LightGlue``, no tag/branch, gitlink SHA unrecoverable;
SURVEY.md F6) and no reference test exercises ``LightGlueMatcher`` (``grep -rni lightglue tests/`` is empty). This
file restates the published upstream algorithm (``cvg/LightGlue`` ``lightglue/lightglue.py``, v0.1 "superpoint"
configuration: 9 layers, 4 heads, descriptor_dim 256, depth_confidence 0.95, width_confidence 0.99,
filter_threshold 0.1) and is anchored on the reference's call sites only:

* ``gtsfm/frontend/matcher/lightglue_matcher.py:37,41``  -- ``LightGlue(features=...).eval()`` with default config
* ``gtsfm/frontend/matcher/lightglue_matcher.py:88-104`` -- feature dicts {keypoints, keypoint_scores, descriptors
  (1,N,256), image (shape only)} -> ``model({"image0", "image1"})``
* ``gtsfm/frontend/matcher/lightglue_matcher.py:107-110`` -- ``rbd(...)["matches"]`` -> (K,2) int64

Any LightGlue parity claim in this repository therefore reads "HIP path == this restatement".

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Pieces (upstream names):
* ``normalize_keypoints``                  -- (k - size/2) / (max(size)/2)
* ``LearnableFourierPositionalEncoding``   -- Wr: 2 -> 32, cos/sin, repeat_interleave(2) -> rotary over head_dim 64
* ``SelfBlock``                            -- Wqkv (channel = h*192 + d*3 + {q,k,v}), rotary on q,k, softmax
  attention, out_proj, ffn(cat[x,msg]) = Linear(512,512) -> LayerNorm -> GELU -> Linear(512,256), residual
* ``CrossBlock``                           -- shared to_qk, to_v, one similarity, softmax along both axes, to_out,
  ffn as above, residual
* ``TokenConfidence``                      -- sigmoid(Linear(256,1)) per layer (all but the last)
* ``MatchAssignment``                      -- final_proj / 256^(1/4), sim, matchability, sigmoid_log_double_softmax
* ``filter_matches``                       -- mutual argmax + exp > threshold
* early stop (``check_if_stop``) and point pruning (``get_pruning_mask``); upstream enables pruning when
  ``N > pruning_keypoint_thresholds[device]`` = {cpu: -1, cuda: 1024, flash: 1536}; the threshold is a parameter
  here (``pruning_threshold``; ``None`` disables pruning).
"""

from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]

NUM_HEADS = 4
DEPTH_CONFIDENCE = 0.95
WIDTH_CONFIDENCE = 0.99
FILTER_THRESHOLD = 0.1


def num_layers(sd: StateDict) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("transformers."))


def confidence_threshold(layer_index: int, n_layers: int) -> float:
    threshold = 0.8 + 0.1 * np.exp(-4.0 * layer_index / n_layers)
    return float(np.clip(threshold, 0, 1))


def normalize_keypoints(kpts: torch.Tensor, height: int, width: int) -> torch.Tensor:
    size = torch.tensor([width, height], dtype=kpts.dtype)
    shift = size / 2
    scale = size.max() / 2
    return (kpts - shift[None, None, :]) / scale


def positional_encoding(sd: StateDict, kpts: torch.Tensor) -> torch.Tensor:
    """Returns [2, B, 1, N, 64] (cos, sin), each frequency repeated twice along the last axis."""
    projected = F.linear(kpts, sd["posenc.Wr.weight"])
    cosines, sines = torch.cos(projected), torch.sin(projected)
    emb = torch.stack([cosines, sines], 0).unsqueeze(-3)
    return emb.repeat_interleave(2, dim=-1)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def apply_cached_rotary_emb(freqs: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    return (t * freqs[0]) + (rotate_half(t) * freqs[1])


def _sdpa(q, k, v):
    """softmax(q k^T / sqrt(d)) v, written out (upstream uses F.scaled_dot_product_attention when available)."""
    s = q.shape[-1] ** -0.5
    sim = torch.einsum("...id,...jd->...ij", q, k) * s
    attn = F.softmax(sim, -1)
    return torch.einsum("...ij,...jd->...id", attn, v)


def _ffn(sd: StateDict, prefix: str, x: torch.Tensor) -> torch.Tensor:
    x = F.linear(x, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"])
    x = F.layer_norm(x, (x.shape[-1],), sd[f"{prefix}.1.weight"], sd[f"{prefix}.1.bias"], eps=1e-5)
    x = F.gelu(x)
    return F.linear(x, sd[f"{prefix}.3.weight"], sd[f"{prefix}.3.bias"])


def self_block(sd: StateDict, prefix: str, x: torch.Tensor, encoding: torch.Tensor) -> torch.Tensor:
    qkv = F.linear(x, sd[f"{prefix}.Wqkv.weight"], sd[f"{prefix}.Wqkv.bias"])
    qkv = qkv.unflatten(-1, (NUM_HEADS, -1, 3)).transpose(1, 2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q = apply_cached_rotary_emb(encoding, q)
    k = apply_cached_rotary_emb(encoding, k)
    context = _sdpa(q, k, v)
    message = F.linear(
        context.transpose(1, 2).flatten(start_dim=-2), sd[f"{prefix}.out_proj.weight"], sd[f"{prefix}.out_proj.bias"]
    )
    return x + _ffn(sd, f"{prefix}.ffn", torch.cat([x, message], -1))


def cross_block(sd: StateDict, prefix: str, x0: torch.Tensor, x1: torch.Tensor):
    scale = (x0.shape[-1] // NUM_HEADS) ** -0.5
    qk0 = F.linear(x0, sd[f"{prefix}.to_qk.weight"], sd[f"{prefix}.to_qk.bias"])
    qk1 = F.linear(x1, sd[f"{prefix}.to_qk.weight"], sd[f"{prefix}.to_qk.bias"])
    v0 = F.linear(x0, sd[f"{prefix}.to_v.weight"], sd[f"{prefix}.to_v.bias"])
    v1 = F.linear(x1, sd[f"{prefix}.to_v.weight"], sd[f"{prefix}.to_v.bias"])
    qk0, qk1, v0, v1 = [t.unflatten(-1, (NUM_HEADS, -1)).transpose(1, 2) for t in (qk0, qk1, v0, v1)]
    qk0, qk1 = qk0 * scale**0.5, qk1 * scale**0.5
    sim = torch.einsum("bhid,bhjd->bhij", qk0, qk1)
    attn01 = F.softmax(sim, dim=-1)
    attn10 = F.softmax(sim.transpose(-2, -1).contiguous(), dim=-1)
    m0 = torch.einsum("bhij,bhjd->bhid", attn01, v1)
    m1 = torch.einsum("bhji,bhjd->bhid", attn10.transpose(-2, -1), v0)
    m0, m1 = [t.transpose(1, 2).flatten(start_dim=-2) for t in (m0, m1)]
    m0 = F.linear(m0, sd[f"{prefix}.to_out.weight"], sd[f"{prefix}.to_out.bias"])
    m1 = F.linear(m1, sd[f"{prefix}.to_out.weight"], sd[f"{prefix}.to_out.bias"])
    x0 = x0 + _ffn(sd, f"{prefix}.ffn", torch.cat([x0, m0], -1))
    x1 = x1 + _ffn(sd, f"{prefix}.ffn", torch.cat([x1, m1], -1))
    return x0, x1


def token_confidence(sd: StateDict, layer: int, desc: torch.Tensor) -> torch.Tensor:
    p = f"token_confidence.{layer}.token.0"
    return torch.sigmoid(F.linear(desc, sd[f"{p}.weight"], sd[f"{p}.bias"])).squeeze(-1)


def matchability(sd: StateDict, layer: int, desc: torch.Tensor) -> torch.Tensor:
    p = f"log_assignment.{layer}.matchability"
    return torch.sigmoid(F.linear(desc, sd[f"{p}.weight"], sd[f"{p}.bias"])).squeeze(-1)


def sigmoid_log_double_softmax(sim: torch.Tensor, z0: torch.Tensor, z1: torch.Tensor) -> torch.Tensor:
    b, m, n = sim.shape
    certainties = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    scores0 = F.log_softmax(sim, 2)
    scores1 = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    scores = sim.new_full((b, m + 1, n + 1), 0)
    scores[:, :m, :n] = scores0 + scores1 + certainties
    scores[:, :-1, -1] = F.logsigmoid(-z0.squeeze(-1))
    scores[:, -1, :-1] = F.logsigmoid(-z1.squeeze(-1))
    return scores


def match_assignment(sd: StateDict, layer: int, desc0: torch.Tensor, desc1: torch.Tensor):
    p = f"log_assignment.{layer}"
    mdesc0 = F.linear(desc0, sd[f"{p}.final_proj.weight"], sd[f"{p}.final_proj.bias"])
    mdesc1 = F.linear(desc1, sd[f"{p}.final_proj.weight"], sd[f"{p}.final_proj.bias"])
    d = mdesc0.shape[-1]
    mdesc0, mdesc1 = mdesc0 / d**0.25, mdesc1 / d**0.25
    sim = torch.einsum("bmd,bnd->bmn", mdesc0, mdesc1)
    z0 = F.linear(desc0, sd[f"{p}.matchability.weight"], sd[f"{p}.matchability.bias"])
    z1 = F.linear(desc1, sd[f"{p}.matchability.weight"], sd[f"{p}.matchability.bias"])
    return sigmoid_log_double_softmax(sim, z0, z1), sim


def filter_matches(scores: torch.Tensor, th: float):
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    m0, m1 = max0.indices, max1.indices
    indices0 = torch.arange(m0.shape[1])[None]
    indices1 = torch.arange(m1.shape[1])[None]
    mutual0 = indices0 == m1.gather(1, m0)
    mutual1 = indices1 == m0.gather(1, m1)
    max0_exp = max0.values.exp()
    zero = max0_exp.new_tensor(0)
    mscores0 = torch.where(mutual0, max0_exp, zero)
    mscores1 = torch.where(mutual1, mscores0.gather(1, m1), zero)
    valid0 = mutual0 & (mscores0 > th)
    valid1 = mutual1 & valid0.gather(1, m1)
    m0 = torch.where(valid0, m0, -1)
    m1 = torch.where(valid1, m1, -1)
    return m0, m1, mscores0, mscores1


def lightglue_forward(
    sd: StateDict,
    kpts0: torch.Tensor,
    kpts1: torch.Tensor,
    desc0: torch.Tensor,
    desc1: torch.Tensor,
    shape0: Tuple[int, int],
    shape1: Tuple[int, int],
    depth_confidence: float = DEPTH_CONFIDENCE,
    width_confidence: float = WIDTH_CONFIDENCE,
    filter_threshold: float = FILTER_THRESHOLD,
    pruning_threshold: Optional[int] = 1536,
    return_intermediates: bool = False,
) -> Dict[str, torch.Tensor]:
    """Upstream ``LightGlue._forward`` for batch size 1. kpts [1,N,2] (x,y); desc [1,N,256]; shape = (H, W).

    Returns ``matches`` (K,2) int64, ``scores`` (K,), ``matches0/1``, ``matching_scores0/1``, ``stop`` (number of
    layers run), ``prune0/1`` (per-keypoint count of layers survived).
    """
    dt = desc0.dtype
    sd = {k: v.to(dt) if v.is_floating_point() else v for k, v in sd.items()}
    n_layers = num_layers(sd)
    b, m, _ = kpts0.shape
    _, n, _ = kpts1.shape
    assert b == 1
    kpts0 = normalize_keypoints(kpts0, shape0[0], shape0[1])
    kpts1 = normalize_keypoints(kpts1, shape1[0], shape1[1])
    desc0 = desc0.contiguous()
    desc1 = desc1.contiguous()
    encoding0 = positional_encoding(sd, kpts0)
    encoding1 = positional_encoding(sd, kpts1)

    do_early_stop = depth_confidence > 0
    do_point_pruning = width_confidence > 0 and pruning_threshold is not None
    ind0 = torch.arange(0, m)[None]
    ind1 = torch.arange(0, n)[None]
    prune0 = torch.ones_like(ind0)
    prune1 = torch.ones_like(ind1)
    token0 = token1 = None
    inter = {}
    i = 0
    for i in range(n_layers):
        if desc0.shape[1] == 0 or desc1.shape[1] == 0:
            break
        desc0 = self_block(sd, f"transformers.{i}.self_attn", desc0, encoding0)
        desc1 = self_block(sd, f"transformers.{i}.self_attn", desc1, encoding1)
        desc0, desc1 = cross_block(sd, f"transformers.{i}.cross_attn", desc0, desc1)
        if return_intermediates:
            inter[f"desc0_l{i}"], inter[f"desc1_l{i}"] = desc0.clone(), desc1.clone()
        if i == n_layers - 1:
            continue
        if do_early_stop:
            token0, token1 = token_confidence(sd, i, desc0), token_confidence(sd, i, desc1)
            confidences = torch.cat([token0, token1], -1)
            threshold = confidence_threshold(i, n_layers)
            ratio_confident = 1.0 - (confidences < threshold).float().sum() / (m + n)
            if ratio_confident > depth_confidence:
                break
        if do_point_pruning and desc0.shape[-2] > pruning_threshold:
            scores0 = matchability(sd, i, desc0)
            keep = scores0 > (1 - width_confidence)
            if token0 is not None:
                keep |= token0 <= confidence_threshold(i, n_layers)
            keep0 = torch.where(keep)[1]
            ind0 = ind0.index_select(1, keep0)
            desc0 = desc0.index_select(1, keep0)
            encoding0 = encoding0.index_select(-2, keep0)
            prune0[:, ind0] += 1
        if do_point_pruning and desc1.shape[-2] > pruning_threshold:
            scores1 = matchability(sd, i, desc1)
            keep = scores1 > (1 - width_confidence)
            if token1 is not None:
                keep |= token1 <= confidence_threshold(i, n_layers)
            keep1 = torch.where(keep)[1]
            ind1 = ind1.index_select(1, keep1)
            desc1 = desc1.index_select(1, keep1)
            encoding1 = encoding1.index_select(-2, keep1)
            prune1[:, ind1] += 1

    if desc0.shape[1] == 0 or desc1.shape[1] == 0:
        m0 = desc0.new_full((b, m), -1, dtype=torch.long)
        m1 = desc0.new_full((b, n), -1, dtype=torch.long)
        out = {
            "matches0": m0, "matches1": m1,
            "matching_scores0": desc0.new_zeros((b, m)), "matching_scores1": desc0.new_zeros((b, n)),
            "matches": desc0.new_empty((0, 2), dtype=torch.long), "scores": desc0.new_empty((0,)),
            "stop": i + 1, "prune0": prune0, "prune1": prune1,
        }
        return out

    scores, sim = match_assignment(sd, i, desc0, desc1)
    m0, m1, mscores0, mscores1 = filter_matches(scores, filter_threshold)
    valid = m0[0] > -1
    m_indices_0 = torch.where(valid)[0]
    m_indices_1 = m0[0][valid]
    if do_point_pruning:
        m_indices_0 = ind0[0, m_indices_0]
        m_indices_1 = ind1[0, m_indices_1]
    matches = torch.stack([m_indices_0, m_indices_1], -1)
    mscores = mscores0[0][valid]

    if do_point_pruning:
        m0_ = torch.full((b, m), -1, dtype=m0.dtype)
        m1_ = torch.full((b, n), -1, dtype=m1.dtype)
        m0_[:, ind0] = torch.where(m0 == -1, -1, ind1.gather(1, m0.clamp(min=0)))
        m1_[:, ind1] = torch.where(m1 == -1, -1, ind0.gather(1, m1.clamp(min=0)))
        mscores0_ = torch.zeros((b, m), dtype=dt)
        mscores1_ = torch.zeros((b, n), dtype=dt)
        mscores0_[:, ind0] = mscores0
        mscores1_[:, ind1] = mscores1
        m0, m1, mscores0, mscores1 = m0_, m1_, mscores0_, mscores1_
    else:
        prune0 = torch.ones_like(mscores0) * n_layers
        prune1 = torch.ones_like(mscores1) * n_layers

    out = {
        "matches0": m0, "matches1": m1, "matching_scores0": mscores0, "matching_scores1": mscores1,
        "matches": matches, "scores": mscores, "stop": i + 1, "prune0": prune0, "prune1": prune1,
    }
    if return_intermediates:
        out.update(inter)
        out.update(log_assignment=scores, sim=sim, ind0=ind0, ind1=ind1)
    return out


def match(
    sd: StateDict,
    coords0: np.ndarray,
    coords1: np.ndarray,
    desc0: np.ndarray,
    desc1: np.ndarray,
    im_shape0: Tuple[int, ...],
    im_shape1: Tuple[int, ...],
    dtype=torch.float32,
    **kwargs,
) -> np.ndarray:
    """gtsfm/frontend/matcher/lightglue_matcher.py:75-112: numpy -> feature dicts -> model -> (K,2) int64."""
    with torch.no_grad():
        out = lightglue_forward(
            sd,
            torch.from_numpy(coords0).unsqueeze(0).float().to(dtype),
            torch.from_numpy(coords1).unsqueeze(0).float().to(dtype),
            torch.from_numpy(desc0).unsqueeze(0).float().to(dtype),
            torch.from_numpy(desc1).unsqueeze(0).float().to(dtype),
            (im_shape0[0], im_shape0[1]),
            (im_shape1[0], im_shape1[1]),
            **kwargs,
        )
    return out["matches"].numpy()


_ = math  # keep import (documented constants above use it in formulas)
