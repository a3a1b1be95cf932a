"""Deterministic synthetic weights and inputs for the deep front-end.

No pretrained weights exist offline (``scripts/download_model_weights.sh:6-21`` fetches them with
wget), so parity work and benchmarks use seeded synthetic ``state_dict``s that follow the tensor names and shapes of
the reference checkpoints:

* SuperPoint  -- ``thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:117-134``
* SuperGlue   -- ``thirdparty/SuperGluePretrainedNetwork/models/superglue.py:49-60,92-138,205-220``
* LightGlue   -- upstream ``cvg/LightGlue`` ``lightglue/lightglue.py`` module names (source absent from the
  reference tree)

The raw Kaiming-uniform initialisation gives degenerate outputs (one keypoint per 8x8 cell, zero matches), so a few
tensors are rescaled to obtain realistic keypoint counts and non-trivial match lists. The same blob is fed to the
oracle and to the HIP path.
"""

from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

StateDict = Dict[str, torch.Tensor]


def _conv_init(gen: torch.Generator, cout: int, cin: int, k: int, nd: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Kaiming-uniform(a=sqrt(5)) weight and matching bias, as torch.nn.ConvNd constructs them."""
    fan_in = cin * k**nd
    bound = 1.0 / math.sqrt(fan_in)
    shape = (cout, cin) + (k,) * nd
    w = (torch.rand(shape, generator=gen) * 2 - 1) * bound
    b = (torch.rand((cout,), generator=gen) * 2 - 1) * bound
    return w, b


def _linear_init(gen: torch.Generator, cout: int, cin: int) -> Tuple[torch.Tensor, torch.Tensor]:
    bound = 1.0 / math.sqrt(cin)
    w = (torch.rand((cout, cin), generator=gen) * 2 - 1) * bound
    b = (torch.rand((cout,), generator=gen) * 2 - 1) * bound
    return w, b


# ----------------------------------------------------------------------------------------------------------------
# SuperPoint
# ----------------------------------------------------------------------------------------------------------------

SUPERPOINT_LAYERS = (
    # name, cin, cout, kernel
    ("conv1a", 1, 64, 3),
    ("conv1b", 64, 64, 3),
    ("conv2a", 64, 64, 3),
    ("conv2b", 64, 64, 3),
    ("conv3a", 64, 128, 3),
    ("conv3b", 128, 128, 3),
    ("conv4a", 128, 128, 3),
    ("conv4b", 128, 128, 3),
    ("convPa", 128, 256, 3),
    ("convPb", 256, 65, 1),
    ("convDa", 128, 256, 3),
    ("convDb", 256, 256, 1),
)


_DESCRIPTOR_MEAN_CACHE: Dict[int, torch.Tensor] = {}


def _descriptor_head_mean(sd: StateDict) -> torch.Tensor:
    """Mean pre-normalisation dense descriptor (convDb output) of a small seeded image, in float64, rounded to a
    2^-12 grid so that the result does not depend on the host's summation order."""
    import torch.nn.functional as F

    w = {k: v.double() for k, v in sd.items()}
    x = torch.from_numpy(synthetic_gray_image(96, 96, 77).astype(np.float64) / 255.0)[None, None]
    with torch.no_grad():
        for name in ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convDa"):
            x = F.relu(F.conv2d(x, w[f"{name}.weight"], w[f"{name}.bias"], padding=1))
            if name in ("conv1b", "conv2b", "conv3b"):
                x = F.max_pool2d(x, 2, 2)
        dense = F.conv2d(x, w["convDb.weight"], w["convDb.bias"])[0, :, 2:-2, 2:-2]
    return torch.round(dense.reshape(256, -1).mean(1) * 4096.0) / 4096.0


def synthetic_superpoint_state_dict(
    seed: int = 1234, logit_gain: float = 12.0, dustbin_bias: float = 11.5, center_descriptors: bool = True
) -> StateDict:
    """Seeded SuperPoint weights (24 tensors, 1 300 865 parameters).

    ``sqrt(2)``-scaled encoder weights keep ReLU activations O(1) through the 8-layer stack; ``logit_gain`` widens the
    65-way detector logits and ``dustbin_bias`` raises the "no keypoint" channel so that only a few percent of the
    pixels clear the 0.005 threshold, as with trained weights. A random ReLU stack maps every image patch to nearly the
    same descriptor (mutual cosine similarity 0.97: no matcher can tell keypoints apart, the round-1 bench found zero
    matches); ``center_descriptors`` subtracts the mean dense descriptor of a calibration image from ``convDb.bias``,
    which leaves the position-dependent part (mutual similarity 0.3, nearest neighbour correct for 98 % of the shared
    keypoints of two overlapping views).
    """
    gen = torch.Generator().manual_seed(seed)
    sd: StateDict = {}
    for name, cin, cout, k in SUPERPOINT_LAYERS:
        w, b = _conv_init(gen, cout, cin, k, 2)
        if name not in ("convPb", "convDb"):
            w = w * math.sqrt(6.0)  # variance-preserving for ReLU: Var(w) = 2 / fan_in
        sd[f"{name}.weight"] = w
        sd[f"{name}.bias"] = b
    sd["convPb.weight"] = sd["convPb.weight"] * logit_gain
    bias = sd["convPb.bias"].clone()
    bias[64] += dustbin_bias
    sd["convPb.bias"] = bias
    if center_descriptors:
        if seed not in _DESCRIPTOR_MEAN_CACHE:
            _DESCRIPTOR_MEAN_CACHE[seed] = _descriptor_head_mean(sd)
        sd["convDb.bias"] = (sd["convDb.bias"].double() - _DESCRIPTOR_MEAN_CACHE[seed]).float()
    return {k: v.contiguous() for k, v in sd.items()}


def synthetic_gray_image(height: int, width: int, seed: int = 0, blur: int = 3) -> np.ndarray:
    """Seeded low-pass-filtered uint8 noise: box blur of uniform noise,
    contrast-stretched back to the full 8-bit range."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(height + 2 * blur, width + 2 * blur)).astype(np.float64)
    if blur > 0:
        k = 2 * blur + 1
        c = np.cumsum(np.pad(img, ((1, 0), (0, 0))), axis=0)
        img = (c[k:, :] - c[:-k, :]) / k
        c = np.cumsum(np.pad(img, ((0, 0), (1, 0))), axis=1)
        img = (c[:, k:] - c[:, :-k]) / k
    else:
        img = img[:height, :width]
    img = img[:height, :width]
    lo, hi = img.min(), img.max()
    img = (img - lo) / max(hi - lo, 1e-9) * 255.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def view_offsets(num_views: int):
    """(dy, dx) canvas offsets of the overlapping views below: multiples of the 8-px SuperPoint cell, so that the
    encoder sees the shared content on the same cell grid in every view."""
    return [(8 * i, 8 * ((7 * i) % num_views)) for i in range(num_views)]


def synthetic_overlapping_views(num_views: int, height: int, width: int, seed: int = 1000) -> np.ndarray:
    """``num_views`` crops [n,H,W] uint8 of ONE seeded canvas (bench.py's images): every pair of views shares most of
    its content, shifted by ``view_offsets``, so exhaustive pairs have true correspondences while every view is still
    detected independently."""
    canvas = synthetic_gray_image(height + 8 * num_views, width + 8 * num_views, seed)
    return np.stack([canvas[dy : dy + height, dx : dx + width] for dy, dx in view_offsets(num_views)])


def synthetic_mixed_scene(num_views: int, height: int, width: int, seed: int = 2000, canvases: int = 3) -> np.ndarray:
    """``num_views`` views [n,H,W] uint8 that do NOT all look alike (bench.py's adaptive-depth leg): view v is cut from canvas
    ``v % canvases`` -- canvases differ in seed and in low-pass radius (2, 3, 5, ...), i.e. in texture scale -- at an offset that grows with
    ``v // canvases`` in steps of 1/8 of the image side (multiples of the 8-px cell). Pairs of one canvas overlap by 100 % ... ~30 %; pairs
    across canvases share nothing, like the unrelated pairs of a real exhaustive visibility graph. A matcher's early exit and point pruning
    then have something to tell apart."""
    per = -(-num_views // canvases)
    step = max(8, (min(height, width) // 8) // 8 * 8)
    radii = [2, 3, 5, 4, 6]
    pads = step * per
    sheets = [synthetic_gray_image(height + pads, width + pads, seed + 17 * c, blur=radii[c % len(radii)]) for c in range(canvases)]
    views = []
    for v in range(num_views):
        c, k = v % canvases, v // canvases
        dy, dx = step * k, step * ((3 * k) % per)
        views.append(sheets[c][dy : dy + height, dx : dx + width])
    return np.stack(views)


def topk_detection_order(scores: np.ndarray, k: int) -> np.ndarray:
    """Indices of the ``k`` largest responses, ties broken by detection order, returned in detection order -- the
    selection of ``kp_select_topk_kernel`` (``Keypoints.get_top_k``'s ``np.argpartition`` leaves ties and order
    implementation-defined, gtsfm/common/keypoints.py:89-110)."""
    if k >= len(scores):
        return np.arange(len(scores))
    return np.sort(np.lexsort((np.arange(len(scores)), -scores.astype(np.float64)))[:k])


# ----------------------------------------------------------------------------------------------------------------
# SuperGlue
# ----------------------------------------------------------------------------------------------------------------


def _bn_init(gen: torch.Generator, c: int, prefix: str, sd: StateDict) -> None:
    """Non-trivial eval-mode BatchNorm1d statistics (SURVEY.md section 8c caveat 2)."""
    sd[f"{prefix}.weight"] = 1.0 + 0.2 * (torch.rand((c,), generator=gen) - 0.5)
    sd[f"{prefix}.bias"] = 0.2 * (torch.rand((c,), generator=gen) - 0.5)
    sd[f"{prefix}.running_mean"] = 0.2 * (torch.rand((c,), generator=gen) - 0.5)
    sd[f"{prefix}.running_var"] = 0.5 + torch.rand((c,), generator=gen)
    sd[f"{prefix}.num_batches_tracked"] = torch.tensor(1000, dtype=torch.long)


def synthetic_superglue_state_dict(
    seed: int = 4321, num_layers: int = 18, delta_gain: float = 0.25, final_gain: float = 24.0
) -> StateDict:
    """Seeded SuperGlue weights (12 023 297 parameters for 18 layers).

    ``delta_gain`` shrinks the last MLP layer of every propagation block so the residual stream stays dominated by
    the input descriptors; ``final_gain`` scales ``final_proj`` so matched descriptors reach a score that survives the
    Sinkhorn dustbin (with raw init every keypoint goes to the dustbin).
    """
    gen = torch.Generator().manual_seed(seed)
    sd: StateDict = {}
    chans = [3, 32, 64, 128, 256, 256]
    for i in range(1, len(chans)):
        w, b = _conv_init(gen, chans[i], chans[i - 1], 1, 1)
        idx = 3 * (i - 1)
        sd[f"kenc.encoder.{idx}.weight"] = w * (math.sqrt(3.0) if i < len(chans) - 1 else 0.5)
        sd[f"kenc.encoder.{idx}.bias"] = b if i < len(chans) - 1 else torch.zeros_like(b)
        if i < len(chans) - 1:
            _bn_init(gen, chans[i], f"kenc.encoder.{idx + 1}", sd)
    for l in range(num_layers):
        p = f"gnn.layers.{l}"
        w, b = _conv_init(gen, 256, 256, 1, 1)
        sd[f"{p}.attn.merge.weight"], sd[f"{p}.attn.merge.bias"] = w * 1.5, b
        for j in range(3):
            w, b = _conv_init(gen, 256, 256, 1, 1)
            # q/k gain makes the softmax non-uniform so that attention is a non-trivial function of the inputs
            sd[f"{p}.attn.proj.{j}.weight"] = w * (6.0 if j < 2 else 1.5)
            sd[f"{p}.attn.proj.{j}.bias"] = b
        w, b = _conv_init(gen, 512, 512, 1, 1)
        sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"] = w * math.sqrt(3.0), b
        _bn_init(gen, 512, f"{p}.mlp.1", sd)
        w, b = _conv_init(gen, 256, 512, 1, 1)
        sd[f"{p}.mlp.3.weight"], sd[f"{p}.mlp.3.bias"] = w * delta_gain, torch.zeros_like(b)
    w, b = _conv_init(gen, 256, 256, 1, 1)
    sd["final_proj.weight"], sd["final_proj.bias"] = w * final_gain, b
    sd["bin_score"] = torch.tensor(1.0)
    return {k: v.contiguous() for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------------------------
# LightGlue
# ----------------------------------------------------------------------------------------------------------------


def synthetic_lightglue_state_dict(
    seed: int = 9876,
    num_layers: int = 9,
    delta_gain: float = 0.25,
    final_gain: float = 16.0,
    conf_bias: float = 0.0,
    conf_gain: float = 1.0,
    match_bias: float = 2.0,
    match_gain: float = 4.0,
    conf_ramp: float = 0.0,
    conf_shared_direction: bool = False,
) -> StateDict:
    """Seeded LightGlue(features="superpoint") weights, upstream module names.

    ``conf_bias`` / ``conf_gain`` shape the token-confidence heads and ``match_bias`` / ``match_gain`` the matchability
    heads so that early stopping and point pruning can be exercised (or suppressed) by tests; ``conf_ramp`` adds ``ramp * layer`` to the
    token-confidence bias (confidence that grows with depth, so that pairs leave at DIFFERENT layers); ``conf_shared_direction`` gives every layer's
    token-confidence head the direction of layer 0's (the residual stream is dominated by the input descriptors, so a keypoint's confidence then
    depends on its image's descriptor statistics plus the ramp: images of different texture become confident at different depths).
    """
    gen = torch.Generator().manual_seed(seed)
    sd: StateDict = {}
    sd["posenc.Wr.weight"] = torch.randn((32, 2), generator=gen) * 2.0
    for l in range(num_layers):
        p = f"transformers.{l}"
        w, b = _linear_init(gen, 768, 256)
        sd[f"{p}.self_attn.Wqkv.weight"], sd[f"{p}.self_attn.Wqkv.bias"] = w * 4.0, b
        w, b = _linear_init(gen, 256, 256)
        sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"] = w * 1.5, b
        w, b = _linear_init(gen, 256, 256)
        sd[f"{p}.cross_attn.to_qk.weight"], sd[f"{p}.cross_attn.to_qk.bias"] = w * 5.0, b
        w, b = _linear_init(gen, 256, 256)
        sd[f"{p}.cross_attn.to_v.weight"], sd[f"{p}.cross_attn.to_v.bias"] = w * 1.5, b
        w, b = _linear_init(gen, 256, 256)
        sd[f"{p}.cross_attn.to_out.weight"], sd[f"{p}.cross_attn.to_out.bias"] = w * 1.5, b
        for blk in ("self_attn", "cross_attn"):
            w, b = _linear_init(gen, 512, 512)
            sd[f"{p}.{blk}.ffn.0.weight"], sd[f"{p}.{blk}.ffn.0.bias"] = w * math.sqrt(3.0), b
            sd[f"{p}.{blk}.ffn.1.weight"] = 1.0 + 0.2 * (torch.rand((512,), generator=gen) - 0.5)
            sd[f"{p}.{blk}.ffn.1.bias"] = 0.2 * (torch.rand((512,), generator=gen) - 0.5)
            w, b = _linear_init(gen, 256, 512)
            sd[f"{p}.{blk}.ffn.3.weight"], sd[f"{p}.{blk}.ffn.3.bias"] = w * delta_gain, b * delta_gain
        w, b = _linear_init(gen, 1, 256)
        sd[f"log_assignment.{l}.matchability.weight"] = w * match_gain
        sd[f"log_assignment.{l}.matchability.bias"] = b + match_bias
        w, b = _linear_init(gen, 256, 256)
        sd[f"log_assignment.{l}.final_proj.weight"] = w * final_gain
        sd[f"log_assignment.{l}.final_proj.bias"] = b
        if l < num_layers - 1:
            w, b = _linear_init(gen, 1, 256)
            if conf_shared_direction and l > 0:  # (the generator is advanced all the same: every other tensor keeps its values)
                w = sd["token_confidence.0.token.0.weight"] / conf_gain
            sd[f"token_confidence.{l}.token.0.weight"] = w * conf_gain
            sd[f"token_confidence.{l}.token.0.bias"] = b + conf_bias + conf_ramp * l
    return {k: v.contiguous() for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------------------------
# Matcher inputs
# ----------------------------------------------------------------------------------------------------------------


def synthetic_pair_features(
    n0: int,
    n1: int,
    shape0: Tuple[int, int] = (1024, 1024),
    shape1: Tuple[int, int] = (1024, 1024),
    overlap: float = 0.6,
    noise: float = 0.15,
    seed: int = 0,
):
    """Two keypoint/descriptor sets where ``overlap * min(n0, n1)`` keypoints of image 1 are shifted, noised copies
    of keypoints of image 0 (SURVEY.md section 7 "hard parts": permuted + noised copies give non-trivial matches).

    Returns ``(kpts0 [n0,2] f32 (x,y), scores0 [n0] f32, desc0 [n0,256] f32, kpts1, scores1, desc1, gt)`` where
    ``gt[i]`` is the index in image 1 of keypoint ``i`` of image 0 (or -1).
    """
    rng = np.random.default_rng(seed)
    h0, w0 = shape0
    h1, w1 = shape1

    def unit(x):
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)

    k0 = np.stack([rng.uniform(4, w0 - 5, n0), rng.uniform(4, h0 - 5, n0)], 1).astype(np.float32)
    k0 = np.round(k0)
    s0 = rng.uniform(0.006, 0.6, n0).astype(np.float32)
    d0 = unit(rng.standard_normal((n0, 256)))
    k1 = np.round(np.stack([rng.uniform(4, w1 - 5, n1), rng.uniform(4, h1 - 5, n1)], 1)).astype(np.float32)
    s1 = rng.uniform(0.006, 0.6, n1).astype(np.float32)
    d1 = unit(rng.standard_normal((n1, 256)))
    m = int(overlap * min(n0, n1))
    gt = -np.ones(n0, dtype=np.int64)
    if m > 0:
        src = rng.permutation(n0)[:m]
        dst = rng.permutation(n1)[:m]
        shift = np.array([7.0, -5.0], dtype=np.float32)
        k1[dst] = np.clip(k0[src] + shift + np.round(rng.normal(0, 1.0, (m, 2))), 4, [w1 - 5, h1 - 5]).astype(np.float32)
        d1[dst] = unit(d0[src] + noise * unit(rng.standard_normal((m, 256))))
        s1[dst] = np.clip(s0[src] * rng.uniform(0.8, 1.2, m), 0.0051, 1.0).astype(np.float32)
        gt[src] = dst
    return k0, s0, d0, k1, s1, d1, gt


def _rotation_about(axis: np.ndarray, angle: float) -> np.ndarray:
    axis = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    k = np.array([[0.0, -axis[2], axis[1]], [axis[2], 0.0, -axis[0]], [-axis[1], axis[0], 0.0]])
    return np.eye(3) + np.sin(angle) * k + (1.0 - np.cos(angle)) * (k @ k)


def synthetic_two_view_matches(
    num_matches: int, outlier_ratio: float = 0.0, noise_px: float = 0.0, seed: int = 0, fx: float = 800.0, width: int = 1024, height: int = 1024,
    num_extra_keypoints: int = 0,
) -> Dict[str, np.ndarray]:
    """A two-view scene for the verifier stage: 3-D points in a slab 6-14 units in front of camera 1, a second camera
    rotated by 0.25 rad about a seeded axis and displaced by a unit vector, pinhole projection with (fx, fx, w/2, h/2),
    Gaussian pixel noise, and the first ``outlier_ratio`` share of the matches re-pointed at random pixels of image 2.
    Keypoint tables are shuffled (and padded with unmatched keypoints) so match indices are not the identity.

    Returns coordinates_i1/2 float32 (N1|N2, 2), match_indices int32 (M, 2), is_inlier bool (M), i2Ri1 (3,3), i2Ui1 (3,),
    intrinsics (fx, fy, cx, cy)."""
    rng = np.random.default_rng(seed)
    cx, cy = width / 2.0, height / 2.0
    pts = np.stack([rng.uniform(-4, 4, num_matches), rng.uniform(-3, 3, num_matches), rng.uniform(6, 14, num_matches)], 1)
    rot = _rotation_about(rng.normal(size=3), 0.25)
    trans = rng.normal(size=3)
    trans /= np.linalg.norm(trans)
    p2 = pts @ rot.T + trans
    uv1 = pts[:, :2] / pts[:, 2:] * fx + [cx, cy] + noise_px * rng.normal(size=(num_matches, 2))
    uv2 = p2[:, :2] / p2[:, 2:] * fx + [cx, cy] + noise_px * rng.normal(size=(num_matches, 2))
    num_out = int(round(outlier_ratio * num_matches))
    uv2[:num_out] = rng.uniform([0, 0], [width, height], size=(num_out, 2))
    is_inlier = np.arange(num_matches) >= num_out
    extra1 = rng.uniform([0, 0], [width, height], size=(num_extra_keypoints, 2))
    extra2 = rng.uniform([0, 0], [width, height], size=(num_extra_keypoints, 2))
    perm1 = rng.permutation(num_matches + num_extra_keypoints)
    perm2 = rng.permutation(num_matches + num_extra_keypoints)
    c1 = np.concatenate([uv1, extra1], 0)[perm1].astype(np.float32)
    c2 = np.concatenate([uv2, extra2], 0)[perm2].astype(np.float32)
    inv1, inv2 = np.argsort(perm1), np.argsort(perm2)
    order = rng.permutation(num_matches)
    match_indices = np.stack([inv1[:num_matches], inv2[:num_matches]], 1)[order].astype(np.int32)
    return {
        "coordinates_i1": c1, "coordinates_i2": c2, "match_indices": match_indices, "is_inlier": is_inlier[order], "i2Ri1": rot,
        "i2Ui1": trans, "intrinsics": (fx, fx, cx, cy),
    }
