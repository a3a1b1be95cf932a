"""ORACLE (test infrastructure, not product code): CPU restatement of the reference SuperPoint forward pass.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product path (``gtsfm_amd``) never does and fails loudly when the HIP library is missing.

Follows ``thirdparty/SuperGluePretrainedNetwork/models/superpoint.py`` of the reference (paths relative to
``/root/reference``), function by function, on plain ``torch`` CPU ops with an explicit ``state_dict``:

* ``simple_nms``          -> superpoint.py:47-62
* ``remove_borders``      -> superpoint.py:65-70
* ``sample_descriptors``  -> superpoint.py:80-92  (``align_corners=True`` forced: the reference pins torch 2.7.x
  where ``int(torch.__version__[2]) > 2`` holds; under this container's torch 2.10 the quirk picks the wrong
  branch -- SURVEY.md F3)
* ``superpoint_forward``  -> superpoint.py:145-202 (``max_keypoints=-1`` as GTSfM runs it, so the in-model top-k
  at :181-184 is not executed)
* ``detect_and_describe`` -> gtsfm/frontend/detector_descriptor/superpoint.py:63-93 (wrapper marshalling)

Pinned by ``oracle/validate_against_reference.py`` (bit-exact against the reference model file executed in this
container on the same synthetic weights) and the golden vectors in ``tests/golden/``.
"""

from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]


def simple_nms(scores: torch.Tensor, nms_radius: int) -> torch.Tensor:
    """superpoint.py:47-62 -- 1 + 2*3 max-pools of (2r+1)^2, equality masks, two suppression rounds."""
    assert nms_radius >= 0

    def max_pool(x):
        return F.max_pool2d(x, kernel_size=nms_radius * 2 + 1, stride=1, padding=nms_radius)

    zeros = torch.zeros_like(scores)
    max_mask = scores == max_pool(scores)
    for _ in range(2):
        supp_mask = max_pool(max_mask.to(scores.dtype)) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == max_pool(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


def remove_borders(keypoints: torch.Tensor, scores: torch.Tensor, border: int, height: int, width: int):
    """superpoint.py:65-70."""
    mask_h = (keypoints[:, 0] >= border) & (keypoints[:, 0] < (height - border))
    mask_w = (keypoints[:, 1] >= border) & (keypoints[:, 1] < (width - border))
    mask = mask_h & mask_w
    return keypoints[mask], scores[mask]


def sample_descriptors(keypoints: torch.Tensor, descriptors: torch.Tensor, s: int = 8) -> torch.Tensor:
    """superpoint.py:80-92 with ``align_corners=True``. keypoints [b,K,2] (x,y); descriptors
    [b,c,h,w]. Returns [b,c,K]."""
    b, c, h, w = descriptors.shape
    keypoints = keypoints - s / 2 + 0.5
    keypoints = keypoints / torch.tensor([(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)]).to(keypoints)[None]
    keypoints = keypoints * 2 - 1
    descriptors = F.grid_sample(descriptors, keypoints.view(b, 1, -1, 2), mode="bilinear", align_corners=True)
    descriptors = F.normalize(descriptors.reshape(b, c, -1), p=2, dim=1)
    return descriptors


def _conv(sd: StateDict, name: str, x: torch.Tensor, pad: int) -> torch.Tensor:
    return F.conv2d(x, sd[f"{name}.weight"], sd[f"{name}.bias"], stride=1, padding=pad)


def superpoint_forward(
    sd: StateDict,
    image: torch.Tensor,
    nms_radius: int = 4,
    keypoint_threshold: float = 0.005,
    border: int = 4,
    return_intermediates: bool = False,
) -> Dict[str, torch.Tensor]:
    """superpoint.py:145-202 for a single image tensor [1,1,H,W] (dtype decides fp32 / fp64 arithmetic).

    Returns keypoints [K,2] (x,y) float, scores [K], descriptors [256,K]; with ``return_intermediates`` also the
    encoder output, detector logits, dense (pre-NMS) score map, NMS output and the normalised dense descriptors.
    """
    assert image.dim() == 4 and image.shape[0] == 1 and image.shape[1] == 1
    sd = {k: v.to(image.dtype) if v.is_floating_point() else v for k, v in sd.items()}
    relu = F.relu
    x = relu(_conv(sd, "conv1a", image, 1))
    x = relu(_conv(sd, "conv1b", x, 1))
    x = F.max_pool2d(x, kernel_size=2, stride=2)
    x = relu(_conv(sd, "conv2a", x, 1))
    x = relu(_conv(sd, "conv2b", x, 1))
    x = F.max_pool2d(x, kernel_size=2, stride=2)
    x = relu(_conv(sd, "conv3a", x, 1))
    x = relu(_conv(sd, "conv3b", x, 1))
    x = F.max_pool2d(x, kernel_size=2, stride=2)
    x = relu(_conv(sd, "conv4a", x, 1))
    x = relu(_conv(sd, "conv4b", x, 1))

    cPa = relu(_conv(sd, "convPa", x, 1))
    logits = _conv(sd, "convPb", cPa, 0)
    scores = F.softmax(logits, 1)[:, :-1]
    b, _, h, w = scores.shape
    scores = scores.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8)
    dense_scores = scores.permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    nms_scores = simple_nms(dense_scores, nms_radius)

    s = nms_scores[0]
    kp = torch.nonzero(s > keypoint_threshold)
    sc = s[tuple(kp.t())]
    kp, sc = remove_borders(kp, sc, border, h * 8, w * 8)
    kp = torch.flip(kp, [1]).to(image.dtype)

    cDa = relu(_conv(sd, "convDa", x, 1))
    dense_desc = _conv(sd, "convDb", cDa, 0)
    dense_desc = F.normalize(dense_desc, p=2, dim=1)
    desc = sample_descriptors(kp[None], dense_desc, 8)[0]

    out = {"keypoints": kp, "scores": sc, "descriptors": desc}
    if return_intermediates:
        out.update(
            encoder=x, logits=logits, dense_scores=dense_scores, nms_scores=nms_scores, dense_descriptors=dense_desc
        )
    return out


def gray_u8_to_tensor(gray: np.ndarray, dtype=torch.float32) -> torch.Tensor:
    """gtsfm/frontend/detector_descriptor/superpoint.py:73-75: ``astype(float32) / 255.0`` + batch/channel dims."""
    assert gray.ndim == 2
    return torch.from_numpy(np.expand_dims(gray.astype(np.float32) / 255.0, (0, 1))).to(dtype)


def detect_and_describe(
    sd: StateDict, gray: np.ndarray, max_keypoints: int = 5000, mask: Optional[np.ndarray] = None, dtype=torch.float32
):
    """gtsfm/frontend/detector_descriptor/superpoint.py:63-93 on an already-gray uint8 image (cv2 is absent here,
    SURVEY.md section 8c caveat 4). Returns (coordinates [K,2] f32, responses [K] f32, descriptors [K,256] f32)
    after ``filter_by_mask`` / ``get_top_k`` (gtsfm/common/keypoints.py:89-127). Pinned: the reference's own wrapper class, run live on gray
    frames, returns the same arrays bit for bit (oracle/validate_wrappers_against_reference.py)."""
    with torch.no_grad():
        res = superpoint_forward(sd, gray_u8_to_tensor(gray, dtype))
    coordinates = res["keypoints"].to(torch.float32).numpy()
    scores = res["scores"].to(torch.float32).numpy()
    descriptors = res["descriptors"].to(torch.float32).numpy().T
    if mask is not None:
        rounded = np.round(coordinates).astype(int)
        valid = np.flatnonzero(mask[rounded[:, 1], rounded[:, 0]] == 1)
        coordinates, scores, descriptors = coordinates[valid], scores[valid], descriptors[valid]
    if max_keypoints < len(coordinates):
        sel = np.argpartition(-scores, max_keypoints)[:max_keypoints]
        coordinates, scores, descriptors = coordinates[sel], scores[sel], descriptors[sel]
    return coordinates, scores, descriptors
