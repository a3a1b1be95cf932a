"""ORACLE (test infrastructure, not product code): CPU restatement of the reference SuperGlue forward pass.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Follows ``thirdparty/SuperGluePretrainedNetwork/models/superglue.py`` of the reference (paths relative to
``/root/reference``) on plain ``torch`` CPU ops with an explicit ``state_dict``:

* ``normalize_keypoints``        -> superglue.py:63-70
* ``_mlp`` / keypoint encoder    -> superglue.py:49-60,73-82 (Conv1d + eval-mode BatchNorm1d + ReLU)
* ``attention`` / MHA            -> superglue.py:85-107 (``view(b, 64, 4, N)``: head is the FAST channel axis)
* ``_propagation`` / GNN         -> superglue.py:110-138
* ``log_sinkhorn_iterations``    -> superglue.py:141-147
* ``log_optimal_transport``      -> superglue.py:150-170
* ``superglue_forward``          -> superglue.py:228-283
* ``match``                      -> gtsfm/frontend/matcher/superglue_matcher.py:47-115 (wrapper marshalling)

Pinned by ``oracle/validate_against_reference.py`` (bit-exact against the reference model file executed in this
container on the same synthetic weights) and the golden vectors in ``tests/golden/``.
"""

from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]


def normalize_keypoints(kpts: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """superglue.py:63-70."""
    one = kpts.new_tensor(1)
    size = torch.stack([one * width, one * height])[None]
    center = size / 2
    scaling = size.max(1, keepdim=True).values * 0.7
    return (kpts - center[:, None, :]) / scaling[:, None, :]


def _conv1d(sd: StateDict, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.conv1d(x, sd[f"{name}.weight"], sd[f"{name}.bias"])


def _bn(sd: StateDict, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(
        x, sd[f"{name}.running_mean"], sd[f"{name}.running_var"], sd[f"{name}.weight"], sd[f"{name}.bias"],
        training=False, momentum=0.1, eps=1e-5,
    )


def _mlp(sd: StateDict, prefix: str, n_linear: int, x: torch.Tensor) -> torch.Tensor:
    """superglue.py:49-60: Conv1d, then (BatchNorm1d, ReLU) after all but the last."""
    for i in range(n_linear):
        x = _conv1d(sd, f"{prefix}.{3 * i}", x)
        if i < n_linear - 1:
            x = F.relu(_bn(sd, f"{prefix}.{3 * i + 1}", x))
    return x


def attention(query, key, value):
    """superglue.py:85-89."""
    dim = query.shape[1]
    scores = torch.einsum("bdhn,bdhm->bhnm", query, key) / dim**0.5
    prob = F.softmax(scores, dim=-1)
    return torch.einsum("bhnm,bdhm->bdhn", prob, value), prob


def _mha(sd: StateDict, prefix: str, query, key, value, num_heads: int = 4):
    """superglue.py:92-107."""
    b = query.size(0)
    d_model = query.size(1)
    dim = d_model // num_heads
    q, k, v = [
        _conv1d(sd, f"{prefix}.proj.{j}", x).view(b, dim, num_heads, -1) for j, x in enumerate((query, key, value))
    ]
    x, _ = attention(q, k, v)
    return _conv1d(sd, f"{prefix}.merge", x.contiguous().view(b, dim * num_heads, -1))


def _propagation(sd: StateDict, prefix: str, x, source):
    """superglue.py:110-119."""
    message = _mha(sd, f"{prefix}.attn", x, source, source)
    return _mlp(sd, f"{prefix}.mlp", 2, torch.cat([x, message], dim=1))


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters: int):
    """superglue.py:141-147."""
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def log_optimal_transport(scores, alpha, iters: int):
    """superglue.py:150-170."""
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    alpha = alpha.expand(b, 1, 1)
    couplings = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, alpha], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])
    log_mu, log_nu = log_mu[None].expand(b, -1), log_nu[None].expand(b, -1)
    Z = log_sinkhorn_iterations(couplings, log_mu, log_nu, iters)
    return Z - norm


def gnn_layer_names(sd: StateDict) -> List[str]:
    n = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("gnn.layers."))
    return (["self", "cross"] * ((n + 1) // 2))[:n]


def superglue_forward(
    sd: StateDict,
    kpts0: torch.Tensor,
    kpts1: torch.Tensor,
    scores0: torch.Tensor,
    scores1: torch.Tensor,
    desc0: torch.Tensor,
    desc1: torch.Tensor,
    shape0: Tuple[int, int],
    shape1: Tuple[int, int],
    sinkhorn_iterations: int = 20,
    match_threshold: float = 0.2,
    return_intermediates: bool = False,
) -> Dict[str, torch.Tensor]:
    """superglue.py:228-283. kpts [1,N,2] (x,y); scores [1,N]; desc [1,256,N]; shape = (H, W)."""
    if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:  # superglue.py:233-240
        shp0, shp1 = kpts0.shape[:-1], kpts1.shape[:-1]
        return {
            "matches0": kpts0.new_full(shp0, -1, dtype=torch.int),
            "matches1": kpts1.new_full(shp1, -1, dtype=torch.int),
            "matching_scores0": kpts0.new_zeros(shp0),
            "matching_scores1": kpts1.new_zeros(shp1),
        }
    dt = desc0.dtype
    sd = {k: v.to(dt) if v.is_floating_point() else v for k, v in sd.items()}
    kpts0 = normalize_keypoints(kpts0, shape0[0], shape0[1])
    kpts1 = normalize_keypoints(kpts1, shape1[0], shape1[1])

    def kenc(kpts, scores):
        inputs = [kpts.transpose(1, 2), scores.unsqueeze(1)]
        return _mlp(sd, "kenc.encoder", 5, torch.cat(inputs, dim=1))

    desc0 = desc0 + kenc(kpts0, scores0)
    desc1 = desc1 + kenc(kpts1, scores1)
    enc0, enc1 = desc0, desc1

    for l, name in enumerate(gnn_layer_names(sd)):
        if name == "cross":
            src0, src1 = desc1, desc0
        else:
            src0, src1 = desc0, desc1
        delta0 = _propagation(sd, f"gnn.layers.{l}", desc0, src0)
        delta1 = _propagation(sd, f"gnn.layers.{l}", desc1, src1)
        desc0, desc1 = (desc0 + delta0), (desc1 + delta1)

    mdesc0, mdesc1 = _conv1d(sd, "final_proj", desc0), _conv1d(sd, "final_proj", desc1)
    scores = torch.einsum("bdn,bdm->bnm", mdesc0, mdesc1)
    scores = scores / desc0.shape[1] ** 0.5
    raw_scores = scores
    scores = log_optimal_transport(scores, sd["bin_score"], iters=sinkhorn_iterations)

    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    indices0, indices1 = max0.indices, max1.indices
    ar0 = torch.arange(indices0.shape[1])[None]
    ar1 = torch.arange(indices1.shape[1])[None]
    mutual0 = ar0 == indices1.gather(1, indices0)
    mutual1 = ar1 == indices0.gather(1, indices1)
    zero = scores.new_tensor(0)
    mscores0 = torch.where(mutual0, max0.values.exp(), zero)
    mscores1 = torch.where(mutual1, mscores0.gather(1, indices1), zero)
    valid0 = mutual0 & (mscores0 > match_threshold)
    valid1 = mutual1 & valid0.gather(1, indices1)
    indices0 = torch.where(valid0, indices0, indices0.new_tensor(-1))
    indices1 = torch.where(valid1, indices1, indices1.new_tensor(-1))
    out = {
        "matches0": indices0,
        "matches1": indices1,
        "matching_scores0": mscores0,
        "matching_scores1": mscores1,
    }
    if return_intermediates:
        out.update(enc0=enc0, enc1=enc1, gnn0=desc0, gnn1=desc1, mdesc0=mdesc0, mdesc1=mdesc1, scores=raw_scores, ot=scores)
    return out


def match(
    sd: StateDict,
    coords0: np.ndarray,
    coords1: np.ndarray,
    resp0: np.ndarray,
    resp1: np.ndarray,
    desc0: np.ndarray,
    desc1: np.ndarray,
    im_shape0: Tuple[int, ...],
    im_shape1: Tuple[int, ...],
    sinkhorn_iterations: int = 20,
    dtype=torch.float32,
) -> np.ndarray:
    """gtsfm/frontend/matcher/superglue_matcher.py:75-113: numpy -> torch dict -> model -> (K,2) uint32. Pinned: the reference's own
    ``SuperGlueMatcher.match``, run live, returns the same arrays (oracle/validate_wrappers_against_reference.py)."""
    with torch.no_grad():
        pred = superglue_forward(
            sd,
            torch.from_numpy(coords0).unsqueeze(0).float().to(dtype),
            torch.from_numpy(coords1).unsqueeze(0).float().to(dtype),
            torch.from_numpy(resp0).unsqueeze(0).float().to(dtype),
            torch.from_numpy(resp1).unsqueeze(0).float().to(dtype),
            torch.from_numpy(desc0).T.unsqueeze(0).float().to(dtype),
            torch.from_numpy(desc1).T.unsqueeze(0).float().to(dtype),
            (im_shape0[0], im_shape0[1]),
            (im_shape1[0], im_shape1[1]),
            sinkhorn_iterations=sinkhorn_iterations,
        )
    matches = pred["matches0"][0].numpy()
    valid = matches > -1
    return np.hstack(
        [np.arange(len(coords0))[valid].reshape(-1, 1), np.arange(len(coords1))[matches[valid]].reshape(-1, 1)]
    ).astype(np.uint32)
