"""Host side of the SuperPoint HIP path: weight packing/upload, workspace management, kernel enqueue.

PyTorch is used for device memory and streams only; all arithmetic runs in libgtsfm_amd.so
(``gtsfm_sp_forward``, replacing ``thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:145-202``).
"""

from __future__ import annotations

import contextlib
import ctypes as C
import os
import queue
import threading
from collections import OrderedDict
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from gtsfm_amd.runtime import lib as _lib

# checkpoint order of superpoint_v1.pth (superpoint.py:119-134)
SUPERPOINT_KEYS = [
    f"{name}.{kind}"
    for name in (
        "conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa",
        "convDb",
    )
    for kind in ("weight", "bias")
]

DEFAULT_NMS_RADIUS = 4  # superpoint.py:111-115 default_config
DEFAULT_KEYPOINT_THRESHOLD = 0.005
DEFAULT_REMOVE_BORDERS = 4


def require_gpu(device: Optional[torch.device] = None) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("gtsfm_amd requires an AMD GPU visible to PyTorch-ROCm; there is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def pack_superpoint_weights(state_dict: Mapping[str, torch.Tensor]) -> np.ndarray:
    """state_dict (reference checkpoint names/layouts) -> packed fp32 blob (host)."""
    lib = _lib.load()
    missing = [k for k in SUPERPOINT_KEYS if k not in state_dict]
    if missing:
        raise KeyError(f"SuperPoint state_dict is missing {missing}")
    tensors = [np.ascontiguousarray(state_dict[k].detach().cpu().numpy().astype(np.float32)) for k in SUPERPOINT_KEYS]
    arr = (C.c_void_p * len(tensors))(*[t.ctypes.data for t in tensors])
    out = np.empty(lib.gtsfm_sp_packed_weight_floats(), dtype=np.float32)
    _lib.check(lib.gtsfm_sp_pack_weights(arr, out.ctypes.data), "gtsfm_sp_pack_weights")
    return out


WORKSPACE_SHAPES_KEPT = 6  # workspaces cached per lane, least recently used evicted one at a time (mixed portrait / landscape scenes)


class _DetectLane:
    """What one in-flight ``detect_lazy`` call owns: page-locked staging buffers, workspaces, a HIP stream. The weights are shared."""

    def __init__(self) -> None:
        self.pinned: Dict[str, torch.Tensor] = {}
        self.workspaces: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()
        self.stream: Optional[torch.cuda.Stream] = None
        self.lock = threading.Lock()  # the staging buffers serve the call AND its later fetch(): one at a time


class SuperPointEngine:
    """Device-resident SuperPoint: packed weights in HBM + cached workspaces."""

    def __init__(self, state_dict: Mapping[str, torch.Tensor], device: Optional[torch.device] = None):
        self.device = require_gpu(device)
        self._lib = _lib.load()
        self.weights = torch.from_numpy(pack_superpoint_weights(state_dict)).to(self.device)
        self._init_lanes()

    @classmethod
    def from_packed(cls, packed: torch.Tensor) -> "SuperPointEngine":
        """Build from an already-packed device blob (e.g. received by an RCCL broadcast)."""
        self = cls.__new__(cls)
        self.device = packed.device
        self._lib = _lib.load()
        self.weights = packed
        self._init_lanes()
        return self

    def _init_lanes(self) -> None:
        # batched callers (the pipeline: one thread, the caller's stream) use the main lane's workspaces; detect_lazy() -- the
        # per-call plugin API, possibly from several worker threads (GTSfM's --threads_per_worker) -- takes a free lane, creating
        # up to GTSFM_PLUGIN_LANES of them, like the matcher engines (matcher_engine._MatcherBase._lane)
        self.max_lanes = max(1, int(os.environ.get("GTSFM_PLUGIN_LANES", "3")))
        self._main = _DetectLane()
        self._lanes = [self._main]
        self._free_lanes: "queue.LifoQueue" = queue.LifoQueue()
        self._free_lanes.put(self._main)
        self._lanes_lock = threading.Lock()

    @contextlib.contextmanager
    def _lane(self):
        try:
            lane = self._free_lanes.get_nowait()
        except queue.Empty:
            lane = None
            with self._lanes_lock:
                if len(self._lanes) < self.max_lanes:
                    lane = _DetectLane()
                    self._lanes.append(lane)
            if lane is None:
                lane = self._free_lanes.get()
        try:
            yield lane
        finally:
            self._free_lanes.put(lane)

    def release_lanes(self) -> None:
        """Drop the extra lanes (their pinned buffers and workspaces) and the main lane's cached workspaces; lanes in use survive."""
        with self._lanes_lock:
            idle = []
            while True:
                try:
                    idle.append(self._free_lanes.get_nowait())
                except queue.Empty:
                    break
            for lane in idle:
                if lane is not self._main:
                    self._lanes.remove(lane)
            self._main.workspaces.clear()
            if self._main in idle:
                self._free_lanes.put(self._main)

    def _workspace(self, b: int, h: int, w: int, lane: Optional[_DetectLane] = None) -> torch.Tensor:
        cache = (lane or self._main).workspaces
        key = (b, h, w)
        ws = cache.get(key)
        if ws is None:
            nbytes = self._lib.gtsfm_sp_workspace_bytes(b, h, w)
            while len(cache) >= WORKSPACE_SHAPES_KEPT:
                cache.popitem(last=False)  # the least recently used shape only (round 3 dropped all of them at the 6th shape)
            ws = cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        else:
            cache.move_to_end(key)
        return ws

    @staticmethod
    def default_capacity(h: int, w: int, nms_radius: int) -> int:
        h8, w8 = (h // 8) * 8, (w // 8) * 8
        r = nms_radius + 1
        return max(64, -(-h8 // r) * -(-w8 // r) + 64)

    def forward(
        self,
        images: torch.Tensor,
        capacity: Optional[int] = None,
        keypoint_threshold: float = DEFAULT_KEYPOINT_THRESHOLD,
        nms_radius: int = DEFAULT_NMS_RADIUS,
        remove_borders: int = DEFAULT_REMOVE_BORDERS,
        return_score_maps: bool = False,
        top_k: int = 0,
        valid_masks: Optional[torch.Tensor] = None,
        _lane: Optional[_DetectLane] = None,
    ) -> Dict[str, torch.Tensor]:
        """images: device tensor [B,H,W], uint8 or float32 in [0,1]; valid_masks (optional): device uint8 [B,H,W], 1 = valid --
        keypoints on other pixels are dropped before the top-k (``Keypoints.filter_by_mask``). Returns device tensors:
        count [B] int32, count_raw [B] int32, xy [B,cap,2], scores [B,cap], descriptors [B,cap,256]."""
        assert images.dim() == 3 and images.is_cuda and images.is_contiguous()
        assert images.dtype in (torch.uint8, torch.float32)
        b, h, w = images.shape
        cap = capacity or self.default_capacity(h, w, nms_radius)
        if top_k > 0:
            cap = top_k  # device-side top-k: the outputs hold at most top_k rows per image
        dev = images.device
        count = torch.empty(b, dtype=torch.int32, device=dev)
        count_raw = torch.empty(b, dtype=torch.int32, device=dev)
        xy = torch.empty((b, cap, 2), dtype=torch.float32, device=dev)
        scores = torch.empty((b, cap), dtype=torch.float32, device=dev)
        desc = torch.empty((b, cap, 256), dtype=torch.float32, device=dev)
        dense = nms = None
        if return_score_maps:
            h8, w8 = (h // 8) * 8, (w // 8) * 8
            dense = torch.empty((b, h8, w8), dtype=torch.float32, device=dev)
            nms = torch.empty((b, h8, w8), dtype=torch.float32, device=dev)
        ws = self._workspace(b, h, w, _lane)
        if valid_masks is not None:
            assert valid_masks.shape == images.shape and valid_masks.dtype == torch.uint8 and valid_masks.is_cuda and valid_masks.is_contiguous()
        rc = self._lib.gtsfm_sp_forward_masked(
            self.weights.data_ptr(), images.data_ptr(), int(images.dtype == torch.uint8), b, h, w,
            float(keypoint_threshold), int(nms_radius), int(remove_borders), cap, int(top_k), ws.data_ptr(), ws.numel(),
            count.data_ptr(), count_raw.data_ptr(), xy.data_ptr(), scores.data_ptr(), desc.data_ptr(),
            _lib.ptr(dense), _lib.ptr(nms), _lib.ptr(valid_masks), torch.cuda.current_stream(dev).cuda_stream,
        )
        _lib.check(rc, "gtsfm_sp_forward_masked")
        out = {"count": count, "count_raw": count_raw, "xy": xy, "scores": scores, "descriptors": desc}
        if return_score_maps:
            out["dense_scores"], out["nms_scores"] = dense, nms
        return out

    @staticmethod
    def _pinned(lane: _DetectLane, name: str, shape, dtype) -> torch.Tensor:
        """A page-locked host buffer that lives as long as its lane (grown geometrically): the per-call plugin API moves an image
        in and keypoints / descriptors out on every call, and must not depend on how fast the host maps fresh pageable memory."""
        need = int(np.prod(shape))
        buf = lane.pinned.get(name)
        if buf is None or buf.numel() < need or buf.dtype != dtype:
            buf = lane.pinned[name] = torch.empty(max(need, int(1.5 * (0 if buf is None else buf.numel()))), dtype=dtype, pin_memory=True)
        return buf[:need].view(*shape)

    def detect_lazy(self, gray: np.ndarray, **kwargs):
        """Single host image -> (coordinates [K,2], scores [K], fetch): the keypoint list in the model's row-major order on the
        host, the descriptors still on the device; ``fetch(indices)`` returns descriptor rows ``indices`` as a fresh [len,256] numpy
        array. The plugin selects on the host with the reference's own ``Keypoints`` methods and downloads only what it keeps
        (5000 of ~8000 rows at GTSfM's cap). Calls from several threads run side by side, each on a lane of its own (staging
        buffers, workspace, stream); round 3 serialised them behind one lock."""
        assert gray.ndim == 2
        with self._lane() as lane:
            if lane.stream is None:
                lane.stream = torch.cuda.Stream(self.device)
            with lane.lock, torch.cuda.stream(lane.stream):
                return self._detect_lazy_on(lane, gray, **kwargs)

    def _detect_lazy_on(self, lane: _DetectLane, gray: np.ndarray, **kwargs):
        stage = self._pinned(lane, "image", gray.shape, torch.uint8 if gray.dtype == np.uint8 else torch.float32)
        stage.numpy()[...] = gray
        img = stage.to(self.device, non_blocking=True)[None]
        out = self.forward(img, _lane=lane, **kwargs)
        k_raw = int(out["count_raw"][0].item())  # synchronises: the staged image has been consumed
        if k_raw > out["xy"].shape[1]:  # ties can exceed the NMS packing bound; rerun with an exact capacity
            out = self.forward(img, _lane=lane, **dict(kwargs, capacity=k_raw))
        k = int(out["count"][0].item())
        small = self._pinned(lane, "xy_scores", (k, 3), torch.float32)
        small[:, :2].copy_(out["xy"][0, :k], non_blocking=True)
        small[:, 2].copy_(out["scores"][0, :k], non_blocking=True)
        lane.stream.synchronize()
        host = small.numpy()
        desc = out["descriptors"][0]
        stream = lane.stream

        def fetch(indices: np.ndarray) -> np.ndarray:
            indices = np.asarray(indices, dtype=np.int64)
            with lane.lock, torch.cuda.stream(stream):  # the lane may serve another call by now: its buffers one at a time
                rows = self._pinned(lane, "descriptors", (len(indices), 256), torch.float32)
                if len(indices):
                    idx = self._pinned(lane, "indices", (len(indices),), torch.int64)
                    idx.numpy()[...] = indices
                    rows.copy_(desc.index_select(0, idx.to(self.device, non_blocking=True)), non_blocking=True)
                    stream.synchronize()
                return rows.numpy().copy()

        return host[:, :2].copy(), host[:, 2].copy(), fetch

    def detect(self, gray: np.ndarray, **kwargs):
        """Single host image (H,W) uint8 or float32 -> numpy (coordinates [K,2] f32 (x,y), scores [K], descriptors
        [K,256]) in the model's row-major order, i.e. what superpoint.py:198-202 returns before the GTSfM wrapper's
        host-side filtering."""
        assert gray.ndim == 2
        img = torch.from_numpy(np.ascontiguousarray(gray)).to(self.device)[None]
        out = self.forward(img, **kwargs)
        k_raw = int(out["count_raw"][0].item())
        cap = out["xy"].shape[1]
        if k_raw > cap:  # ties can exceed the NMS packing bound; rerun with an exact capacity
            kwargs = dict(kwargs, capacity=k_raw)
            out = self.forward(img, **kwargs)
        k = int(out["count"][0].item())
        return (
            out["xy"][0, :k].cpu().numpy(),
            out["scores"][0, :k].cpu().numpy(),
            out["descriptors"][0, :k].cpu().numpy(),
        )
