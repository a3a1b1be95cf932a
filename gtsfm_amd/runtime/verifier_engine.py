"""Host side of the device verifier stage: batches of image pairs through
``gtsfm_verify_essential_f64`` -- five-point RANSAC on the squared Sampson error plus the cheirality choice of
``cv.recoverPose``, replacing the per-pair OpenCV calls of ``gtsfm/frontend/verifier/opencv_verifier_base.py:47-111`` /
``ransac.py:52-84`` (called from ``gtsfm/two_view_estimator.py:391-397``). PyTorch provides device memory and streams only.
PARITY UNPINNED (OpenCV absent; see ``oracle/verifier_oracle.py``)."""

from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from gtsfm_amd.runtime import lib as _lib
from gtsfm_amd.runtime.superpoint_engine import require_gpu

_NUMPY_DTYPE = {torch.int32: np.int32, torch.int64: np.int64, torch.float64: np.float64}
STATS_FIELDS = ("inliers", "hypotheses", "winner_hypothesis", "winner_root", "good_r1_t", "good_r2_t", "good_r1_mt", "good_r2_mt")


class VerifierEngine:
    """Stateless apart from a cached workspace; one instance per process / GPU."""

    def __init__(self, device: Optional[torch.device] = None):
        self.device = require_gpu(device)
        self._lib = _lib.load()
        self._ws: Optional[torch.Tensor] = None

    def _workspace(self, total_matches: int) -> torch.Tensor:
        need = int(self._lib.gtsfm_verify_workspace_bytes(total_matches))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self._ws

    def _dev(self, a, dtype) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=_NUMPY_DTYPE[dtype])).to(self.device)

    def verify_batch(
        self,
        kp_xy: torch.Tensor,
        kp_off1: Sequence[int],
        kp_off2: Sequence[int],
        match_idx: torch.Tensor,
        match_off: Sequence[int],
        intrinsics: np.ndarray,
        threshold_px: float,
        seeds: Optional[Sequence[int]] = None,
        match_count: Optional[torch.Tensor] = None,
        use_intrinsics: bool = True,
    ) -> Dict[str, torch.Tensor]:
        """kp_xy [*,2] float32 pixel coordinates (device); pair p reads image i1's rows from kp_off1[p] and image i2's from
        kp_off2[p]; match_idx [M,2] int32 (device) rows relative to those offsets, pair p owning match_off[p]:match_off[p+1];
        intrinsics [P,8] = (fx, fy, cx, cy) of i1 then i2; match_count [P] int32 (device, optional): only the first
        match_count[p] rows of pair p's slice are matches (capacity layout of ``compact_matches``). ``use_intrinsics=False``:
        fundamental-matrix estimation on pixel coordinates (adds "F" [P,3,3]; E = K2^T F K1). Returns device tensors: E [P,3,3],
        R [P,3,3], t [P,3] (NaN when a pair has no model), mask [M] uint8, stats [P,8] int32 (``STATS_FIELDS``). Enqueued on the current stream."""
        num_pairs = len(kp_off1)
        assert len(kp_off2) == num_pairs and len(match_off) == num_pairs + 1
        assert kp_xy.is_cuda and kp_xy.dtype == torch.float32 and kp_xy.is_contiguous()
        total = int(match_off[-1])
        assert match_count is None or (match_count.dtype == torch.int32 and match_count.is_cuda and match_count.numel() == num_pairs)
        assert match_idx.dtype == torch.int32 and match_idx.is_contiguous() and match_idx.numel() == 2 * total
        dev = self.device
        off1, off2 = self._dev(kp_off1, torch.int64), self._dev(kp_off2, torch.int64)
        moff = self._dev(match_off, torch.int64)
        intr = self._dev(np.asarray(intrinsics, dtype=np.float64).reshape(num_pairs, 8), torch.float64)
        seed_arr = np.zeros(num_pairs, dtype=np.uint64) if seeds is None else np.asarray(seeds, dtype=np.uint64)
        seeds_dev = torch.from_numpy(seed_arr.view(np.int64)).to(dev)
        out = {
            "E": torch.empty((num_pairs, 3, 3), dtype=torch.float64, device=dev),
            "R": torch.empty((num_pairs, 3, 3), dtype=torch.float64, device=dev),
            "t": torch.empty((num_pairs, 3), dtype=torch.float64, device=dev),
            "mask": torch.empty(total, dtype=torch.uint8, device=dev),
            "stats": torch.empty((num_pairs, 8), dtype=torch.int32, device=dev),
        }
        if not use_intrinsics:
            out["F"] = torch.empty((num_pairs, 3, 3), dtype=torch.float64, device=dev)
        if num_pairs == 0:
            return out
        ws = self._workspace(total)
        head = (kp_xy.data_ptr(), off1.data_ptr(), off2.data_ptr(), match_idx.data_ptr() if total else None, moff.data_ptr(),
                match_count.data_ptr() if match_count is not None else None, total, intr.data_ptr(), seeds_dev.data_ptr(), float(threshold_px),
                num_pairs, ws.data_ptr(), ws.numel())
        tail = (out["E"].data_ptr(), out["R"].data_ptr(), out["t"].data_ptr(), out["mask"].data_ptr() if total else None,
                out["stats"].data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if use_intrinsics:
            _lib.check(self._lib.gtsfm_verify_essential_f64(*head, *tail), "gtsfm_verify_essential_f64")
        else:
            _lib.check(self._lib.gtsfm_verify_fundamental_f64(*head, out["F"].data_ptr(), *tail), "gtsfm_verify_fundamental_f64")
        return out

    def compact_matches(self, matches: torch.Tensor, row_off: Sequence[int], n0: Sequence[int]):
        """Matcher output [T] int32 (per pair: matches0 then matches1) -> (match_idx [sum(n0),2] int32, match_off host list
        [P+1] = capacity prefix of n0, match_count [P] int32 device): the (K, 2) arrays of the plugins, left on the device."""
        num_pairs = len(n0)
        assert matches.dtype == torch.int32 and matches.is_cuda and matches.is_contiguous() and len(row_off) == num_pairs
        match_off = np.concatenate([[0], np.cumsum(np.asarray(n0, dtype=np.int64))]).astype(np.int64)
        idx = torch.empty((int(match_off[-1]), 2), dtype=torch.int32, device=self.device)
        count = torch.zeros(num_pairs, dtype=torch.int32, device=self.device)
        if num_pairs:
            rows_dev, n0_dev, off_dev = self._dev(row_off, torch.int64), self._dev(n0, torch.int32), self._dev(match_off, torch.int64)  # keep alive
            _lib.check(
                self._lib.gtsfm_verify_compact_matches(
                    matches.data_ptr(), rows_dev.data_ptr(), n0_dev.data_ptr(), off_dev.data_ptr(), num_pairs, idx.data_ptr(), count.data_ptr(),
                    torch.cuda.current_stream(self.device).cuda_stream,
                ),
                "gtsfm_verify_compact_matches",
            )
        return idx, match_off.tolist(), count
