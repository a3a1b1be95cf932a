"""Stand-in for ``FrontEndPipeline`` WITHOUT kernels -- plumbing tests only (``bench.py --plumbing-only``, the world-2 / world-8 gloo
tests of the sharded correspondence generator). It produces deterministic pseudo-features from an image's bytes and pseudo-match lists
from two images' pseudo-features on CPU tensors, so that everything AROUND the kernels -- launchers, partitioning, the feature exchange,
the ragged gathers, the timing protocol -- runs where there is no GPU. Nothing here computes SuperPoint / SuperGlue / LightGlue, no
product path selects it by itself (a caller must pass it in explicitly), and lines printed with it are marked ``plumbing_only``."""

from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch


class StandInPipeline:
    """``detect`` / ``detect_image_objects`` / ``match`` / ``matches_to_numpy`` with the shapes and dtypes of ``FrontEndPipeline``.
    An image's pseudo-features depend on its bytes alone and a pair's pseudo-matches on the two images' pseudo-features alone, so a
    sharded run and a single-process run of the same scene must agree -- which is what the plumbing tests assert."""

    def __init__(self, k: int = 32):
        self.k = int(k)
        self.max_keypoints = self.k
        self.last_shared_images = 0

    def _features_of(self, array: np.ndarray) -> Tuple[int, torch.Tensor, torch.Tensor, torch.Tensor]:
        a = np.ascontiguousarray(array)
        seed = int(a.astype(np.int64).sum()) * 31 + int(a.size)
        g = torch.Generator().manual_seed(seed % (2**31))
        count = 0 if (a.size and int(a.reshape(-1)[0]) == 255) else self.k - seed % 3  # a leading 255 marks an image without keypoints
        xy, sc, de = torch.rand((self.k, 2), generator=g), torch.rand((self.k,), generator=g), torch.rand((self.k, 256), generator=g)
        xy[count:], sc[count:], de[count:] = 0, 0, 0
        return count, xy, sc, de

    def _table(self, arrays: Sequence[np.ndarray]) -> Dict[str, torch.Tensor]:
        n = len(arrays)
        out = {"count": torch.zeros((n,), dtype=torch.int32), "xy": torch.zeros((n, self.k, 2)), "scores": torch.zeros((n, self.k)),
               "descriptors": torch.zeros((n, self.k, 256))}
        for i, a in enumerate(arrays):
            c, xy, sc, de = self._features_of(a)
            out["count"][i], out["xy"][i], out["scores"][i], out["descriptors"][i] = c, xy, sc, de
        return out

    def detect(self, images: torch.Tensor, image_chunk: int = 16) -> Dict[str, torch.Tensor]:
        return self._table([images[i].cpu().numpy() for i in range(images.shape[0])])

    def detect_image_objects(self, imgs: Sequence, image_batch: int = 16) -> Dict[str, torch.Tensor]:
        return self._table([np.asarray(im.value_array) for im in imgs])

    def match(self, feats, pairs, shapes, counts=None, **kw) -> List[Dict[str, torch.Tensor]]:
        if counts is None:
            counts = feats["count"].cpu().numpy()
        res = []
        for c0 in range(0, len(pairs), 32):
            chunk = [(int(i), int(j)) for i, j in pairs[c0 : c0 + 32]]
            n0, n1 = [int(counts[i]) for i, _ in chunk], [int(counts[j]) for _, j in chunk]
            m = torch.full((sum(n0) + sum(n1),), -1, dtype=torch.int32)
            row = 0
            for (i, j), a, b in zip(chunk, n0, n1):
                t = (int(float(feats["scores"][i, 0]) * 997) + int(float(feats["scores"][j, 0]) * 991)) % (min(a, b) + 1)
                idx = torch.arange(t, dtype=torch.int32)
                m[row : row + t] = (idx + 1) % max(1, b) if t < b else idx  # keypoint q of image i <-> keypoint q + 1 of image j
                row += a + b
            res.append({"matches": m, "mscores": torch.zeros(m.shape), "pairs": chunk, "n0": n0, "n1": n1})
        return res

    @staticmethod
    def matches_to_numpy(results, dtype=np.int64):
        from gtsfm_amd.runtime.pipeline import FrontEndPipeline

        return FrontEndPipeline.matches_to_numpy(results, dtype=dtype)


def stand_in_pipeline_factory(generator, device) -> StandInPipeline:
    """``pipeline_factory`` for ``ShardedDetDescCorrespondenceGenerator`` in plumbing tests (picklable: a module-level function)."""
    return StandInPipeline(min(int(getattr(generator._detector_descriptor, "max_keypoints", 32)), 32))


class FailingStandInPipeline(StandInPipeline):
    """Failure injection for the sharded generator's tests: on rank ``GTSFM_STANDIN_FAIL_RANK`` the ``GTSFM_STANDIN_FAIL_PHASE`` phase
    ("detect" / "match") either raises (``GTSFM_STANDIN_FAIL_HOW=raise``, default) or kills the process the way an outside signal would
    (``=exit``: no exception, no message, no agreement with the peers)."""

    def _maybe_fail(self, phase: str) -> None:
        import os

        from gtsfm_amd import parallel

        rank, _ = parallel.world_info()
        if os.environ.get("GTSFM_STANDIN_FAIL_PHASE") == phase and int(os.environ.get("GTSFM_STANDIN_FAIL_RANK", "-1")) == rank:
            if os.environ.get("GTSFM_STANDIN_FAIL_HOW", "raise") == "exit":
                os._exit(3)
            raise ValueError(f"injected failure in {phase} on rank {rank}")

    def detect_image_objects(self, imgs, image_batch: int = 16):
        self._maybe_fail("detect")
        return super().detect_image_objects(imgs, image_batch)

    def match(self, feats, pairs, shapes, counts=None, **kw):
        self._maybe_fail("match")
        return super().match(feats, pairs, shapes, counts=counts, **kw)


def failing_stand_in_pipeline_factory(generator, device) -> FailingStandInPipeline:
    return FailingStandInPipeline(min(int(getattr(generator._detector_descriptor, "max_keypoints", 32)), 32))
