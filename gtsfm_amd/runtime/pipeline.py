"""GPU-resident detect+match pipeline: images in HBM -> SuperPoint (batched) -> device top-k -> matcher over a list of
image pairs, with features never leaving the device between the two stages.

This is the throughput path of the front-end (BASELINE.json's metric) and the engine behind the batched correspondence
generator. The reference runs one Dask task per image and per pair with an H2D/D2H round trip and a pickle each
(``gtsfm/frontend/correspondence_generator/det_desc_correspondence_generator.py:65-87``); here a whole cluster's
pairs are matched from one resident feature table.
"""

from __future__ import annotations

import contextlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from gtsfm_amd.runtime.matcher_engine import LightGlueEngine, SuperGlueEngine
from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine


class FrontEndPipeline:
    def __init__(self, detector: SuperPointEngine, matcher, max_keypoints: int = 5000, pair_chunk: int = 32, num_streams: int = 2):
        self.detector = detector
        self.matcher = matcher
        self.max_keypoints = max_keypoints
        self.pair_chunk = pair_chunk
        # pair chunks are independent: alternate them over HIP streams (each with its own workspace) so that one chunk's
        # small / tail-heavy kernels overlap the other's MFMA-bound ones
        self.num_streams = max(1, num_streams)
        self._streams: List[torch.cuda.Stream] = []
        self._stream_ws: List[Optional[torch.Tensor]] = []

    def detect(self, images: torch.Tensor, image_chunk: int = 16) -> Dict[str, torch.Tensor]:
        """images [n,H,W] (device, uint8 / float32) -> count [n], xy [n,K,2], scores [n,K], descriptors [n,K,256] with
        K = max_keypoints (top-k by response on the device, detection order)."""
        outs = [self.detector.forward(images[i : i + image_chunk], top_k=self.max_keypoints) for i in range(0, images.shape[0], image_chunk)]
        return {k: torch.cat([o[k] for o in outs], 0) for k in ("count", "xy", "scores", "descriptors")}

    def match(
        self, feats: Dict[str, torch.Tensor], pairs: Sequence[Tuple[int, int]], shapes: Sequence[Tuple[int, int]], counts: Optional[np.ndarray] = None,
        **matcher_kwargs,
    ) -> List[Dict[str, torch.Tensor]]:
        """Match `pairs` (indices into the feature table). `counts` = host copy of feats["count"] (fetched if None: the
        one host synchronisation of the pipeline, needed to size the ragged batch). Returns one dict per chunk with the
        engine outputs plus the chunk's pair list and per-pair keypoint counts."""
        if counts is None:
            counts = feats["count"].cpu().numpy()
        results = []
        full = bool((counts == feats["xy"].shape[1]).all())
        device = feats["xy"].device
        nstreams = min(self.num_streams, max(1, -(-len(pairs) // self.pair_chunk)))
        if nstreams > 1 and len(self._streams) < nstreams:
            self._streams = [torch.cuda.Stream(device) for _ in range(nstreams)]
            self._stream_ws = [None] * nstreams
        main = torch.cuda.current_stream(device)
        ready = torch.cuda.Event()
        ready.record(main)
        for ci, c0 in enumerate(range(0, len(pairs), self.pair_chunk)):
            chunk = list(pairs[c0 : c0 + self.pair_chunk])
            if nstreams > 1:
                si = ci % nstreams
                stream = self._streams[si]
                if ci < nstreams:
                    stream.wait_event(ready)  # features were produced on the caller's stream
                need = self.matcher.workspace_bytes([int(counts[i]) for i, _ in chunk], [int(counts[j]) for _, j in chunk])
                if self._stream_ws[si] is None or self._stream_ws[si].numel() < need:
                    self._stream_ws[si] = torch.empty(int(need * 1.1) + 256, dtype=torch.uint8, device=device)
                matcher_kwargs = dict(matcher_kwargs, workspace=self._stream_ws[si])
                ctx = torch.cuda.stream(stream)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                results.append(self._match_chunk(feats, chunk, shapes, counts, full, matcher_kwargs))
        if nstreams > 1:
            for stream in self._streams[:nstreams]:
                main.wait_stream(stream)  # the caller's stream sees every chunk's outputs
        return results

    def _match_chunk(self, feats, chunk, shapes, counts, full, matcher_kwargs):
        """One ragged multi-pair launch sequence on the current stream."""
        idx = torch.tensor([i for p in chunk for i in p], dtype=torch.long, device=feats["xy"].device)
        n0 = [int(counts[i]) for i, _ in chunk]
        n1 = [int(counts[j]) for _, j in chunk]
        hw = [[shapes[i][0], shapes[i][1], shapes[j][0], shapes[j][1]] for i, j in chunk]
        if full:  # every image has exactly K keypoints: plain gathers
            kp = feats["xy"].index_select(0, idx).reshape(-1, 2)
            sc = feats["scores"].index_select(0, idx).reshape(-1)
            de = feats["descriptors"].index_select(0, idx).reshape(-1, 256)
        else:
            kp = torch.cat([feats["xy"][i, : counts[i]] for i in idx.tolist()], 0)
            sc = torch.cat([feats["scores"][i, : counts[i]] for i in idx.tolist()], 0)
            de = torch.cat([feats["descriptors"][i, : counts[i]] for i in idx.tolist()], 0)
        if isinstance(self.matcher, SuperGlueEngine):
            out = self.matcher.match_batch(kp, sc, de, n0, n1, hw, **matcher_kwargs)
        else:
            out = self.matcher.match_batch(kp, de, n0, n1, hw, **matcher_kwargs)
        out["pairs"], out["n0"], out["n1"] = chunk, n0, n1
        return out

    @staticmethod
    def matches_to_numpy(results: List[Dict[str, torch.Tensor]], dtype=np.int64) -> Dict[Tuple[int, int], np.ndarray]:
        """(K,2) index arrays per pair, image-i1 keypoint order (the plugins' output format)."""
        out: Dict[Tuple[int, int], np.ndarray] = {}
        for res in results:
            m = res["matches"].cpu().numpy()
            row = 0
            for (i, j), a, b in zip(res["pairs"], res["n0"], res["n1"]):
                m0 = m[row : row + a]
                valid = m0 > -1
                out[(i, j)] = np.stack([np.flatnonzero(valid), m0[valid]], -1).astype(dtype)
                row += a + b
        return out
