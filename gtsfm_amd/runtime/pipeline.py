"""GPU-resident detect+match pipeline: images in HBM -> SuperPoint (batched) -> device top-k -> matcher over a list of
image pairs, with features never leaving the device between the two stages.

This is the throughput path of the front-end (BASELINE.json's metric) and the engine behind the batched correspondence
generator. The reference runs one Dask task per image and per pair with an H2D/D2H round trip and a pickle each
(``gtsfm/frontend/correspondence_generator/det_desc_correspondence_generator.py:65-87``); here a whole cluster's
pairs are matched from one resident feature table.
"""

from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from gtsfm_amd.runtime import lib as _lib
from gtsfm_amd.runtime.matcher_engine import LightGlueEngine, SuperGlueEngine
from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine


def _move_blocks(src: torch.Tensor, dst: torch.Tensor, src_index: Optional[torch.Tensor] = None, dst_index: Optional[torch.Tensor] = None) -> None:
    """Image blocks of a feature table moved by index on the current stream (``gtsfm_move_blocks_f32``): dst[b] = src[src_index[b]],
    or dst[dst_index[b]] = src[b]. src / dst: contiguous float32 [blocks, ...] with equal block shapes; indices int32 on the device."""
    n = int((src_index if src_index is not None else dst_index).numel())
    block = int(np.prod(src.shape[1:]))
    assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.float32 and tuple(src.shape[1:]) == tuple(dst.shape[1:])
    if block % 2:  # scores [images, K] with odd K: two blocks of one table never alias, a trailing float moves with torch
        (dst.index_copy_(0, dst_index.long(), src[:n]) if dst_index is not None else torch.index_select(src, 0, src_index.long(), out=dst[:n]))
        return
    lib = _lib.load()
    _lib.check(lib.gtsfm_move_blocks_f32(src.data_ptr(), _lib.ptr(src_index), dst.data_ptr(), _lib.ptr(dst_index), n, block,
                                         torch.cuda.current_stream(src.device).cuda_stream), "gtsfm_move_blocks_f32")


class _GraphedChunk:
    """The matcher's launch sequence for one chunk shape (P pairs of exactly K keypoints per image), captured once as a
    hipGraph and replayed per chunk: the sequence is static -- LightGlue's adaptive depth / width run on the device -- so
    only the gather of the chunk's features into the static input buffers stays eager."""

    def __init__(self, matcher, num_pairs: int, k: int, hw, matcher_kwargs: dict, stream: torch.cuda.Stream, feats, idx: torch.Tensor):
        dev = matcher.device
        self.matcher, self.kwargs = matcher, dict(matcher_kwargs)
        self.is_sg = isinstance(matcher, SuperGlueEngine)
        self.n, self.hw = [k] * num_pairs, hw
        self.kp = torch.zeros((2 * num_pairs, k, 2), dtype=torch.float32, device=dev)
        self.sc = torch.zeros((2 * num_pairs, k), dtype=torch.float32, device=dev)
        self.de = torch.zeros((2 * num_pairs, k, 256), dtype=torch.float32, device=dev)
        self.ws = torch.empty(matcher.workspace_bytes(self.n, self.n) + 256, dtype=torch.uint8, device=dev)
        # the captured copy node reads the pristine descriptor block at its address: pinned in the engine's cache for good
        with matcher.pin_descriptors():
            with torch.cuda.stream(stream):
                self._gather(feats, idx)
                self._run()  # eager warm-up on real features: descriptor cache, allocator
            stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=stream):
                self.out = self._run()

    def _run(self):
        kp, sc, de = self.kp.reshape(-1, 2), self.sc.reshape(-1), self.de.reshape(-1, 256)
        if self.is_sg:
            return self.matcher.match_batch(kp, sc, de, self.n, self.n, self.hw, workspace=self.ws, **self.kwargs)
        return self.matcher.match_batch(kp, de, self.n, self.n, self.hw, workspace=self.ws, **self.kwargs)

    def _gather(self, feats, idx: torch.Tensor):
        _move_blocks(feats["xy"], self.kp, src_index=idx)
        _move_blocks(feats["descriptors"], self.de, src_index=idx)
        if self.is_sg:
            _move_blocks(feats["scores"], self.sc, src_index=idx)

    def replay(self, feats, idx: torch.Tensor):
        self._gather(feats, idx)
        self.graph.replay()
        return {k: v.clone() for k, v in self.out.items() if isinstance(v, torch.Tensor) and not k.startswith("_")}


class FrontEndPipeline:
    def __init__(self, detector: SuperPointEngine, matcher, max_keypoints: int = 5000, pair_chunk: int = 32, num_streams: int = 2,
                 use_graphs: bool = False, share_first_layer: bool = True):
        self.detector = detector
        # The matchers' first layer begins with a block that sees ONE image (SuperGlue: keypoint encoder + first self layer;
        # LightGlue: first self block). When the pair list reuses images -- exhaustive or retrieval pairs do -- that block runs
        # once per image per match() call instead of once per pair side; same arithmetic per row, bit-identical matches.
        self.share_first_layer = share_first_layer
        self.matcher = matcher
        self.max_keypoints = max_keypoints
        self.pair_chunk = pair_chunk
        # pair chunks are independent: alternate them over HIP streams (each with its own workspace) so that one chunk's
        # small / tail-heavy kernels overlap the other's MFMA-bound ones
        self.num_streams = max(1, num_streams)
        self._streams: List[torch.cuda.Stream] = []
        self._stream_ws: List[Optional[torch.Tensor]] = []
        # full chunks (every image at the keypoint cap) replay a captured hipGraph of the matcher's launch sequence
        self.use_graphs = use_graphs
        self._graphs: Dict[tuple, _GraphedChunk] = {}

    def detect(self, images: torch.Tensor, image_chunk: int = 16) -> Dict[str, torch.Tensor]:
        """images [n,H,W] (device, uint8 / float32) -> count [n], xy [n,K,2], scores [n,K], descriptors [n,K,256] with
        K = max_keypoints (top-k by response on the device, detection order)."""
        outs = [self.detector.forward(images[i : i + image_chunk], top_k=self.max_keypoints) for i in range(0, images.shape[0], image_chunk)]
        return {k: torch.cat([o[k] for o in outs], 0) for k in ("count", "xy", "scores", "descriptors")}

    def detect_image_objects(self, imgs: Sequence, image_batch: int = 16) -> Dict[str, torch.Tensor]:
        """``detect`` for a list of ``Image`` objects as a correspondence generator receives them (``value_array`` HxW[xC] host arrays of
        possibly different sizes, optional ``mask``): equally-sized images are uploaded and detected in batches with the top-k taken on the
        device; RGB(A) uint8 batches are converted to gray on the device (``ImagePrep.rgb_to_gray``: the same 15-bit fixed-point formula as
        the host path); image masks (``Keypoints.filter_by_mask`` ahead of ``get_top_k``, gtsfm/frontend/detector_descriptor/
        superpoint.py:76-91) ride along as a uint8 batch and are applied on the device between the NMS and the keypoint extraction.
        Returns count [n] (int32), xy [n,K,2], scores [n,K], descriptors [n,K,256] in the order of ``imgs``; n = 0 gives empty tables."""
        from gtsfm_amd.common.image import rgb_to_gray_u8

        device, k, n = self.detector.device, self.max_keypoints, len(imgs)
        xy = torch.zeros((n, k, 2), dtype=torch.float32, device=device)
        sc = torch.zeros((n, k), dtype=torch.float32, device=device)
        de = torch.zeros((n, k, 256), dtype=torch.float32, device=device)
        cnt = torch.zeros((n,), dtype=torch.int32, device=device)
        by_shape: Dict[Tuple[int, int], List[int]] = {}
        for i, im in enumerate(imgs):
            by_shape.setdefault((int(im.height), int(im.width)), []).append(i)
        prep = None
        for (h, w), idxs in by_shape.items():
            for b0 in range(0, len(idxs), image_batch):
                sel = idxs[b0 : b0 + image_batch]
                arrays = [imgs[i].value_array for i in sel]
                if all(a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == arrays[0].shape[2] for a in arrays):
                    if prep is None:
                        from gtsfm_amd.runtime.image_prep import ImagePrep

                        prep = ImagePrep(device)
                    batch = prep.rgb_to_gray(torch.from_numpy(np.ascontiguousarray(np.stack(arrays))).to(device))  # one upload of the batch
                else:
                    gray = np.stack([np.ascontiguousarray(rgb_to_gray_u8(a)) for a in arrays])
                    if gray.dtype != np.uint8:
                        gray = gray.astype(np.float32) / 255.0
                    batch = torch.from_numpy(gray).to(device)
                masks = None
                if any(imgs[i].mask is not None for i in sel):
                    masks = torch.from_numpy(np.ascontiguousarray(np.stack(
                        [np.ones((h, w), dtype=np.uint8) if imgs[i].mask is None else (np.asarray(imgs[i].mask) == 1).astype(np.uint8) for i in sel]
                    ))).to(device)
                out = self.detector.forward(batch, top_k=k, valid_masks=masks)
                ii = torch.tensor(sel, dtype=torch.long, device=device)
                xy[ii], sc[ii], de[ii], cnt[ii] = out["xy"], out["scores"], out["descriptors"], out["count"].to(torch.int32)
        return {"count": cnt, "xy": xy, "scores": sc, "descriptors": de}

    def match(
        self, feats: Dict[str, torch.Tensor], pairs: Sequence[Tuple[int, int]], shapes: Sequence[Tuple[int, int]], counts: Optional[np.ndarray] = None,
        **matcher_kwargs,
    ) -> List[Dict[str, torch.Tensor]]:
        """Match `pairs` (indices into the feature table). `counts` = host copy of feats["count"] (fetched if None: the
        one host synchronisation of the pipeline, needed to size the ragged batch). Returns one dict per chunk with the
        engine outputs plus the chunk's pair list and per-pair keypoint counts."""
        self.last_shared_images = 0
        if len(pairs) == 0:  # a rank of a sharded scene may own no pair at all
            return []
        if counts is None:
            counts = feats["count"].cpu().numpy()
        results = []
        used = np.unique(np.asarray(pairs, dtype=np.int64).reshape(-1)) if len(pairs) else np.zeros(0, dtype=np.int64)
        full = bool((np.asarray(counts)[used] == feats["xy"].shape[1]).all())  # only the images the pairs touch (a gathered table may hold empty slots)
        device = feats["xy"].device
        nstreams = min(self.num_streams, max(1, -(-len(pairs) // self.pair_chunk)))
        side_streams = nstreams > 1 or self.use_graphs  # graphs are captured on (and replayed from) pipeline-owned streams
        if side_streams and len(self._streams) < nstreams:
            self._streams = [torch.cuda.Stream(device) for _ in range(nstreams)]
            self._stream_ws = [None] * nstreams
        main = torch.cuda.current_stream(device)
        self.last_shared_images = 0
        if self.share_first_layer and len(used) and 2 * len(pairs) > len(used) and getattr(self.matcher, "num_layers", 0) >= 1:
            feats = dict(feats, descriptors=self._first_layer_per_image(feats, used, shapes, counts, full))
            matcher_kwargs = dict(matcher_kwargs, first_layer_done=True)
            self.last_shared_images = int(len(used))
        ready = torch.cuda.Event()
        ready.record(main)
        chunks = [list(pairs[c0 : c0 + self.pair_chunk]) for c0 in range(0, len(pairs), self.pair_chunk)]
        if side_streams:
            # size every stream's workspace for the largest chunk it will see BEFORE anything is enqueued: a workspace
            # regrown between chunks would hand its old block back to the allocator while the side stream may still use it
            for si in range(nstreams):
                need = max(self.matcher.workspace_bytes([int(counts[i]) for i, _ in c], [int(counts[j]) for _, j in c]) for c in chunks[si::nstreams])
                if self._stream_ws[si] is None or self._stream_ws[si].numel() < need:
                    with torch.cuda.stream(self._streams[si]):
                        self._stream_ws[si] = torch.empty(int(need * 1.1) + 256, dtype=torch.uint8, device=device)
        k = feats["xy"].shape[1]
        for ci, chunk in enumerate(chunks):
            if side_streams:
                si = ci % nstreams
                stream = self._streams[si]
                if ci < nstreams:
                    stream.wait_event(ready)  # features were produced on the caller's stream
                kwargs = dict(matcher_kwargs, workspace=self._stream_ws[si])
                ctx = torch.cuda.stream(stream)
            else:
                si, stream, kwargs, ctx = 0, main, matcher_kwargs, contextlib.nullcontext()
            with ctx:
                hw0 = shapes[chunk[0][0]]
                uniform = all(shapes[i] == hw0 and shapes[j] == hw0 for i, j in chunk)
                if self.use_graphs and full and uniform and len(chunk) == self.pair_chunk:
                    results.append(self._replay_chunk(feats, chunk, k, hw0, matcher_kwargs, si, stream))
                else:
                    results.append(self._match_chunk(feats, chunk, shapes, counts, full, kwargs))
        if side_streams:
            for stream in self._streams[:nstreams]:
                main.wait_stream(stream)  # the caller's stream sees every chunk's outputs
        return results

    def _first_layer_per_image(self, feats, used, shapes, counts, full) -> torch.Tensor:
        """x after the matcher's per-image first block for the images in `used`, as a table shaped like feats["descriptors"]
        (rows of other images keep their descriptors; they are never read). On the caller's stream, 2 * pair_chunk images per
        launch sequence (the matcher workspace a pair chunk needs anyway)."""
        table = torch.empty_like(feats["descriptors"])  # only the rows of the images in `used` are written -- and only those are read
        is_sg = isinstance(self.matcher, SuperGlueEngine)
        step = 2 * self.pair_chunk
        for c0 in range(0, len(used), step):
            ids = [int(i) for i in used[c0 : c0 + step]]
            cnt = [int(counts[i]) for i in ids]
            if min(cnt) == 0:  # an empty keypoint set never reaches the matcher (empty-input early-out): drop it here too
                keep = [q for q, c in enumerate(cnt) if c > 0]
                ids, cnt = [ids[q] for q in keep], [cnt[q] for q in keep]
                if not ids:
                    continue
            idx = torch.tensor(ids, dtype=torch.int32, device=table.device)
            hw = [shapes[i] for i in ids]
            if full:
                k = feats["xy"].shape[1]
                kp = torch.empty((len(ids), k, 2), dtype=torch.float32, device=table.device)
                de = torch.empty((len(ids), k, 256), dtype=torch.float32, device=table.device)
                _move_blocks(feats["xy"], kp, src_index=idx)
                _move_blocks(feats["descriptors"], de, src_index=idx)
                kp, de = kp.reshape(-1, 2), de.reshape(-1, 256)
                sc = None
                if is_sg:
                    sc = torch.empty((len(ids), k), dtype=torch.float32, device=table.device)
                    _move_blocks(feats["scores"], sc, src_index=idx)
                    sc = sc.reshape(-1)
            else:
                kp = torch.cat([feats["xy"][i, :c] for i, c in zip(ids, cnt)], 0)
                de = torch.cat([feats["descriptors"][i, :c] for i, c in zip(ids, cnt)], 0)
                sc = torch.cat([feats["scores"][i, :c] for i, c in zip(ids, cnt)], 0) if is_sg else None
            x = self.matcher.prepare_images(kp, sc, de, cnt, hw) if is_sg else self.matcher.prepare_images(kp, de, cnt, hw)
            if full:
                _move_blocks(x.reshape(len(ids), -1, 256), table, dst_index=idx)
            else:
                row = 0
                for i, c in zip(ids, cnt):
                    table[i, :c] = x[row : row + c]
                    row += c
        return table

    def _replay_chunk(self, feats, chunk, k, hw0, matcher_kwargs, si, stream):
        # (the arithmetic switches are read by the C side when a graph is CAPTURED: a graph is only replayed under the values it was captured with)
        key = (si, len(chunk), k, tuple(hw0), tuple(sorted(matcher_kwargs.items())), os.environ.get("GTSFM_ATTENTION_MATH"), os.environ.get("GTSFM_GEMM_MATH"))
        idx = torch.tensor([i for p in chunk for i in p], dtype=torch.int32, device=feats["xy"].device)
        g = self._graphs.get(key)
        if g is None:
            hw = [[hw0[0], hw0[1], hw0[0], hw0[1]]] * len(chunk)
            g = self._graphs[key] = _GraphedChunk(self.matcher, len(chunk), k, hw, matcher_kwargs, stream, feats, idx)
        out = g.replay(feats, idx)
        out["pairs"], out["n0"], out["n1"] = chunk, [k] * len(chunk), [k] * len(chunk)
        return out

    def _match_chunk(self, feats, chunk, shapes, counts, full, matcher_kwargs):
        """One ragged multi-pair launch sequence on the current stream."""
        ids = [i for p in chunk for i in p]
        n0 = [int(counts[i]) for i, _ in chunk]
        n1 = [int(counts[j]) for _, j in chunk]
        hw = [[shapes[i][0], shapes[i][1], shapes[j][0], shapes[j][1]] for i, j in chunk]
        if full:  # every image has exactly K keypoints: block gathers by image index
            idx = torch.tensor(ids, dtype=torch.int32, device=feats["xy"].device)
            k = feats["xy"].shape[1]
            kp = torch.empty((len(ids), k, 2), dtype=torch.float32, device=idx.device)
            sc = torch.empty((len(ids), k), dtype=torch.float32, device=idx.device)
            de = torch.empty((len(ids), k, 256), dtype=torch.float32, device=idx.device)
            _move_blocks(feats["xy"], kp, src_index=idx)
            _move_blocks(feats["scores"], sc, src_index=idx)
            _move_blocks(feats["descriptors"], de, src_index=idx)
            kp, sc, de = kp.reshape(-1, 2), sc.reshape(-1), de.reshape(-1, 256)
        else:
            kp = torch.cat([feats["xy"][i, : counts[i]] for i in ids], 0)
            sc = torch.cat([feats["scores"][i, : counts[i]] for i in ids], 0)
            de = torch.cat([feats["descriptors"][i, : counts[i]] for i in ids], 0)
        if isinstance(self.matcher, SuperGlueEngine):
            out = self.matcher.match_batch(kp, sc, de, n0, n1, hw, **matcher_kwargs)
        else:
            out = self.matcher.match_batch(kp, de, n0, n1, hw, **matcher_kwargs)
        out["pairs"], out["n0"], out["n1"] = chunk, n0, n1
        return out

    def verify(self, feats: Dict[str, torch.Tensor], results: List[Dict[str, torch.Tensor]], intrinsics: np.ndarray, threshold_px: float,
               engine=None, use_intrinsics: bool = True) -> List[Dict[str, torch.Tensor]]:
        """The verifier stage on the matcher's device output (``two_view_estimator.py:391-397`` per pair in the reference):
        all chunks of ``match()`` together in one compaction launch and one RANSAC launch, nothing copied to the host. ``intrinsics``
        [num_images, 4] = (fx, fy, cx, cy) per row of the feature table. Pair (i, j) draws its minimal samples from the seed
        ``i << 32 | j``, so a pair's result does not depend on how the pair list was chunked or sharded. Returns a list (one
        entry per 65535 pairs) of E / R / t / mask / stats plus match_idx, match_off (host), match_count and the pair list."""
        if engine is None:
            if getattr(self, "_verifier", None) is None:
                from gtsfm_amd.runtime.verifier_engine import VerifierEngine

                self._verifier = VerifierEngine(feats["xy"].device)
            engine = self._verifier
        k = feats["xy"].shape[1]
        table = feats["xy"].reshape(-1, 2)
        intrinsics = np.asarray(intrinsics, dtype=np.float64)
        out = []
        group: List[Dict[str, torch.Tensor]] = []
        for res in list(results) + [None]:  # all chunks of the step in one launch pair (the ABI takes up to 65535 pairs per call)
            if res is not None and sum(len(r["pairs"]) for r in group) + len(res["pairs"]) <= 65535:
                group.append(res)
                continue
            if group:
                pairs = [p for r in group for p in r["pairs"]]
                n0 = [a for r in group for a in r["n0"]]
                n1 = [b for r in group for b in r["n1"]]
                matches = group[0]["matches"] if len(group) == 1 else torch.cat([r["matches"] for r in group])
                rows = np.concatenate([[0], np.cumsum([a + b for a, b in zip(n0, n1)])])[:-1]
                idx, match_off, count = engine.compact_matches(matches, rows.tolist(), n0)
                intr = np.concatenate([intrinsics[[i for i, _ in pairs]], intrinsics[[j for _, j in pairs]]], axis=1)
                ver = engine.verify_batch(table, [i * k for i, _ in pairs], [j * k for _, j in pairs], idx, match_off, intr, threshold_px,
                                          seeds=[(i << 32) | j for i, j in pairs], match_count=count, use_intrinsics=use_intrinsics)
                ver.update(match_idx=idx, match_off=match_off, match_count=count, pairs=pairs)
                out.append(ver)
            group = [res] if res is not None else []
        return out

    @staticmethod
    def verified_to_numpy(verified: List[Dict[str, torch.Tensor]]) -> Dict[Tuple[int, int], Dict[str, np.ndarray]]:
        """Per pair the verifier plugins' return values: R, t (None without a model), v_corr_idxs (N,2), inlier_ratio."""
        out: Dict[Tuple[int, int], Dict[str, np.ndarray]] = {}
        for ver in verified:
            idx, mask, count = ver["match_idx"].cpu().numpy(), ver["mask"].cpu().numpy().astype(bool), ver["match_count"].cpu().numpy()
            rot, trans, stats = ver["R"].cpu().numpy(), ver["t"].cpu().numpy(), ver["stats"].cpu().numpy()
            for p, pair in enumerate(ver["pairs"]):
                lo = ver["match_off"][p]
                m, keep = idx[lo : lo + count[p]], mask[lo : lo + count[p]]
                ok = stats[p, 0] > 0
                out[pair] = {"R": rot[p] if ok else None, "t": trans[p] if ok else None, "v_corr_idxs": m[keep].astype(np.int64),
                             "inlier_ratio": float(keep.mean()) if ok else 0.0, "putative": m.astype(np.int64), "hypotheses": int(stats[p, 1])}
        return out

    @staticmethod
    def matches_to_numpy(results: List[Dict[str, torch.Tensor]], dtype=np.int64) -> Dict[Tuple[int, int], np.ndarray]:
        """(K,2) index arrays per pair, image-i1 keypoint order (the plugins' output format)."""
        out: Dict[Tuple[int, int], np.ndarray] = {}
        from gtsfm_amd.runtime.matcher_engine import check_split_arithmetic_range, split_arithmetic_has_fp16_range

        for res in results:
            m = res["matches"].cpu().numpy()
            if "mscores" in res and split_arithmetic_has_fp16_range():  # (the scores travel to the host only under the opt-in f16x2 switches)
                check_split_arithmetic_range(res["mscores"].cpu().numpy())
            row = 0
            for (i, j), a, b in zip(res["pairs"], res["n0"], res["n1"]):
                m0 = m[row : row + a]
                valid = m0 > -1
                out[(i, j)] = np.stack([np.flatnonzero(valid), m0[valid]], -1).astype(dtype)
                row += a + b
        return out
