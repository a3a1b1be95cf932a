"""Host side of the matcher HIP paths (SuperGlue, LightGlue): checkpoint -> logical matrices -> packed blob, ragged
batch descriptors, workspace management, kernel enqueue. PyTorch provides device memory and streams only.

Weight preparation (done once, on the host, in float64 then rounded to fp32):

SuperGlue (``thirdparty/SuperGluePretrainedNetwork/models/superglue.py``)
  * eval-mode ``BatchNorm1d`` layers (superglue.py:57-58) are folded into the preceding ``Conv1d``:
    ``W' = W * g / sqrt(var + eps)``, ``b' = (b - mean) * g / sqrt(var + eps) + beta``
  * the attention head layout ``view(b, 64, 4, N)`` (superglue.py:104: channel = d*4 + h, head is the FAST axis) is
    made head-major (channel = h*64 + d) by permuting the rows of the three projections and the columns of ``merge``
  * the q/k/v projections are fused into one 256 -> 768 matrix
  * ``attn.merge`` is folded into ``mlp.0`` (``_fold_projection``)
  blob entry order: kenc (32,3) (64,32) (128,64) (256,128) (256,256); per GNN layer (768,256) (512,512) (256,512);
  final_proj (256,256).

LightGlue (upstream ``cvg/LightGlue`` module names)
  * ``Wqkv`` rows are regrouped from ``h*192 + d*3 + j`` to ``j*256 + h*64 + d`` (q | k | v, head-major)
  * the cross block's ``to_qk`` and ``to_v`` are fused into one 256 -> 512 matrix
  * ``out_proj`` / ``to_out`` are folded into the following ``ffn.0`` (``_fold_projection``)
  blob entry order: posenc.Wr (raw 64); per layer: Wqkv (768,256), self ffn.0 (512,512), self LN gamma (raw 512), beta
  (raw 512), self ffn.3 (256,512), cross to_qk|to_v (512,256), cross ffn.0, LN gamma, beta, ffn.3, log_assignment
  final_proj (256,256), matchability w (raw 256), [token_confidence w (raw 256) for all but the last layer]; the two
  scalar biases per layer travel separately (host floats).
"""

from __future__ import annotations

import contextlib
import ctypes as C
import hashlib
import os
import queue
import threading
from collections import OrderedDict
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from gtsfm_amd.runtime import lib as _lib
from gtsfm_amd.runtime.superpoint_engine import require_gpu

BN_EPS = 1e-5
HEAD_PERM = np.array([(c % 64) * 4 + c // 64 for c in range(256)])  # new channel h*64+d <- old channel d*4+h

Entry = Tuple[int, np.ndarray, Optional[np.ndarray]]  # (kind, W or raw vector, bias)


def _f64(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().double().numpy()


def pack_blob(entries: Sequence[Entry]) -> np.ndarray:
    lib = _lib.load()
    count = len(entries)
    kinds = np.array([e[0] for e in entries], dtype=np.int32)
    ws = [np.ascontiguousarray(e[1], dtype=np.float32) for e in entries]
    bs = [None if e[2] is None else np.ascontiguousarray(e[2], dtype=np.float32) for e in entries]
    n = np.array([w.shape[0] for w in ws], dtype=np.int32)
    k = np.array([w.shape[1] if (e[0] == 0) else 0 for w, e in zip(ws, entries)], dtype=np.int32)
    total = lib.gtsfm_blob_floats(count, kinds.ctypes.data, n.ctypes.data, k.ctypes.data)
    out = np.empty(total, dtype=np.float32)
    wp = (C.c_void_p * count)(*[w.ctypes.data for w in ws])
    bp = (C.c_void_p * count)(*[None if b is None else b.ctypes.data for b in bs])
    _lib.check(lib.gtsfm_pack_blob(count, kinds.ctypes.data, n.ctypes.data, k.ctypes.data, wp, bp, out.ctypes.data), "gtsfm_pack_blob")
    return out


# ------------------------------------------------------------------------------------------------------------------
# SuperGlue
# ------------------------------------------------------------------------------------------------------------------


def superglue_num_layers(sd: Mapping[str, torch.Tensor]) -> int:
    return 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("gnn.layers."))


def _fold_bn(sd, conv: str, bn: Optional[str]) -> Tuple[np.ndarray, np.ndarray]:
    w = _f64(sd[f"{conv}.weight"])[:, :, 0]
    b = _f64(sd[f"{conv}.bias"])
    if bn is not None:
        scale = _f64(sd[f"{bn}.weight"]) / np.sqrt(_f64(sd[f"{bn}.running_var"]) + BN_EPS)
        w = w * scale[:, None]
        b = (b - _f64(sd[f"{bn}.running_mean"])) * scale + _f64(sd[f"{bn}.bias"])
    return w, b


def _fold_projection(w0: np.ndarray, b0: np.ndarray, wp: np.ndarray, bp: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """The attention output projection (SuperGlue ``attn.merge``, LightGlue ``out_proj`` / ``to_out``) feeds only the
    second half of the following ``cat([x, message])`` linear layer (superglue.py:117-118), so it folds into that layer
    exactly: ``W0 [x ; Wp a + bp] + b0 = [W0x | W0m Wp] [x ; a] + (b0 + W0m bp)``. One 256 x 256 GEMM per block and the
    round trip of the message through HBM disappear; the product is formed in float64 (closer to the exact result than
    the reference's two fp32 steps)."""
    half = w0.shape[1] // 2
    assert wp.shape == (half, half)
    return np.concatenate([w0[:, :half], w0[:, half:] @ wp], 1), b0 + w0[:, half:] @ bp


def superglue_entries(sd: Mapping[str, torch.Tensor]) -> List[Entry]:
    entries: List[Entry] = []
    for i in range(5):  # kenc.encoder: conv at 0,3,6,9,12; BN at 1,4,7,10
        w, b = _fold_bn(sd, f"kenc.encoder.{3 * i}", f"kenc.encoder.{3 * i + 1}" if i < 4 else None)
        entries.append((0, w, b))
    for l in range(superglue_num_layers(sd)):
        p = f"gnn.layers.{l}"
        ws, bs = [], []
        for j in range(3):
            w, b = _fold_bn(sd, f"{p}.attn.proj.{j}", None)
            ws.append(w[HEAD_PERM]), bs.append(b[HEAD_PERM])
        entries.append((0, np.concatenate(ws, 0), np.concatenate(bs, 0)))
        wm, bm = _fold_bn(sd, f"{p}.attn.merge", None)
        entries.append((0, *_fold_projection(*_fold_bn(sd, f"{p}.mlp.0", f"{p}.mlp.1"), wm[:, HEAD_PERM], bm)))
        entries.append((0, *_fold_bn(sd, f"{p}.mlp.3", None)))
    entries.append((0, *_fold_bn(sd, "final_proj", None)))
    return entries


class _ImageEntry:
    """One image's device-resident matcher inputs (the per-call plugin path): its keypoints (and scores) as uploaded and the output
    of ``prepare_images`` -- the matcher block that sees ONE image. ``ready`` orders consumers on other lanes' streams behind it."""

    __slots__ = ("kpts", "scores", "x", "ready", "stream", "digest")

    def __init__(self, kpts, scores, x, ready, stream, digest=None):
        self.kpts, self.scores, self.x, self.ready, self.stream, self.digest = kpts, scores, x, ready, stream, digest


_WARNED_SLOW_DIGEST = False


def _full_digest(arrays: Sequence[np.ndarray]) -> bytes:
    """Checksum over EVERY byte of an image's host arrays. The per-call path's image cache is validated against it on every hit, while the GPU already
    works on the pair (``_MatcherBase._entries_still_valid``): a cached image can never be served for changed arrays.

    With the ``xxhash`` package (an OPTIONAL dependency: INTEGRATION.md section 3): xxh3-128, 0.4 ms per 5000 x 256 float32 descriptor matrix -- far
    inside the 9 - 11 ms the GPU spends on the pair, so the check costs a call nothing. Without it: ``hashlib.blake2b`` (16-byte digest, standard library,
    ~5 ms per matrix): two images per hit then cost about what the GPU does, the per-call path becomes host-bound, and a warning says so once. Either way
    the digest is 128 bits (the round-5 fallback was 64 bits of crc32 + adler32)."""
    try:
        import xxhash
    except ImportError:
        xxhash = None
    if xxhash is not None:
        h = xxhash.xxh3_128()
        for a in arrays:
            h.update(np.ascontiguousarray(a))
        return h.digest()
    import hashlib
    import warnings

    global _WARNED_SLOW_DIGEST
    if not _WARNED_SLOW_DIGEST:
        _WARNED_SLOW_DIGEST = True
        warnings.warn("gtsfm_amd: the `xxhash` package is not installed; the per-call image cache validates its hits with hashlib.blake2b instead "
                      "(~5 ms per 5000 x 256 descriptor matrix: about the GPU time of a pair). Install xxhash, or set GTSFM_PLUGIN_IMAGE_CACHE=0.",
                      RuntimeWarning, stacklevel=2)
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        h.update(np.ascontiguousarray(a))
    return h.digest()


def split_arithmetic_has_fp16_range() -> bool:
    """True under the opt-in f16x2 switches (read from the environment like the library reads them, per call)."""
    import os

    return any((os.environ.get(k) or "")[:2] == "f1" for k in ("GTSFM_ATTENTION_MATH", "GTSFM_GEMM_MATH"))


def check_split_arithmetic_range(mscores: np.ndarray) -> None:
    """The opt-in f16x2 arithmetic (GTSFM_ATTENTION_MATH / GTSFM_GEMM_MATH = f16x2; gtsfm_amd/csrc/f16x2.h) carries every operand in two fp16 pieces:
    a value beyond +-65504 becomes inf in its leading piece and NaN in every score it reaches. That must not pass for "no matches": raise."""
    if split_arithmetic_has_fp16_range() and np.isnan(np.asarray(mscores)).any():
        raise FloatingPointError("f16x2 arithmetic: an operand of the matcher left fp16's range (|x| > 65504) and the scores are NaN; "
                                 "run this input with GTSFM_ATTENTION_MATH / GTSFM_GEMM_MATH unset (exact fp32) or =bf16x3")


class _MatcherBase:
    # what a lane shares with the engine it was made from: read-only after construction (the weight blob lives on the device)
    _SHARED_ATTRS: Tuple[str, ...] = ("device", "_lib", "weights", "num_layers", "desc_cache_capacity", "max_lanes", "pair_streams")

    def __init__(self, device: Optional[torch.device]):
        self.device = require_gpu(device)
        self._lib = _lib.load()
        self._init_host_state()

    def _init_host_state(self) -> None:
        self.desc_cache_capacity = 64
        # match_pair() re-uses an engine's staging buffers and workspace, so one thread at a time owns them. Calls from several
        # threads of one worker process (GTSfM's ``--threads_per_worker``, gtsfm/runner.py:155,436) do not queue behind each other,
        # though: up to ``max_lanes`` of them run side by side, each on a LANE = an engine of its own (own workspace, staging
        # buffers, descriptor cache and HIP stream) that shares this engine's weights. One pair leaves a fifth of the chip idle
        # between and inside its ~230 launches (320 attention workgroups on 512 slots at N = 5000); a second and third launch sequence
        # fill it: 86 -> 102 -> 107 pairs/s at the cap, 346 -> 429 -> 450 at N = 2048 (tools/bench_plugin_threads.py). Lanes are created
        # when a call finds every existing one busy; a single-threaded caller only ever has the first. Memory: a lane at the
        # 5000-keypoint cap holds ~0.5 GB of workspace + 10 MB of pinned / device staging; ``release_lanes()``
        # returns it.
        self.max_lanes = max(1, int(os.environ.get("GTSFM_PLUGIN_LANES", "3")))
        # launch sequences of ONE pair (LightGlue: 2 = one per image, gtsfm_lg_forward_streams). Built in round 5, bit-identical, and OFF by
        # default: the two sequences must meet four times per layer, and a cross-stream event wait costs ~40 us on this runtime -- measured
        # 10.85 vs 10.97 ms per pair at the 5000-keypoint cap (the overlap wins 1.5 ms and the 36 waits take 1.4 back) and 4.28 vs 2.89 ms at
        # N = 2048. Two caller threads, which share nothing, do gain: 103 vs 91 pairs/s.
        self.pair_streams = max(1, int(os.environ.get("GTSFM_PAIR_STREAMS", "1")))
        self._init_call_state()
        self._lanes: list = [self]
        self._free_lanes: "queue.LifoQueue" = queue.LifoQueue()
        self._free_lanes.put(self)
        self._lanes_lock = threading.Lock()
        self._root: "_MatcherBase" = self
        # per-call path: device-resident per-image inputs, shared by the lanes (see _image_entries)
        self.image_cache_capacity = max(0, int(os.environ.get("GTSFM_PLUGIN_IMAGE_CACHE", "64")))
        self._image_cache: "OrderedDict[tuple, _ImageEntry]" = OrderedDict()
        self._image_cache_lock = threading.Lock()
        self.image_cache_hits = self.image_cache_misses = self.image_cache_stale = 0

    def _init_call_state(self) -> None:
        """Everything a call writes; a lane has its own."""
        self._workspace: Optional[torch.Tensor] = None
        # pristine descriptor blocks per batch shape, least recently used first. A block a captured hipGraph copies from must
        # never be freed (the graph's copy node holds its ADDRESS): blocks touched inside ``pin_descriptors()`` are exempt
        # from eviction for the life of the engine.
        self._desc_cache: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()
        self._desc_pinned: set = set()
        self._pinning = False
        self._staging: Optional[dict] = None
        self._lane_stream: Optional[torch.cuda.Stream] = None
        self._last_lookup: tuple = ((), [])  # what the last _image_entries call of this lane found (read by _entries_still_valid)
        self._side_stream: Optional[torch.cuda.Stream] = None  # second launch sequence of a single pair (LightGlueEngine.match_batch)

    def _sibling(self) -> "_MatcherBase":
        """An engine sharing this one's weights (read-only on the device) and nothing a call writes: built from the explicit list
        of shared attributes, so a per-call attribute added later cannot be shared between threads by accident."""
        other = object.__new__(type(self))
        for name in self._SHARED_ATTRS:
            setattr(other, name, getattr(self, name))
        other._init_call_state()
        other._lanes, other._free_lanes, other._lanes_lock = [other], None, None  # lanes are handed out by the engine they were made from
        other._root = self
        return other

    @contextlib.contextmanager
    def _lane(self):
        """An engine no other thread is using -- this one or a sibling -- with its own stream current."""
        try:
            eng = self._free_lanes.get_nowait()
        except queue.Empty:
            eng = None
            with self._lanes_lock:
                if len(self._lanes) < self.max_lanes:
                    eng = self._sibling()
                    self._lanes.append(eng)
            if eng is None:
                eng = self._free_lanes.get()
        try:
            if eng._lane_stream is None:
                eng._lane_stream = torch.cuda.Stream(self.device)
            with torch.cuda.stream(eng._lane_stream):
                yield eng
        finally:
            self._free_lanes.put(eng)

    def release_lanes(self) -> None:
        """Give back what the per-call path holds between calls: the idle sibling lanes (workspace, staging buffers, descriptor-block
        caches), this engine's own workspace and staging buffers, and the per-image cache. Lanes serving a call at this moment survive;
        descriptor blocks pinned by captured hipGraphs stay (the graphs hold their addresses). The next call rebuilds what it needs."""
        with self._lanes_lock:
            idle = []
            while True:
                try:
                    idle.append(self._free_lanes.get_nowait())
                except queue.Empty:
                    break
            for eng in idle:
                if eng is not self:
                    self._lanes.remove(eng)
            if self in idle:
                self._workspace = self._staging = None
                for key in [k for k in self._desc_cache if k not in self._desc_pinned]:
                    del self._desc_cache[key]
                self._free_lanes.put(self)
        with self._image_cache_lock:
            self._image_cache.clear()

    # -- per-call path: every image once ------------------------------------------------------------------------------------------

    @staticmethod
    def _image_key(arrays: Sequence[np.ndarray], shape: Tuple[int, int]) -> tuple:
        """LOOKUP key of one image's host arrays as a ``match()`` call hands them over: address, shape, dtype and strides of every array
        plus a hash over a sample of rows (the first and last 10 and 16 evenly spaced ones) -- the reference's own MatcherCacher keys
        a pair on the first 10 rows alone (gtsfm/frontend/cacher/matcher_cacher.py:24,46-80). Cheap enough for the critical path of a
        call (0.05 ms); it only FINDS a candidate entry. Whether the candidate still describes the arrays is decided by the checksum over
        all of their bytes (``_full_digest``), verified on every hit while the GPU runs the pair (``_entries_still_valid``): an array
        overwritten in place that keeps all sampled rows is detected there and the pair is redone from the arrays (round 5; until then
        such a caller was served the old features). GTSFM_PLUGIN_IMAGE_CACHE=0 turns the cache off."""
        n = len(arrays[0])
        rows = np.unique(np.concatenate([np.arange(min(10, n)), np.arange(max(0, n - 10), n), np.linspace(0, n - 1, 16).astype(np.int64)])) if n else np.zeros(0, np.int64)
        h = hashlib.blake2b(digest_size=16)
        ident = []
        for a in arrays:
            a = np.asarray(a)
            ident.append((a.__array_interface__["data"][0], a.shape, a.dtype.str, a.strides))
            h.update(np.ascontiguousarray(a[rows]).tobytes())
        # (the cached entry holds the first block's OUTPUT: it belongs to the arithmetic it was computed under)
        import os

        return (tuple(ident), (int(shape[0]), int(shape[1])), h.digest(), os.environ.get("GTSFM_ATTENTION_MATH"), os.environ.get("GTSFM_GEMM_MATH"))

    def _image_entries(self, images: Sequence[Tuple[Sequence[np.ndarray], Tuple[int, int]]]) -> List[_ImageEntry]:
        """Device-resident inputs of the images of one ``match_pair`` call, on the CURRENT lane (self) and stream. The reference's
        wrappers upload both images' features for every pair (superglue_matcher.py:75-102, lightglue_matcher.py:75-100) and the
        matcher's first block then runs per pair side; an image that appears in k pairs is uploaded and taken through that block ONCE
        here (``prepare_images``: bit-identical to the per-pair form, tests/test_matchers_gpu.py) and found in the cache k - 1 times.
        The cache belongs to the engine the lanes were made from; an entry made on another lane's stream is ordered by its event."""
        root = self._root
        keys = [self._image_key(arrs, shape) for arrs, shape in images]
        stream = torch.cuda.current_stream(self.device)
        with root._image_cache_lock:
            found = [root._image_cache.get(k) for k in keys]
            for k, e in zip(keys, found):
                if e is not None:
                    root._image_cache.move_to_end(k)
        miss = [i for i, e in enumerate(found) if e is None and keys[i] not in keys[:i]]
        if miss:
            staged = self._stage_sets([images[i][0] for i in miss])
            counts = [len(images[i][0][0]) for i in miss]
            x = self._prepare_staged(staged, counts, [images[i][1] for i in miss])
            ready = torch.cuda.Event()
            off = 0
            for i, c in zip(miss, counts):
                # every tensor an entry holds is its own allocation (evicting one image frees its memory while its siblings of the same
                # call live on) and is enqueued on the producer stream BEFORE the event consumers on other lanes wait for
                kp = staged[0][off : off + c].clone()
                sc = staged[1][off : off + c].reshape(-1).clone() if len(staged) == 3 else None
                found[i] = _ImageEntry(kp, sc, x[off : off + c].clone() if len(miss) > 1 else x, ready, stream)
                off += c
            ready.record(stream)  # after the clones: a consumer lane that waits for `ready` sees kpts / scores / x complete
            for i in miss:  # over every byte, on the host while the device runs the upload and the per-image block; before the entry is published
                found[i].digest = _full_digest(images[i][0])
            with root._image_cache_lock:
                for i in miss:
                    root._image_cache[keys[i]] = found[i]
                while len(root._image_cache) > root.image_cache_capacity:
                    root._image_cache.popitem(last=False)  # consumers hold their own references; the allocator waits for recorded streams
                root.image_cache_misses += len(miss)
                root.image_cache_hits += len(images) - len(miss)
        else:
            with root._image_cache_lock:
                root.image_cache_hits += len(images)
        for i, e in enumerate(found):
            if e is None:  # the same image twice in one call
                found[i] = e = found[keys.index(keys[i])]
            if e.stream != stream:
                stream.wait_event(e.ready)
                for t in (e.kpts, e.scores, e.x):
                    if t is not None:
                        t.record_stream(stream)
        self._last_lookup = (keys, [i not in miss and keys[i] not in [keys[m] for m in miss] for i in range(len(images))])
        return found

    def _entries_still_valid(self, images: Sequence[Tuple[Sequence[np.ndarray], Tuple[int, int]]], entries: Sequence[_ImageEntry]) -> bool:
        """Called between enqueueing a pair's launches and fetching its result: the entries that came from the cache (hits of the cheap
        lookup key) are checked against the checksum over ALL bytes of the arrays they were found for. The GPU is busy with the pair
        meanwhile (11 ms at the 5000-keypoint cap against 0.8 ms for two images), so a hit costs the call nothing. An entry that fails is
        dropped from the cache and False is returned: the caller discards the enqueued result and runs the pair again from the arrays."""
        root = self._root
        keys, was_hit = self._last_lookup
        ok = True
        for (arrs, _), e, key, hit in zip(images, entries, keys, was_hit):
            if hit and e.digest != _full_digest(arrs):
                ok = False
                with root._image_cache_lock:
                    if root._image_cache.get(key) is e:
                        del root._image_cache[key]
                    root.image_cache_stale += 1
        return ok

    def _stage_pair(self, arrays0: Sequence[np.ndarray], arrays1: Sequence[np.ndarray]) -> List[torch.Tensor]:
        """Host arrays of the two images of ONE pair -> device tensors [n0 + n1, ...], one per array (see ``_stage_sets``)."""
        return self._stage_sets([arrays0, arrays1])

    def _stage_sets(self, sets: Sequence[Sequence[np.ndarray]]) -> List[torch.Tensor]:
        """Host arrays of one or more images -> device tensors [sum of rows, ...], one per array kind: converted to float32 straight
        into page-locked staging buffers that live as long as the engine (grown geometrically), then one asynchronous copy each
        into equally persistent device buffers. The per-call plugin API (``match(...)`` with numpy in / numpy out, once per pair of
        the Dask graph) pays no allocation, no pageable-memory bounce and no intermediate concatenation this way."""
        rows_of = [len(arrs[0]) for arrs in sets]
        t = int(sum(rows_of))
        widths = [int(np.prod(np.shape(a)[1:])) for a in sets[0]]
        st = self._staging
        if st is None or st["rows"] < t or st["widths"] != widths:
            rows = max(t, int(1.5 * st["rows"]) if st is not None and st["widths"] == widths else 0, 256)
            st = self._staging = {
                "rows": rows, "widths": widths,
                "host": [torch.empty((rows, w), dtype=torch.float32, pin_memory=True) for w in widths],
                "dev": [torch.empty((rows, w), dtype=torch.float32, device=self.device) for w in widths],
                "done": torch.cuda.Event(),
            }
            st["views"] = [h.numpy() for h in st["host"]]
        else:
            st["done"].synchronize()  # the previous call's copies have left the staging buffers
        out = []
        for j, (view, host, dev, w) in enumerate(zip(st["views"], st["host"], st["dev"], widths)):
            off = 0
            for arrs, n in zip(sets, rows_of):
                view[off : off + n] = np.asarray(arrs[j]).reshape(n, w)  # numpy casts (float64 / integer inputs of the reference's own tests) while copying
                off += n
            dev[:t].copy_(host[:t], non_blocking=True)
            out.append(dev[:t])
        st["done"].record(torch.cuda.current_stream(self.device))
        return out

    def _get_workspace(self, nbytes: int) -> torch.Tensor:
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = None
            self._workspace = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self._workspace

    def _build_desc(self, superglue: bool, n0: np.ndarray, n1: np.ndarray, hw: np.ndarray) -> torch.Tensor:
        """A fresh device copy of the batch descriptor block (LightGlue updates its counts in place). The pristine block of
        a batch shape is built and uploaded once and cached: chunks of a scene share shapes, so the steady state is one
        device-to-device copy per chunk instead of a host build + H2D."""
        key = (superglue, n0.tobytes(), n1.tobytes(), hw.tobytes())
        pristine = self._desc_cache.get(key)
        if pristine is None:
            p = len(n0)
            host = np.empty(self._lib.gtsfm_match_desc_ints(int(superglue), p, n0.ctypes.data, n1.ctypes.data), dtype=np.int32)
            _lib.check(
                self._lib.gtsfm_match_build_desc(int(superglue), p, n0.ctypes.data, n1.ctypes.data, hw.ctypes.data, host.ctypes.data),
                "gtsfm_match_build_desc",
            )
            for old in [k for k in self._desc_cache if k not in self._desc_pinned][: max(0, len(self._desc_cache) + 1 - self.desc_cache_capacity)]:
                del self._desc_cache[old]  # oldest unpinned shapes go; the caching allocator keeps a block alive until its queued readers ran
            pristine = self._desc_cache[key] = torch.from_numpy(host).to(self.device)
        else:
            self._desc_cache.move_to_end(key)
        if self._pinning:
            self._desc_pinned.add(key)
        return pristine.clone()

    @contextlib.contextmanager
    def pin_descriptors(self):
        """Descriptor blocks built or reused inside this context stay cached for good (hipGraph capture: pipeline._GraphedChunk)."""
        self._pinning = True
        try:
            yield
        finally:
            self._pinning = False

    def workspace_bytes(self, n0: Sequence[int], n1: Sequence[int]) -> int:
        a0 = np.ascontiguousarray(n0, dtype=np.int32)
        a1 = np.ascontiguousarray(n1, dtype=np.int32)
        fn = self._lib.gtsfm_sg_workspace_bytes if isinstance(self, SuperGlueEngine) else self._lib.gtsfm_lg_workspace_bytes
        return int(fn(len(a0), a0.ctypes.data, a1.ctypes.data))


class SuperGlueEngine(_MatcherBase):
    """Device-resident SuperGlue (superglue.py:228-283) for ragged batches of pairs."""

    _SHARED_ATTRS = _MatcherBase._SHARED_ATTRS + ("bin_score",)

    def _prepare_staged(self, staged, counts, shapes):
        return self.prepare_images(staged[0], staged[1].reshape(-1), staged[2], counts, shapes)

    def __init__(self, state_dict: Mapping[str, torch.Tensor], device: Optional[torch.device] = None):
        super().__init__(device)
        self.num_layers = superglue_num_layers(state_dict)
        self.bin_score = float(state_dict["bin_score"])
        self.weights = torch.from_numpy(pack_blob(superglue_entries(state_dict))).to(self.device)

    def packed_meta(self) -> dict:
        """What besides the packed weight blob a rank needs to run this model (``from_packed``): small host values."""
        return {"kind": "superglue", "num_layers": int(self.num_layers), "bin_score": float(self.bin_score)}

    @classmethod
    def from_packed(cls, weights: torch.Tensor, meta: Mapping) -> "SuperGlueEngine":
        """Build from an already-packed device blob + ``packed_meta()`` (received by an RCCL broadcast: the rank never reads a checkpoint)."""
        assert meta["kind"] == "superglue"
        self = cls.__new__(cls)
        _MatcherBase.__init__(self, weights.device)
        self.num_layers, self.bin_score, self.weights = int(meta["num_layers"]), float(meta["bin_score"]), weights
        return self

    def match_batch(
        self,
        kpts: torch.Tensor,
        scores: torch.Tensor,
        desc: torch.Tensor,
        n0: Sequence[int],
        n1: Sequence[int],
        hw: Sequence[Sequence[int]],
        sinkhorn_iterations: int = 20,
        match_threshold: float = 0.2,
        return_ot: bool = False,
        workspace: Optional[torch.Tensor] = None,
        first_layer_done: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """Token-major device inputs concatenated as pair0/img0, pair0/img1, pair1/img0, ...: kpts [T,2], scores [T],
        desc [T,256]; n0/n1 per-pair keypoint counts (all > 0); hw per pair (H0, W0, H1, W1). ``first_layer_done``: desc holds
        the output of ``prepare_images`` (keypoint encoder + first self layer already applied per image).
        Returns matches [T] int32 (matches0 for img0 rows, matches1 for img1 rows) and mscores [T]."""
        n0 = np.ascontiguousarray(n0, dtype=np.int32)
        n1 = np.ascontiguousarray(n1, dtype=np.int32)
        hw = np.ascontiguousarray(hw, dtype=np.int32).reshape(-1, 4)
        p = len(n0)
        t = int(n0.sum() + n1.sum())
        assert kpts.shape == (t, 2) and scores.shape == (t,) and desc.shape == (t, 256)
        assert kpts.is_contiguous() and scores.is_contiguous() and desc.is_contiguous()
        assert kpts.dtype == scores.dtype == desc.dtype == torch.float32
        dsc = self._build_desc(True, n0, n1, hw)
        need = self._lib.gtsfm_sg_workspace_bytes(p, n0.ctypes.data, n1.ctypes.data)
        ws = workspace if workspace is not None and workspace.numel() >= need else self._get_workspace(need)
        matches = torch.empty(t, dtype=torch.int32, device=self.device)
        mscores = torch.empty(t, dtype=torch.float32, device=self.device)
        ot = None
        if return_ot:
            zf = sum((int(a) + 1) * ((int(b) + 1 + 3) // 4 * 4) for a, b in zip(n0, n1))
            ot = torch.empty(zf, dtype=torch.float32, device=self.device)
        rc = self._lib.gtsfm_sg_forward_phase(
            self.weights.data_ptr(), self.num_layers, self.bin_score, p, n0.ctypes.data, n1.ctypes.data, dsc.data_ptr(),
            kpts.data_ptr(), scores.data_ptr(), desc.data_ptr(), int(sinkhorn_iterations), float(match_threshold),
            ws.data_ptr(), ws.numel(), matches.data_ptr(), mscores.data_ptr(), _lib.ptr(ot), 2 if first_layer_done else 0, None,
            torch.cuda.current_stream(self.device).cuda_stream,
        )
        _lib.check(rc, "gtsfm_sg_forward_phase")
        out = {"matches": matches, "mscores": mscores, "_desc": dsc}
        if return_ot:
            out["ot"] = ot
        return out

    def prepare_images(self, kpts: torch.Tensor, scores: torch.Tensor, desc: torch.Tensor, counts: Sequence[int], shapes: Sequence[Sequence[int]],
                       workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The part of SuperGlue that sees ONE image -- keypoint encoder and the first (self) GNN layer (superglue.py:243-248 and
        the first iteration of :126-137) -- for the keypoint sets of `counts` images concatenated in kpts / scores / desc.
        Returns x [T,256] in the same row order; feed it to ``match_batch(..., first_layer_done=True)`` in place of desc. An image
        in k pairs then pays for this once instead of k times, with bit-identical matches."""
        if self.num_layers < 1:
            raise ValueError("prepare_images needs at least one GNN layer")
        counts = [int(c) for c in counts]
        shapes = [tuple(int(v) for v in s) for s in shapes]
        t = sum(counts)
        assert kpts.shape == (t, 2) and scores.shape == (t,) and desc.shape == (t, 256) and min(counts) > 0
        kpts, scores, desc = kpts.contiguous(), scores.contiguous(), desc.contiguous()
        if len(counts) % 2:  # the ABI takes keypoint sets two at a time: an odd last image leaves the second slot empty (count 0: no rows, no work)
            counts, shapes = counts + [0], shapes + [shapes[-1]]
        n0 = np.ascontiguousarray(counts[0::2], dtype=np.int32)
        n1 = np.ascontiguousarray(counts[1::2], dtype=np.int32)
        hw = np.ascontiguousarray([[*shapes[2 * q], *shapes[2 * q + 1]] for q in range(len(n0))], dtype=np.int32)
        dsc = self._build_desc(True, n0, n1, hw)
        need = self._lib.gtsfm_sg_workspace_bytes(len(n0), n0.ctypes.data, n1.ctypes.data)
        ws = workspace if workspace is not None and workspace.numel() >= need else self._get_workspace(need)
        x = torch.empty((t, 256), dtype=torch.float32, device=self.device)
        rc = self._lib.gtsfm_sg_forward_phase(
            self.weights.data_ptr(), self.num_layers, self.bin_score, len(n0), n0.ctypes.data, n1.ctypes.data, dsc.data_ptr(),
            kpts.data_ptr(), scores.data_ptr(), desc.data_ptr(), 0, 0.0, ws.data_ptr(), ws.numel(),
            None, None, None, 1, x.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream,
        )
        _lib.check(rc, "gtsfm_sg_forward_phase")
        return x

    def match_pair(
        self, k0: np.ndarray, s0: np.ndarray, d0: np.ndarray, k1: np.ndarray, s1: np.ndarray, d1: np.ndarray,
        shape0: Tuple[int, int], shape1: Tuple[int, int], sinkhorn_iterations: int = 20, match_threshold: float = 0.2,
        return_ot: bool = False,
    ) -> Dict[str, np.ndarray]:
        """One pair from host arrays -> matches0 [n0] int64, matches1 [n1] int64, matching_scores0/1 (superglue.py
        output dict), with the empty-input early-out of superglue.py:233-240."""
        n0, n1 = len(k0), len(k1)
        if n0 == 0 or n1 == 0:
            return {
                "matches0": np.full(n0, -1, dtype=np.int32), "matches1": np.full(n1, -1, dtype=np.int32),
                "matching_scores0": np.zeros(n0, dtype=np.float32), "matching_scores1": np.zeros(n1, dtype=np.float32),
            }
        with self._lane() as eng:
            hw = [[shape0[0], shape0[1], shape1[0], shape1[1]]]
            if self.image_cache_capacity > 0 and self.num_layers >= 1:
                images = [((k0, s0, d0), shape0), ((k1, s1, d1), shape1)]
                for attempt in range(2):  # a second round only when a cached image turned out to be stale (its entry is gone by then)
                    e0, e1 = eng._image_entries(images)
                    out = eng.match_batch(torch.cat([e0.kpts, e1.kpts]), torch.cat([e0.scores, e1.scores]), torch.cat([e0.x, e1.x]), [n0], [n1], hw,
                                          sinkhorn_iterations, match_threshold, return_ot, first_layer_done=True)
                    if eng._entries_still_valid(images, (e0, e1)):
                        break
            else:
                kp, sc, de = eng._stage_pair((k0, s0, d0), (k1, s1, d1))
                out = eng.match_batch(kp, sc.reshape(-1), de, [n0], [n1], hw, sinkhorn_iterations, match_threshold, return_ot)
            m = out["matches"].cpu().numpy().astype(np.int64)
            ms = out["mscores"].cpu().numpy()
            check_split_arithmetic_range(ms)
            ot = out["ot"].cpu().numpy() if return_ot else None
        res = {"matches0": m[:n0], "matches1": m[n0:], "matching_scores0": ms[:n0], "matching_scores1": ms[n0:]}
        if return_ot:
            ld = (n1 + 1 + 3) // 4 * 4
            res["ot"] = ot.reshape(n0 + 1, ld)[:, : n1 + 1]
        return res


# ------------------------------------------------------------------------------------------------------------------
# LightGlue
# ------------------------------------------------------------------------------------------------------------------

LIGHTGLUE_DEPTH_CONFIDENCE = 0.95  # upstream default_conf
LIGHTGLUE_WIDTH_CONFIDENCE = 0.99
LIGHTGLUE_FILTER_THRESHOLD = 0.1
# upstream pruning_keypoint_thresholds = {"cpu": -1, "mps": -1, "cuda": 1024, "flash": 1536}; GTSfM runs the matcher on
# "cuda" with torch >= 2 (scaled_dot_product_attention available) -> "flash"
LIGHTGLUE_PRUNING_THRESHOLD = 1536
NO_PRUNING = 2**31 - 1


def normalize_lightglue_state_dict(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept both key layouts of upstream ``cvg/LightGlue`` checkpoints. The published ``superpoint_lightglue.pth``
    (v0.1_arxiv, the file the reference's ``LightGlue(features="superpoint")`` downloads,
    gtsfm/frontend/matcher/lightglue_matcher.py:41) stores ``self_attn.{i}.*`` / ``cross_attn.{i}.*``; upstream renames
    them to ``transformers.{i}.self_attn.*`` / ``transformers.{i}.cross_attn.*`` when loading ("rename old state dict
    entries"). The same renames are applied here; already-renamed dicts pass through unchanged."""
    out: Dict[str, torch.Tensor] = {}
    for key, value in sd.items():
        parts = key.split(".")
        if len(parts) > 2 and parts[0] in ("self_attn", "cross_attn") and parts[1].isdigit():
            key = ".".join(["transformers", parts[1], parts[0]] + parts[2:])
        out[key] = value
    if not any(k.startswith("transformers.") for k in out):
        raise KeyError(
            "LightGlue state_dict holds neither 'transformers.{i}.*' nor 'self_attn.{i}.*' / 'cross_attn.{i}.*' entries "
            f"(first keys: {sorted(sd)[:4]})"
        )
    return out


def lightglue_num_layers(sd: Mapping[str, torch.Tensor]) -> int:
    return 1 + max(int(k.split(".")[1]) for k in normalize_lightglue_state_dict(sd) if k.startswith("transformers."))


def lightglue_entries(sd: Mapping[str, torch.Tensor]) -> Tuple[List[Entry], np.ndarray, np.ndarray]:
    sd = normalize_lightglue_state_dict(sd)
    n_layers = lightglue_num_layers(sd)
    qkv_perm = np.array([(r % 256 // 64) * 192 + (r % 64) * 3 + r // 256 for r in range(768)])  # new j*256+h*64+d <- old h*192+d*3+j
    entries: List[Entry] = [(1, _f64(sd["posenc.Wr.weight"]).reshape(-1), None)]
    match_bias, conf_bias = [], []
    for l in range(n_layers):
        p = f"transformers.{l}"
        entries.append((0, _f64(sd[f"{p}.self_attn.Wqkv.weight"])[qkv_perm], _f64(sd[f"{p}.self_attn.Wqkv.bias"])[qkv_perm]))

        def ffn(blk, proj):
            wp, bp = _f64(sd[f"{p}.{blk}.{proj}.weight"]), _f64(sd[f"{p}.{blk}.{proj}.bias"])
            entries.append((0, *_fold_projection(_f64(sd[f"{p}.{blk}.ffn.0.weight"]), _f64(sd[f"{p}.{blk}.ffn.0.bias"]), wp, bp)))
            entries.append((1, _f64(sd[f"{p}.{blk}.ffn.1.weight"]), None))
            entries.append((1, _f64(sd[f"{p}.{blk}.ffn.1.bias"]), None))
            entries.append((0, _f64(sd[f"{p}.{blk}.ffn.3.weight"]), _f64(sd[f"{p}.{blk}.ffn.3.bias"])))

        ffn("self_attn", "out_proj")
        entries.append((
            0,
            np.concatenate([_f64(sd[f"{p}.cross_attn.to_qk.weight"]), _f64(sd[f"{p}.cross_attn.to_v.weight"])], 0),
            np.concatenate([_f64(sd[f"{p}.cross_attn.to_qk.bias"]), _f64(sd[f"{p}.cross_attn.to_v.bias"])], 0),
        ))
        ffn("cross_attn", "to_out")
        a = f"log_assignment.{l}"
        entries.append((0, _f64(sd[f"{a}.final_proj.weight"]), _f64(sd[f"{a}.final_proj.bias"])))
        entries.append((1, _f64(sd[f"{a}.matchability.weight"]).reshape(-1), None))
        match_bias.append(float(sd[f"{a}.matchability.bias"].reshape(-1)[0]))
        if l < n_layers - 1:
            entries.append((1, _f64(sd[f"token_confidence.{l}.token.0.weight"]).reshape(-1), None))
            conf_bias.append(float(sd[f"token_confidence.{l}.token.0.bias"].reshape(-1)[0]))
    return entries, np.array(match_bias, dtype=np.float32), np.array(conf_bias + [0.0], dtype=np.float32)


class LightGlueEngine(_MatcherBase):
    """Device-resident LightGlue(features="superpoint") for ragged batches of pairs; adaptive depth and width are
    evaluated on the device (no host synchronisation inside the layer loop)."""

    _SHARED_ATTRS = _MatcherBase._SHARED_ATTRS + ("match_bias", "conf_bias")

    def _prepare_staged(self, staged, counts, shapes):
        return self.prepare_images(staged[0], staged[1], counts, shapes)

    def __init__(self, state_dict: Mapping[str, torch.Tensor], device: Optional[torch.device] = None):
        super().__init__(device)
        self.num_layers = lightglue_num_layers(state_dict)
        entries, self.match_bias, self.conf_bias = lightglue_entries(state_dict)
        self.weights = torch.from_numpy(pack_blob(entries)).to(self.device)

    def packed_meta(self) -> dict:
        """What besides the packed weight blob a rank needs to run this model (``from_packed``): small host values."""
        return {"kind": "lightglue", "num_layers": int(self.num_layers), "match_bias": [float(v) for v in self.match_bias],
                "conf_bias": [float(v) for v in self.conf_bias]}

    @classmethod
    def from_packed(cls, weights: torch.Tensor, meta: Mapping) -> "LightGlueEngine":
        """Build from an already-packed device blob + ``packed_meta()`` (received by an RCCL broadcast: the rank never reads a checkpoint)."""
        assert meta["kind"] == "lightglue"
        self = cls.__new__(cls)
        _MatcherBase.__init__(self, weights.device)
        self.num_layers, self.weights = int(meta["num_layers"]), weights
        self.match_bias = np.array(meta["match_bias"], dtype=np.float32)
        self.conf_bias = np.array(meta["conf_bias"], dtype=np.float32)
        return self

    def match_batch(
        self,
        kpts: torch.Tensor,
        desc: torch.Tensor,
        n0: Sequence[int],
        n1: Sequence[int],
        hw: Sequence[Sequence[int]],
        depth_confidence: float = LIGHTGLUE_DEPTH_CONFIDENCE,
        width_confidence: float = LIGHTGLUE_WIDTH_CONFIDENCE,
        filter_threshold: float = LIGHTGLUE_FILTER_THRESHOLD,
        pruning_threshold: Optional[int] = LIGHTGLUE_PRUNING_THRESHOLD,
        return_sim: bool = False,
        workspace: Optional[torch.Tensor] = None,
        first_layer_done: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """kpts [T,2], desc [T,256] concatenated as pair0/img0, pair0/img1, ...; ``first_layer_done``: desc holds the output of
        ``prepare_images`` (first self block already applied per image). Returns matches [T] int32 in ORIGINAL
        keypoint indices, mscores [T], stop [P] (layers run), kept [2P] (keypoints alive at the final assignment)."""
        n0 = np.ascontiguousarray(n0, dtype=np.int32)
        n1 = np.ascontiguousarray(n1, dtype=np.int32)
        hw = np.ascontiguousarray(hw, dtype=np.int32).reshape(-1, 4)
        p = len(n0)
        t = int(n0.sum() + n1.sum())
        assert kpts.shape == (t, 2) and desc.shape == (t, 256) and kpts.is_contiguous() and desc.is_contiguous()
        assert kpts.dtype == desc.dtype == torch.float32
        dsc = self._build_desc(False, n0, n1, hw)
        need = self._lib.gtsfm_lg_workspace_bytes(p, n0.ctypes.data, n1.ctypes.data)
        ws = workspace if workspace is not None and workspace.numel() >= need else self._get_workspace(need)
        matches = torch.empty(t, dtype=torch.int32, device=self.device)
        mscores = torch.empty(t, dtype=torch.float32, device=self.device)
        sim = None
        if return_sim:
            sim = torch.zeros(sum(int(a) * ((int(b) + 3) // 4 * 4) for a, b in zip(n0, n1)), dtype=torch.float32, device=self.device)
        # ONE pair (the per-call plugin path) can run its launch sequence as two -- image 0's per-image work on the current stream, image 1's
        # on a side stream of this lane, joined inside the call (gtsfm_lg_forward_streams; bit-identical; opt-in: GTSFM_PAIR_STREAMS=2, see
        # _init_host_state for why it is not the default). A batch fills the chip by itself, and a sequence being captured into a hipGraph
        # stays on its one stream.
        side = None
        if p == 1 and self.pair_streams > 1 and not torch.cuda.is_current_stream_capturing():
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(self.device)
            side = self._side_stream.cuda_stream
        rc = self._lib.gtsfm_lg_forward_streams(
            self.weights.data_ptr(), self.num_layers, self.match_bias.ctypes.data, self.conf_bias.ctypes.data, p,
            n0.ctypes.data, n1.ctypes.data, dsc.data_ptr(), kpts.data_ptr(), desc.data_ptr(), float(depth_confidence),
            float(width_confidence), float(filter_threshold), NO_PRUNING if pruning_threshold is None else int(pruning_threshold),
            ws.data_ptr(), ws.numel(), matches.data_ptr(), mscores.data_ptr(), _lib.ptr(sim), 2 if first_layer_done else 0, None,
            torch.cuda.current_stream(self.device).cuda_stream, side,
        )
        _lib.check(rc, "gtsfm_lg_forward_streams")
        out = {
            "matches": matches, "mscores": mscores,
            "kept": dsc[2 * p : 4 * p],        # final counts section of the descriptor block
            "stop": dsc[10 * p : 11 * p] + 1,  # stop-layer section (+1 = number of layers run)
        }
        if return_sim:
            out["sim"] = sim
        return out

    def prepare_images(self, kpts: torch.Tensor, desc: torch.Tensor, counts: Sequence[int], shapes: Sequence[Sequence[int]],
                       workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The part of LightGlue that sees ONE image: the first layer's self block (rotary self-attention + FFN) on the keypoint
        sets of `counts` images concatenated in kpts / desc. Returns x [T,256]; feed it to ``match_batch(..., first_layer_done=True)``
        in place of desc. Bit-identical matches; an image in k pairs pays for the block once instead of k times."""
        counts = [int(c) for c in counts]
        shapes = [tuple(int(v) for v in s) for s in shapes]
        t = sum(counts)
        assert kpts.shape == (t, 2) and desc.shape == (t, 256) and min(counts) > 0
        kpts, desc = kpts.contiguous(), desc.contiguous()
        if len(counts) % 2:  # the ABI takes keypoint sets two at a time: an odd last image leaves the second slot empty (count 0: no rows, no work)
            counts, shapes = counts + [0], shapes + [shapes[-1]]
        n0 = np.ascontiguousarray(counts[0::2], dtype=np.int32)
        n1 = np.ascontiguousarray(counts[1::2], dtype=np.int32)
        hw = np.ascontiguousarray([[*shapes[2 * q], *shapes[2 * q + 1]] for q in range(len(n0))], dtype=np.int32)
        dsc = self._build_desc(False, n0, n1, hw)
        need = self._lib.gtsfm_lg_workspace_bytes(len(n0), n0.ctypes.data, n1.ctypes.data)
        ws = workspace if workspace is not None and workspace.numel() >= need else self._get_workspace(need)
        x = torch.empty((t, 256), dtype=torch.float32, device=self.device)
        rc = self._lib.gtsfm_lg_forward_phase(
            self.weights.data_ptr(), self.num_layers, self.match_bias.ctypes.data, self.conf_bias.ctypes.data, len(n0),
            n0.ctypes.data, n1.ctypes.data, dsc.data_ptr(), kpts.data_ptr(), desc.data_ptr(), 0.0, 0.0, 0.0, NO_PRUNING,
            ws.data_ptr(), ws.numel(), None, None, None, 1, x.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream,
        )
        _lib.check(rc, "gtsfm_lg_forward_phase")
        return x

    def match_pair(
        self, k0: np.ndarray, d0: np.ndarray, k1: np.ndarray, d1: np.ndarray, shape0: Tuple[int, int], shape1: Tuple[int, int],
        **kwargs,
    ) -> Dict[str, np.ndarray]:
        """One pair from host arrays -> upstream's output dict subset: matches (K,2) int64, scores (K,), matches0/1,
        matching_scores0/1, stop. Empty inputs give empty matches (upstream breaks out of the layer loop)."""
        n0, n1 = len(k0), len(k1)
        if n0 == 0 or n1 == 0:
            return {
                "matches": np.zeros((0, 2), dtype=np.int64), "scores": np.zeros((0,), dtype=np.float32),
                "matches0": np.full(n0, -1, dtype=np.int64), "matches1": np.full(n1, -1, dtype=np.int64),
                "matching_scores0": np.zeros(n0, dtype=np.float32), "matching_scores1": np.zeros(n1, dtype=np.float32), "stop": 1,
            }
        with self._lane() as eng:
            hw = [[shape0[0], shape0[1], shape1[0], shape1[1]]]
            if self.image_cache_capacity > 0 and not kwargs.get("first_layer_done", False):
                images = [((k0, d0), shape0), ((k1, d1), shape1)]
                for attempt in range(2):  # a second round only when a cached image turned out to be stale (its entry is gone by then)
                    e0, e1 = eng._image_entries(images)
                    out = eng.match_batch(torch.cat([e0.kpts, e1.kpts]), torch.cat([e0.x, e1.x]), [n0], [n1], hw, **dict(kwargs, first_layer_done=True))
                    if eng._entries_still_valid(images, (e0, e1)):
                        break
            else:
                kp, de = eng._stage_pair((k0, d0), (k1, d1))
                out = eng.match_batch(kp, de, [n0], [n1], hw, **kwargs)
            m = out["matches"].cpu().numpy().astype(np.int64)
            ms = out["mscores"].cpu().numpy()
            check_split_arithmetic_range(ms)
            out = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in out.items() if k in ("stop", "kept", "sim")}
        m0 = m[:n0]
        valid = m0 > -1
        res = {
            "matches": np.stack([np.flatnonzero(valid), m0[valid]], -1).astype(np.int64), "scores": ms[:n0][valid],
            "matches0": m0, "matches1": m[n0:], "matching_scores0": ms[:n0], "matching_scores1": ms[n0:],
            "stop": int(out["stop"][0]), "kept": out["kept"].cpu().numpy(),
        }
        if "sim" in out:
            res["sim"] = out["sim"].cpu().numpy().reshape(n0, (n1 + 3) // 4 * 4)[:, :n1]
        return res
