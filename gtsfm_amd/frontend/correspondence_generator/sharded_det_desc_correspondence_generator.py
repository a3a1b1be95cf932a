"""Correspondence generator for ONE scene sharded over the GPUs of a node (BASELINE config 4).

Same contract as ``DetDescCorrespondenceGenerator`` (``gtsfm/frontend/correspondence_generator/
det_desc_correspondence_generator.py:33-87``): ``generate_correspondences(client, images, visibility_graph)`` returns
``(List[Keypoints], Dict[(i1, i2) -> (K, 2) index array])`` and is constructed from the same two plugin objects. The reference
reaches N GPUs by starting one Dask worker per device (``gtsfm/runner.py:359-414``, ``_create_local_cuda_cluster``) and submitting
one task per image and per pair, each with its own upload, download and pickle. Here it is one process per GPU under
``torch.distributed`` (backend "nccl" = RCCL over xGMI), and per scene:

1. rank 0 packs the SuperPoint / matcher checkpoints once and **broadcasts the packed blobs** (5 + 48 MB); the other ranks never
   read a checkpoint (``parallel.broadcast_packed_weights``);
2. rank r **detects** the images ``{i : i mod R = r}`` (batched by shape, top-k on the device);
3. **one exchange step**: every detected image goes to exactly the ranks whose pairs touch it -- one ``all_to_all_single`` per
   feature array, straight into each rank's own feature table (``parallel.ScenePlan`` / ``exchange_feature_rows``: with the 2-D
   block-cyclic pair ownership a rank holds about ``n / rows + n / cols`` of the ``n`` images, 71 of 101 for BASELINE config 4 on
   8 ranks, instead of all of them);
4. rank r **matches** its share of the visibility graph from its resident table (``FrontEndPipeline.match``: ragged multi-pair
   launches, the matcher's per-image first block once per image);
5. the ragged **match lists are gathered** (``parallel.gather_matches``) and the keypoint lists all-gathered.

Three ways to run it, chosen at the first call:

* **joined** -- ``torch.distributed`` is already initialised in the calling process (a job launched with ``torchrun`` /
  ``python -m torch.distributed.run``, e.g. ``bench.py --mode scene``): ``generate_correspondences`` is a COLLECTIVE call, made on
  every rank with the same arguments; every rank returns the whole result.
* **launcher** -- no process group, ``num_gpus > 1`` (default: every visible GPU): the first call starts one rank process per GPU
  (``multiprocessing`` "spawn", rendezvous on 127.0.0.1), ships them this object (plugins pickle before any device state exists,
  as the reference's do for ``client.scatter``) and, per call, each rank's share of the images; rank 0 hands the result back. The
  rank processes live until ``close()`` / garbage collection / interpreter exit.
* **single** -- one GPU: the same steps in the calling process without a process group (equals ``BatchedDetDescCorrespondenceGenerator``).

Visible differences to the reference's generator, by design: keypoints come back in detection (row-major) order -- index pairs refer
to the returned lists, so downstream code is unaffected --, and ``client`` is only used to resolve image futures.
"""

from __future__ import annotations

import atexit
import os
import socket
import traceback
import weakref
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from gtsfm_amd import parallel
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.correspondence_generator.batched_det_desc_correspondence_generator import (
    BatchedDetDescCorrespondenceGenerator,
    match_output_convention,
)
from gtsfm_amd.frontend.correspondence_generator.correspondence_generator_base import CorrespondenceGeneratorBase
from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
from gtsfm_amd.frontend.matcher.matcher_base import MatcherBase
from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher


class SceneResult(NamedTuple):
    """What one rank holds after ``run_scene``: the plan, its feature table (rows = ``plan.table_images``), the matcher's device results
    for its own pairs (table-row indices), and the gathered (K, 2) int64 arrays of EVERY pair of the scene (global image indices)."""

    plan: parallel.ScenePlan
    table: Dict[str, Any]
    results: List[Dict[str, Any]]
    matches: Dict[Tuple[int, int], np.ndarray]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class ShardedDetDescCorrespondenceGenerator(CorrespondenceGeneratorBase):
    """SuperPoint + {SuperGlue, LightGlue} correspondences of one scene, sharded over the GPUs of a node via RCCL."""

    def __init__(
        self,
        matcher: MatcherBase,
        detector_descriptor: SuperPointDetectorDescriptor,
        num_gpus: Optional[int] = None,
        image_batch: int = 16,
        pair_batch: int = 0,
        backend: str = "nccl",
        pipeline_factory: Optional[Callable[..., Any]] = None,
        pipeline_options: Optional[Dict[str, Any]] = None,
        always_launch: bool = False,
        collective_timeout_s: float = 600.0,
    ) -> None:
        """``num_gpus``: ranks to start in launcher mode (None = every visible GPU); ignored when the caller is already part of a process
        group. ``pair_batch``: pairs per matcher launch sequence (0 = by keypoint count: 32 up to 2560 keypoints, 16 at the 5000 cap).
        ``pipeline_options``: extra ``FrontEndPipeline`` arguments (``num_streams``, ``use_graphs``, ``share_first_layer``).
        ``pipeline_factory(generator, device) -> pipeline``: replaces the HIP pipeline (CPU plumbing tests with stand-in kernels; must be
        a picklable top-level callable in launcher mode); the plugins' models are then never built and no weights are broadcast.
        ``always_launch``: start rank processes even for one GPU (keeps the calling process free of device state; also how the launcher path
        is exercised on a one-GPU box). ``collective_timeout_s``: process-group timeout of launcher mode -- a rank that never arrives at a
        collective fails its peers after this long instead of blocking them for good (the reference's counterpart is the connect-retry
        loop of ``gtsfm/runner.py:313-357``)."""
        if pipeline_factory is None:
            if not isinstance(detector_descriptor, SuperPointDetectorDescriptor):
                raise TypeError("ShardedDetDescCorrespondenceGenerator needs gtsfm_amd's SuperPointDetectorDescriptor")
            if not isinstance(matcher, (SuperGlueMatcher, LightGlueMatcher)):
                raise TypeError("ShardedDetDescCorrespondenceGenerator needs gtsfm_amd's SuperGlueMatcher or LightGlueMatcher")
        self._detector_descriptor = detector_descriptor
        self._matcher = matcher
        self._num_gpus = num_gpus
        self._image_batch = int(image_batch)
        self._pair_batch = int(pair_batch)
        self._backend = backend
        self._pipeline_factory = pipeline_factory
        self._pipeline_options = dict(pipeline_options or {})
        self._always_launch = bool(always_launch)
        self._collective_timeout_s = float(collective_timeout_s)
        self._pipe = None  # per process: FrontEndPipeline over this rank's engines
        self._pool = None  # launcher mode: the rank processes
        self.last_scene: Optional[SceneResult] = None

    def __repr__(self) -> str:
        return f"""
        ShardedDetDescCorrespondenceGenerator (one rank per GPU, RCCL):
           {self._detector_descriptor}
           {self._matcher}
        """

    def __getstate__(self):
        # device state and child processes belong to one process; a pickled copy (a rank of the launcher, a Dask worker) builds its own
        return {**self.__dict__, "_pipe": None, "_pool": None, "last_scene": None}

    # -- engines: rank 0 packs, RCCL broadcasts -------------------------------------------------------------------------------------

    def _device(self):
        import torch

        if self._pipeline_factory is not None and not torch.cuda.is_available():
            return torch.device("cpu")
        if not torch.cuda.is_available():
            raise RuntimeError("gtsfm_amd requires an AMD GPU visible to PyTorch-ROCm; there is no CPU fallback.")
        return torch.device("cuda", torch.cuda.current_device())

    def attach(self):
        """This rank's pipeline, built on first use. Inside a process group: rank 0 loads and packs both checkpoints, the packed blobs and
        the few host scalars that go with them are broadcast (RCCL over xGMI), the other ranks build their engines from the blobs."""
        if self._pipe is not None:
            return self._pipe
        device = self._device()
        if self._pipeline_factory is not None:
            self._pipe = self._pipeline_factory(self, device)
            return self._pipe
        import torch

        from gtsfm_amd.runtime import lib as L
        from gtsfm_amd.runtime import matcher_engine as ME
        from gtsfm_amd.runtime.pipeline import FrontEndPipeline
        from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

        rank, world = parallel.world_info()
        det, mat = self._detector_descriptor, self._matcher
        if world == 1:
            det._ensure_model_loaded()
            mat._ensure_model_loaded()
        else:
            import torch.distributed as dist

            meta: List[Any] = [None]
            if rank == 0:
                det._ensure_model_loaded()
                mat._ensure_model_loaded()
                meta = [{"matcher": mat._model.packed_meta(), "matcher_floats": int(mat._model.weights.numel())}]
            dist.broadcast_object_list(meta, src=0)
            sp_blob = parallel.broadcast_packed_weights(det._model.weights if rank == 0 else None, int(L.load().gtsfm_sp_packed_weight_floats()), device)
            mt_blob = parallel.broadcast_packed_weights(mat._model.weights if rank == 0 else None, meta[0]["matcher_floats"], device)
            if rank != 0:
                det._model = SuperPointEngine.from_packed(sp_blob)
                engine = ME.SuperGlueEngine if meta[0]["matcher"]["kind"] == "superglue" else ME.LightGlueEngine
                mat._model = engine.from_packed(mt_blob, meta[0]["matcher"])
        k = det.max_keypoints
        chunk = self._pair_batch if self._pair_batch > 0 else (32 if k <= 2560 else 16)
        self._pipe = FrontEndPipeline(det._model, mat._model, max_keypoints=k, pair_chunk=chunk, **self._pipeline_options)
        torch.cuda.synchronize(device)
        return self._pipe

    # -- the sharded scene on resident data (bench.py --mode scene times exactly this) ------------------------------------------------

    def run_scene(self, local_feats: Dict[str, Any], num_images: int, shapes: Sequence[Tuple[int, int]], pairs: Sequence[Tuple[int, int]],
                  **matcher_kwargs) -> SceneResult:
        """Steps 3 - 5 for detections that are already on this rank's device: ``local_feats`` row s = image ``partition_images(...)[s]``
        (count [s], xy [s,K,2], scores [s,K], descriptors [s,K,256]); ``shapes[i]`` = (H, W) of image i; ``pairs`` = the scene's edges
        (global image indices), identical on every rank. Collective. Pairs with an empty keypoint set never reach the matcher
        (superglue.py:233-240 early-out) and come back as (0, 2) arrays."""
        import torch

        pipe = self.attach()
        rank, world = parallel.world_info()
        plan = parallel.ScenePlan(num_images, pairs, rank, world)
        device = local_feats["xy"].device
        table = parallel.exchange_feature_rows(plan, local_feats, gather_rows=self._gather_rows if device.type == "cuda" else None)
        counts = table["count"].cpu().numpy().astype(np.int64)
        table_shapes = [tuple(shapes[i]) for i in plan.table_images]
        todo = [(p, q) for p, q in zip(plan.my_pairs, plan.local_pairs) if counts[q[0]] > 0 and counts[q[1]] > 0]
        results, local, failure = [], {}, None
        try:
            results = pipe.match(table, [q for _, q in todo], table_shapes, counts=counts, **matcher_kwargs) if todo else []
            local = pipe.matches_to_numpy(results) if results else {}
        except Exception as exc:  # noqa: BLE001 - every rank must learn of it BEFORE the gather below, or the healthy ones block in it
            failure = exc
        parallel.agree_or_raise(failure, "matching")
        mine = {p: local[q] for p, q in todo}
        for p in plan.my_pairs:
            mine.setdefault(p, np.zeros((0, 2), dtype=np.int64))
        gathered = parallel.gather_matches(mine, device if device.type == "cuda" else torch.device("cpu"))
        if sorted(gathered) != sorted(set(plan.pairs)):
            raise RuntimeError("sharded scene: the gathered match lists do not cover the scene's pairs exactly once")
        self.last_scene = SceneResult(plan, table, results, gathered)
        return self.last_scene

    @staticmethod
    def _gather_rows(t, index):
        """Send-buffer assembly on the device: image blocks moved by index with the block-move kernel (float32 tables); counts with ATen."""
        import torch

        from gtsfm_amd.runtime.pipeline import _move_blocks

        if t.dtype != torch.float32 or t.dim() < 2 or index.numel() == 0:
            return torch.index_select(t, 0, index)
        out = torch.empty((index.numel(),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        _move_blocks(t, out, src_index=index.to(torch.int32))
        return out

    def detect_and_run_scene(self, local_images, num_images: int, shapes, pairs, **matcher_kwargs) -> SceneResult:
        """Steps 2 - 5 for this rank's views already in HBM: ``local_images`` [s,H,W] uint8 / float32 on the device, row s = image
        ``partition_images(num_images, rank, world)[s]``. The timed step of ``bench.py --mode scene``."""
        pipe = self.attach()
        return self.run_scene(self._agreed("detection", pipe.detect, local_images), num_images, shapes, pairs, **matcher_kwargs)

    @staticmethod
    def _agreed(phase: str, fn, *args):
        """``fn(*args)`` as one phase of a sharded scene: the ranks agree that nobody failed before anyone enters the next collective."""
        out = failure = None
        try:
            out = fn(*args)
        except Exception as exc:  # noqa: BLE001
            failure = exc
        parallel.agree_or_raise(failure, phase)
        return out

    # -- the generator contract ------------------------------------------------------------------------------------------------------

    def generate_correspondences(
        self, client: Any, images: List[Any], visibility_graph: List[Tuple[int, int]]
    ) -> Tuple[List[Keypoints], Dict[Tuple[int, int], np.ndarray]]:
        pairs = [(int(i1), int(i2)) for (i1, i2) in visibility_graph]
        rank, world = parallel.world_info()
        if parallel._dist() is None and (self._world_to_launch() > 1 or self._always_launch):
            imgs = BatchedDetDescCorrespondenceGenerator._resolve(client, images)
            try:
                return self._launcher().run(imgs, pairs)
            except RuntimeError:  # the pool has been torn down (every rank terminated); the next call starts a new one
                self._pool = None
                raise
        imgs = BatchedDetDescCorrespondenceGenerator._resolve(client, images)
        shapes = [(int(im.height), int(im.width)) for im in imgs]
        mine = {i: imgs[i] for i in parallel.partition_images(len(imgs), rank, world)}
        return self._generate_on_this_rank(mine, shapes, pairs)

    def _world_to_launch(self) -> int:
        if self._num_gpus is not None:
            return max(1, int(self._num_gpus))
        import torch

        return max(1, torch.cuda.device_count())

    def _generate_on_this_rank(self, my_images: Dict[int, Any], shapes: List[Tuple[int, int]], pairs: List[Tuple[int, int]]):
        """Steps 1 - 5 on this rank (collective inside a process group): ``my_images`` = {global index: Image} of the rank's share."""
        pipe = self.attach()
        rank, world = parallel.world_info()
        n = len(shapes)
        order = parallel.partition_images(n, rank, world)
        local = self._agreed("detection", pipe.detect_image_objects, [my_images[i] for i in order], self._image_batch)
        dtype, kwargs = match_output_convention(self._matcher) if self._pipeline_factory is None else (np.int64, {})
        scene = self.run_scene(local, n, shapes, pairs, **kwargs)
        # every image's keypoints for the caller: count / xy / scores all-gathered (60 KB per image; the descriptors stay where they are)
        full = parallel.all_gather_feature_table(local, n, keys=("count", "xy", "scores"))
        cnt, xy, sc = full["count"].cpu().numpy(), full["xy"].cpu().numpy(), full["scores"].cpu().numpy()
        keypoints_list = []
        for i in range(n):
            row = parallel.table_index(i, n, world) if world > 1 else i
            c = int(cnt[row])
            keypoints_list.append(Keypoints(coordinates=xy[row, :c].copy(), scales=None, responses=sc[row, :c].copy()))
        return keypoints_list, {p: scene.matches[p].astype(dtype) for p in pairs}

    # -- launcher mode ---------------------------------------------------------------------------------------------------------------

    def _launcher(self) -> "_RankPool":
        if self._pool is None:
            self._pool = _RankPool(self, self._world_to_launch(), self._backend, self._collective_timeout_s)
        return self._pool

    def close(self) -> None:
        """Stop the rank processes of launcher mode (no-op otherwise). The next call starts new ones."""
        if self._pool is not None:
            self._pool.close()
            self._pool = None


def _rank_main(rank: int, world: int, port: int, backend: str, timeout_s: float, generator: ShardedDetDescCorrespondenceGenerator, tasks, results) -> None:
    """One rank process of launcher mode: joins the process group (RCCL: one GPU per rank), then serves scenes until told to stop.

    Failure handling: a scene that raises is reported to the parent at once and the process leaves WITHOUT ``destroy_process_group`` (which
    can itself wait for peers that sit in a collective); the parent then terminates every rank (``_RankPool._kill_now``). Between the
    phases of a scene the ranks agree that nobody failed (``parallel.agree_or_raise``), so a peer's failure normally surfaces as an
    exception on every rank; what that cannot catch -- a rank killed from outside, a fault inside a collective -- ends at the
    process-group timeout (``timeout_s``) or at the parent's liveness check, whichever comes first."""
    import datetime

    import torch
    import torch.distributed as dist

    def report_and_leave(kind: str, payload) -> None:
        results.put((kind, rank, payload))
        results.close()
        results.join_thread()  # the message is on the pipe before the process goes
        os._exit(1)

    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes of one node
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")  # a collective that times out tears the process down instead of hanging it
        timeout = datetime.timedelta(seconds=timeout_s)
        if backend == "nccl":
            torch.cuda.set_device(rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank), timeout=timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    except Exception:  # noqa: BLE001
        report_and_leave("start_error", traceback.format_exc())
    results.put(("ready", rank, None))
    while True:
        task = tasks.get()
        if task is None:
            break
        my_images, shapes, pairs = task
        try:
            out = generator._generate_on_this_rank(my_images, shapes, pairs)
        except parallel.PeerRankFailed as exc:  # the rank that failed sends its own traceback; this one only confirms it left the scene
            report_and_leave("peer_error", str(exc))
        except Exception:  # noqa: BLE001 - reported to the parent, which tears the pool down
            report_and_leave("error", traceback.format_exc())
        if rank == 0:
            results.put(("done", rank, out))
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


class _RankPool:
    """The rank processes of launcher mode: started once ("spawn": a fresh interpreter per GPU, no inherited device state), fed one scene
    at a time. Each rank receives only its own share of the images (cyclic ownership); rank 0 returns the result. Any failure -- a rank
    reports an exception, a rank process dies, nobody answers in time -- terminates ALL rank processes at once (peers of a failed rank may
    sit in a collective that will never complete) and raises ``RuntimeError`` in the caller; the generator starts a new pool on its next call."""

    START_TIMEOUT_S = 600.0
    START_ATTEMPTS = 3  # the rendezvous port is picked by binding port 0 and closing the socket: somebody else may take it before rank 0 binds

    def __init__(self, generator: ShardedDetDescCorrespondenceGenerator, world: int, backend: str, collective_timeout_s: float = 600.0):
        import multiprocessing as mp

        self._ctx = mp.get_context("spawn")
        self.world = world
        self._procs, self._tasks, self._results = [], [], None
        self._finalizer = None
        for attempt in range(self.START_ATTEMPTS):
            failure = self._start(generator, backend, collective_timeout_s)
            if failure is None:
                return
            self._kill_now(self._procs, self._results)
            rank, payload = failure
            in_use = "EADDRINUSE" in payload or "address already in use" in payload.lower()
            if not in_use or attempt == self.START_ATTEMPTS - 1:
                raise RuntimeError(f"sharded generator: rank {rank} failed to start\n{payload}")

    def _start(self, generator, backend: str, collective_timeout_s: float):
        """Start the rank processes on a fresh port; None when all of them joined the process group, else (rank, what it said)."""
        ctx = self._ctx
        self._results = ctx.Queue()
        self._tasks = [ctx.Queue() for _ in range(self.world)]
        port = _free_port()
        self._procs = [ctx.Process(target=_rank_main, args=(r, self.world, port, backend, collective_timeout_s, generator, self._tasks[r], self._results),
                                   daemon=True) for r in range(self.world)]
        for p in self._procs:
            p.start()
        if self._finalizer is not None:
            atexit.unregister(self._finalizer)
        self._finalizer = weakref.finalize(self, _RankPool._shutdown, self._procs, self._tasks, self._results)
        atexit.register(self._finalizer)
        ready = 0
        while ready < self.world:
            kind, rank, payload = self._get(self.START_TIMEOUT_S)
            if kind != "ready":
                return rank, str(payload)
            ready += 1
        return None

    def _get(self, timeout: float):
        import queue

        waited = 0.0
        while True:
            try:
                return self._results.get(timeout=1.0)
            except queue.Empty:
                waited += 1.0
                dead = [r for r, p in enumerate(self._procs) if not p.is_alive() and p.exitcode not in (0, None)]
                if dead:
                    try:  # its last words may still be on the pipe
                        return self._results.get(timeout=0.5)
                    except queue.Empty:
                        return ("error", dead[0], f"rank process exited with code {self._procs[dead[0]].exitcode}")
                if waited >= timeout:
                    return ("error", -1, f"no answer from the rank processes within {timeout:.0f} s")

    def run(self, imgs: List[Any], pairs: List[Tuple[int, int]], timeout: float = 3600.0):
        shapes = [(int(im.height), int(im.width)) for im in imgs]
        for r in range(self.world):
            self._tasks[r].put(({i: imgs[i] for i in parallel.partition_images(len(imgs), r, self.world)}, shapes, pairs))
        kind, rank, payload = self._get(timeout)
        if kind == "peer_error":  # a healthy rank left because a peer failed: the peer's own traceback is on its way (or its process is gone)
            import time

            first, deadline = (kind, rank, payload), time.monotonic() + 10.0
            while kind == "peer_error" and time.monotonic() < deadline:
                kind, rank, payload = self._get(max(1.0, deadline - time.monotonic()))
            if kind != "error" or rank == -1:
                kind, rank, payload = first
        if kind != "done":
            # a rank that was killed from outside says nothing; what its peers report is a broken collective. Name the processes that are gone.
            gone = [f"rank {r} exit code {p.exitcode}" for r, p in enumerate(self._procs) if not p.is_alive() and p.exitcode not in (0, 1, None)]
            self.close(graceful=False)
            raise RuntimeError(f"sharded generator: rank {rank} failed" + (f" (rank processes already gone: {', '.join(gone)})" if gone else "") + f"\n{payload}")
        return payload

    @staticmethod
    def _kill_now(procs, results) -> None:
        """Terminate every rank process without asking: after a failure the healthy ranks may sit in a collective that will never complete."""
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=5)
        for p in procs:
            if p.is_alive():
                p.kill()
                p.join(timeout=5)
        if results is not None:  # late messages of the dying ranks must not meet the next reader
            import queue

            try:
                while True:
                    results.get_nowait()
            except (queue.Empty, OSError, ValueError, EOFError):
                pass

    @staticmethod
    def _shutdown(procs, tasks, results=None) -> None:
        for q in tasks:
            try:
                q.put(None)
            except Exception:  # noqa: BLE001
                pass
        for p in procs:
            p.join(timeout=20)
        _RankPool._kill_now(procs, results)

    def close(self, graceful: bool = True) -> None:
        if not graceful:
            self._kill_now(self._procs, self._results)
        if self._finalizer is not None:
            self._finalizer()
