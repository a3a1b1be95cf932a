"""CPU: ``ShardedDetDescCorrespondenceGenerator`` -- the product class that shards ONE scene over the GPUs of a node (SURVEY.md
section 8e) -- run as a CLASS on gloo with the stand-in pipeline (``gtsfm_amd.utils.standin``: no kernels; an image's pseudo-features
depend on its bytes alone, a pair's pseudo-matches on the two images' pseudo-features alone). What is exercised is everything the class
adds around the kernels: launcher mode (it starts its own rank processes) with 2 and 8 ranks, joined mode (called collectively from an
existing process group), cyclic image ownership, the ``all_to_all_single`` exchange that ships an image only to the ranks whose pairs
touch it, 2-D block-cyclic pair ownership with ranks that own no pair or no image, pairs with an empty keypoint set, the ragged gather --
every pair back exactly once -- and the result must EQUAL the single-process generator's, keypoints and match arrays.
The same class runs on RCCL on the device in ``tests/test_rccl_gpu.py``."""

import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from gtsfm_amd import parallel
from gtsfm_amd.common.image import Image
from gtsfm_amd.frontend.correspondence_generator.sharded_det_desc_correspondence_generator import ShardedDetDescCorrespondenceGenerator
from gtsfm_amd.utils.standin import failing_stand_in_pipeline_factory, stand_in_pipeline_factory


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(n: int, seed: int = 7):
    """n small images of mixed shapes (gray and RGB); image 2 starts with 255 = the stand-in's "no keypoints" marker."""
    rng = np.random.default_rng(seed)
    images = []
    for i in range(n):
        shape = (6 + i % 3, 8) if i % 2 else (6 + i % 3, 8, 3)
        a = rng.integers(0, 255, shape).astype(np.uint8)
        a.reshape(-1)[0] = 255 if i == 2 else 7
        images.append(Image(value_array=a, file_name=f"{i}.png"))
    return images


def _generator(num_gpus):
    return ShardedDetDescCorrespondenceGenerator(None, types.SimpleNamespace(max_keypoints=24), num_gpus=num_gpus, backend="gloo",
                                                 pipeline_factory=stand_in_pipeline_factory)


def _assert_same(got, want, pairs):
    kp_g, m_g = got
    kp_w, m_w = want
    assert len(kp_g) == len(kp_w)
    for a, b in zip(kp_g, kp_w):
        np.testing.assert_array_equal(a.coordinates, b.coordinates)
        np.testing.assert_array_equal(a.responses, b.responses)
    assert list(m_g) == list(pairs) == list(m_w)  # the visibility graph's order, every edge exactly once
    for p in pairs:
        assert m_g[p].dtype == m_w[p].dtype and m_g[p].shape[1] == 2
        np.testing.assert_array_equal(m_g[p], m_w[p])


@pytest.mark.parametrize("world,n,npairs", [(2, 9, 30), (8, 5, 7), (8, 21, 150)])
def test_launcher_mode_equals_the_single_process_generator(world, n, npairs):
    """The class starts ``world`` rank processes itself (gloo here, RCCL on a GPU node), feeds each its share of the images and returns
    rank 0's gathered result: equal to the same class run in this process without a process group. (8, 5, 7): ranks without a pair AND
    ranks without an image take part in every collective."""
    images = _scene(n)
    pairs = parallel.exhaustive_pairs(n)[:npairs]
    pairs = pairs[::-1]  # not sorted: the result keeps the graph's order
    single = _generator(1).generate_correspondences(None, images, pairs)
    assert len(single[0][2]) == 0 and all(single[1][p].shape == (0, 2) for p in pairs if 2 in p)  # the image without keypoints
    assert sum(len(m) for m in single[1].values()) > 0
    gen = _generator(world)
    try:
        got = gen.generate_correspondences(None, images, pairs)
        _assert_same(got, single, pairs)
        again = gen.generate_correspondences(None, images[: n - 1], [p for p in pairs if n - 1 not in p])  # the ranks serve a second scene
        assert len(again[0]) == n - 1
    finally:
        gen.close()
    if world == 8 and n == 5:
        parts = [parallel.partition_pairs_2d(pairs, r, 8) for r in range(8)]
        assert any(not part for part in parts) and not parallel.partition_images(5, 7, 8)


def _joined_worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        images = _scene(11)
        pairs = parallel.exhaustive_pairs(11)[:40]
        gen = _generator(None)
        kps, matches = gen.generate_correspondences(None, images, pairs)  # collective: every rank, same arguments
        plan = gen.last_scene.plan
        # the exchange shipped only the images this rank's pairs touch, and the table is ordered by (owner, slot)
        assert plan.table_images == sorted(parallel.images_touched(plan.my_pairs), key=lambda i: (i % world, i // world))
        assert gen.last_scene.table["xy"].shape[0] == len(plan.table_images) <= 11
        assert sum(plan.recv_counts) == len(plan.table_images)
        np.save(os.path.join(tmp, f"m{rank}.npy"), np.concatenate([matches[p].reshape(-1) for p in pairs] + [np.array([len(k) for k in kps])]))
        np.save(os.path.join(tmp, f"t{rank}.npy"), np.array(plan.table_images))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["all_to_all", "all_gather"])
def test_joined_mode_is_a_collective_call_and_every_rank_gets_the_result(tmp_path, monkeypatch, exchange):
    """Both forms of the exchange step: the ragged ``all_to_all_single`` (default: an image goes only to the ranks that match it) and the
    ``GTSFM_SHARD_EXCHANGE=all_gather`` escape hatch (everything to everybody, then the table rows are selected) give the same tables and results."""
    monkeypatch.setenv("GTSFM_SHARD_EXCHANGE", exchange)
    world = 3  # a 1 x 3 process grid
    mp.spawn(_joined_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    images, pairs = _scene(11), parallel.exhaustive_pairs(11)[:40]
    kps, matches = _generator(1).generate_correspondences(None, images, pairs)
    want = np.concatenate([matches[p].reshape(-1) for p in pairs] + [np.array([len(k) for k in kps])])
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"m{r}.npy"), want)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_scene_plan_is_consistent_across_ranks(world):
    """Every rank computes every rank's plan: what rank a sends to rank b is what rank b expects from rank a, tables hold exactly the
    touched images, pairs are covered once, and on 8 ranks BASELINE config 4's tables hold at most 76 of the 101 images."""
    n = 101
    pairs = parallel.exhaustive_pairs(n)[:5000]
    plans = [parallel.ScenePlan(n, pairs, r, world) for r in range(world)]
    assert sorted(p for pl in plans for p in pl.my_pairs) == sorted(pairs)
    for a in plans:
        assert a.table_images == sorted(set(a.table_images), key=lambda i: (i % world, i // world))
        assert set(a.table_images) == set(parallel.images_touched(a.my_pairs))
        assert [(a.table_images[i], a.table_images[j]) for i, j in a.local_pairs] == a.my_pairs
        for b in plans:
            assert a.send_counts[b.rank] == b.recv_counts[a.rank]
            sent = [a.my_images[s] for s in a.send_slots[b.rank]]
            off = sum(b.recv_counts[: a.rank])
            assert sent == b.table_images[off : off + b.recv_counts[a.rank]]
    if world == 8:
        assert max(len(pl.table_images) for pl in plans) <= 76 and max(pl.images_sent() for pl in plans) <= 13 * 7


def test_constructor_checks_and_pickling(tmp_path):
    """Built from the two plugin objects like the reference's generator; pickles before any device state or rank process exists."""
    import pickle

    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.utils import synthetic

    torch.save(synthetic.synthetic_superpoint_state_dict(), str(tmp_path / "sp.pth"))
    torch.save(synthetic.synthetic_lightglue_state_dict(num_layers=2), str(tmp_path / "lg.pth"))
    det = SuperPointDetectorDescriptor(max_keypoints=1000, weights_path=tmp_path / "sp.pth")
    mt = LightGlueMatcher("superpoint", weights_path=tmp_path / "lg.pth")
    gen = ShardedDetDescCorrespondenceGenerator(mt, det, num_gpus=8)
    clone = pickle.loads(pickle.dumps(gen))
    assert clone._pipe is None and clone._pool is None and clone._matcher._model is None and clone._world_to_launch() == 8
    assert "ShardedDetDescCorrespondenceGenerator" in repr(clone)
    with pytest.raises(TypeError):
        ShardedDetDescCorrespondenceGenerator(object(), det)
    with pytest.raises(TypeError):
        ShardedDetDescCorrespondenceGenerator(mt, object())


def _children_alive():
    import multiprocessing

    return [p for p in multiprocessing.active_children() if p.is_alive()]


@pytest.mark.parametrize("phase,how", [("match", "raise"), ("detect", "raise"), ("match", "exit")])
def test_a_rank_that_fails_mid_scene_fails_the_call_quickly_and_leaves_no_process_behind(monkeypatch, phase, how):
    """one of three gloo ranks fails in the middle of a scene -- by raising (its peers learn of it at the
    agreement point before the next collective and leave with it) or by dying without a word (its peers sit in a collective that will never
    complete; the parent sees the dead process and terminates them). Either way the caller gets ``RuntimeError`` well inside a minute, no
    rank process survives, and the generator serves the next scene with a fresh pool."""
    import time

    monkeypatch.setenv("GTSFM_STANDIN_FAIL_RANK", "1")
    monkeypatch.setenv("GTSFM_STANDIN_FAIL_PHASE", phase)
    monkeypatch.setenv("GTSFM_STANDIN_FAIL_HOW", how)
    images, pairs = _scene(9), parallel.exhaustive_pairs(9)[:30]
    gen = ShardedDetDescCorrespondenceGenerator(None, types.SimpleNamespace(max_keypoints=24), num_gpus=3, backend="gloo",
                                                pipeline_factory=failing_stand_in_pipeline_factory, collective_timeout_s=120.0)
    before = len(_children_alive())
    t0 = time.monotonic()
    try:
        with pytest.raises(RuntimeError) as err:
            gen.generate_correspondences(None, images, pairs)
        took = time.monotonic() - t0
        assert took < 60.0, f"the failure took {took:.0f} s to reach the caller"
        if how == "raise":  # the failing rank's own traceback, not a peer's "somebody failed"
            assert f"injected failure in {phase} on rank 1" in str(err.value) and "rank 1 failed" in str(err.value)
        else:
            assert "rank 1 exit code 3" in str(err.value) or "rank 1 failed\nrank process exited with code 3" in str(err.value)
        assert gen._pool is None
        deadline = time.monotonic() + 10.0
        while len(_children_alive()) > before and time.monotonic() < deadline:
            time.sleep(0.1)
        assert len(_children_alive()) == before, "rank processes outlived the failed call"
        # the next scene gets a fresh pool; with the injection switched off it equals the single-process result
        monkeypatch.setenv("GTSFM_STANDIN_FAIL_RANK", "-1")
        got = gen.generate_correspondences(None, images, pairs)
        _assert_same(got, _generator(1).generate_correspondences(None, images, pairs), pairs)
    finally:
        gen.close()


def _disagreeing_worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["GTSFM_SHARD_EXCHANGE"] = "all_gather" if rank == 1 else "all_to_all"  # rank 1's launcher exported something else
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel.agreed_exchange_mode() == "all_to_all"  # rank 0's, everywhere
        images, pairs = _scene(7), parallel.exhaustive_pairs(7)
        _, matches = _generator(None).generate_correspondences(None, images, pairs)
        np.save(os.path.join(tmp, f"d{rank}.npy"), np.concatenate([matches[p].reshape(-1) for p in pairs]))
    finally:
        dist.destroy_process_group()


def test_ranks_with_different_exchange_settings_follow_rank_zero(tmp_path):
    """GTSFM_SHARD_EXCHANGE was read per rank at call time; ranks that disagreed entered different collectives and hung.
    The mode is now rank 0's, broadcast once per process group."""
    mp.spawn(_disagreeing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    images, pairs = _scene(7), parallel.exhaustive_pairs(7)
    _, matches = _generator(1).generate_correspondences(None, images, pairs)
    want = np.concatenate([matches[p].reshape(-1) for p in pairs])
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"d{r}.npy"), want)
