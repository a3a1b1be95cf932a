"""-m gpu: the multi-GPU path's collectives executed by RCCL on a device. The driver's GPU box has ONE
MI355X, so (1) bench.py's scene mode runs with a one-rank "nccl" process group -- init_process_group, the weight broadcast,
all_gather_into_tensor of the feature table, the ragged match gather, barrier and the max-reduce of the step time all go through
RCCL on the device -- and must give the same matches as the undistributed pipeline; (2) two ranks are started on the one device,
which RCCL normally refuses ("Duplicate GPU detected"): the test then skips, cleanly and quickly."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(extra_env, *flags):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, str(REPO / "bench.py"), "--mode", "scene", "--images", "9", "--pairs", "30", "--size", "256", "--keypoints", "300", "--sinkhorn", "20",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-roofline", *flags]
    run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(REPO))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    return json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_scene_mode_through_rccl_on_the_device(gpu_device):
    dist = _bench({"GTSFM_BENCH_FORCE_DIST": "1"}, "--dump-matches", "1")
    plain = _bench({}, "--dump-matches", "1")
    assert dist["distributed"]["backend"] == "nccl" and dist["distributed"]["world_size"] == 1
    assert any("all_gather_into_tensor" in c for c in dist["distributed"]["collectives"]) and any("all_to_all_single" in c for c in dist["distributed"]["collectives"])
    assert dist["exchange"]["class"].endswith("ShardedDetDescCorrespondenceGenerator")  # bench.py's scene step IS the product class's
    assert "distributed" not in plain
    assert dist["config"]["pairs_per_gpu_per_step"] == 30 and dist["scaling"] == "strong"
    # the exchange step changes nothing: same pairs, same match lists (checksum over the gathered (K,2) arrays)
    assert dist["match_digest"] == plain["match_digest"] and dist["config"]["matches_per_pair"] == plain["config"]["matches_per_pair"] > 5


@pytest.mark.gpu
def test_replica_mode_under_the_drivers_launcher_through_rccl(gpu_device):
    """The driver's multi-GPU command shape -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N --steps K --warmup W`` -- with the one rank a one-GPU box allows and the process group forced on: rendezvous from the launcher's
    environment, RCCL weight broadcasts, barrier + max-reduce of the step time, one JSON line from rank 0 in replica mode (weak scaling)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GTSFM_BENCH_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(REPO / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "30", "--keypoints", "1024", "--size", "512", "--no-secondary",
           "--no-cpu-baseline", "--no-roofline"]
    run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(REPO))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["distributed"] == {"backend": "nccl", "world_size": 1, "collectives": ["broadcast (packed weights)", "barrier", "all_reduce MAX (step time)"]}
    assert r["n_gpus"] == 1 and r["scaling"] == "weak" and r["config"]["mode"] == "replica" and r["config"]["pairs_per_gpu_per_step"] == 30
    assert r["value"] > 0 and r["config"]["matches_per_pair"] > 5 and r["steps"] == 2 and r["warmup"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("matcher", ["superglue", "lightglue"])
def test_sharded_generator_class_through_rccl_equals_the_single_process_generators(gpu_device, matcher, tmp_path):
    """``ShardedDetDescCorrespondenceGenerator.generate_correspondences`` in joined mode on a one-rank RCCL group (subprocess) == the same
    class without a process group == ``BatchedDetDescCorrespondenceGenerator``: identical keypoints and (K, 2) arrays for every edge,
    including a fully masked view (empty keypoint set: its pairs come back as (0, 2) arrays) and an RGB view."""
    sys.path.insert(0, str(REPO / "tests"))
    import rccl_sharded_one_rank as helper
    from gtsfm_amd.frontend.correspondence_generator.batched_det_desc_correspondence_generator import BatchedDetDescCorrespondenceGenerator

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, str(REPO / "tests" / "rccl_sharded_one_rank.py"), matcher], capture_output=True, text=True, env=env, timeout=600, cwd=str(REPO))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    joined = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert joined["backend"] == "nccl" and joined["table_images"] == [0, 1, 2, 3, 4, 5]

    gen = helper.build(str(tmp_path), matcher, 1)
    images, pairs = helper.scene()
    kps, matches = gen.generate_correspondences(None, images, pairs)
    single = helper.digest(kps, matches, pairs)
    assert single["keypoints"][3] == 0 and min(single["keypoints"][:3]) > 100 and single["matches"] > 50
    assert all(matches[p].shape == (0, 2) for p in pairs if 3 in p)
    assert single["dtype"] == ("uint32" if matcher == "superglue" else "int64")
    assert {k: joined[k] for k in single} == single
    batched = BatchedDetDescCorrespondenceGenerator(gen._matcher, gen._detector_descriptor)
    kps_b, matches_b = batched.generate_correspondences(None, images, pairs)
    assert helper.digest(kps_b, matches_b, pairs) == single


@pytest.mark.gpu
def test_sharded_generator_launcher_mode_on_the_device(gpu_device, tmp_path):
    """LAUNCHER mode on real hardware, with the one rank a one-GPU box allows: the class spawns its rank process ("spawn": a fresh
    interpreter), ships it the pickled generator (plugins without device state) and the rank's images, the rank joins an RCCL process group,
    builds its engines, runs the scene and hands the result back -- equal to the in-process run; a second scene reuses the rank; close() ends it."""
    sys.path.insert(0, str(REPO / "tests"))
    import rccl_sharded_one_rank as helper

    inproc = helper.build(str(tmp_path), "lightglue", 1)
    images, pairs = helper.scene()
    want = helper.digest(*inproc.generate_correspondences(None, images, pairs), pairs)
    gen = helper.build(str(tmp_path), "lightglue", 1)
    gen._always_launch = True
    try:
        got = helper.digest(*gen.generate_correspondences(None, images, pairs), pairs)
        assert got == want and gen._pool is not None and gen._pool.world == 1 and gen._pipe is None  # the caller holds no device state of its own
        again = helper.digest(*gen.generate_correspondences(None, images[:5], [p for p in pairs if 5 not in p]), [p for p in pairs if 5 not in p])
        assert again["keypoints"] == want["keypoints"][:5]
        procs = list(gen._pool._procs)
    finally:
        gen.close()
    assert gen._pool is None and all(not p.is_alive() for p in procs)


@pytest.mark.gpu
def test_two_ranks_on_one_device_if_rccl_allows_it(gpu_device):
    """Two RCCL ranks through the front-end's collectives (broadcast, all_gather, the ragged all_to_all_single, the ragged match gather). On a box
    with two or more GPUs the ranks take one device each and the run MUST pass -- the first time RCCL runs between distinct devices in this project
    is then a test, not the driver's scaling bench. On a one-GPU box both ranks share device 0, which RCCL normally refuses ("Duplicate GPU
    detected"): skipped there."""
    import torch

    distinct = torch.cuda.device_count() >= 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(REPO / "tests" / "rccl_two_ranks.py")]
    try:
        run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300 if distinct else 150, cwd=str(REPO))
    except subprocess.TimeoutExpired:
        if distinct:
            raise
        pytest.skip("two RCCL ranks on one device did not finish within 150 s")
    if run.returncode != 0 and not distinct:
        pytest.skip("RCCL does not run two ranks on one device here: " + (run.stdout + run.stderr)[-300:].replace("\n", " "))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert run.stdout.count("rccl_two_ranks OK") == 2
    assert ("one rank per GPU" in run.stdout) == distinct
