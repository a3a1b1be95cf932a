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
    assert any("all_gather_into_tensor" in c for c in dist["distributed"]["collectives"])
    assert "distributed" not in plain
    assert dist["config"]["pairs_per_gpu_per_step"] == 30 and dist["scaling"] == "strong"
    # the exchange step changes nothing: same pairs, same match lists (checksum over the gathered (K,2) arrays)
    assert dist["match_digest"] == plain["match_digest"] and dist["config"]["matches_per_pair"] == plain["config"]["matches_per_pair"] > 5


@pytest.mark.gpu
def test_two_ranks_on_one_device_if_rccl_allows_it(gpu_device):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(REPO / "tests" / "rccl_two_ranks.py")]
    try:
        run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=150, cwd=str(REPO))
    except subprocess.TimeoutExpired:
        pytest.skip("two RCCL ranks on one device did not finish within 150 s")
    if run.returncode != 0:
        pytest.skip("RCCL does not run two ranks on one device here: " + (run.stdout + run.stderr)[-300:].replace("\n", " "))
    assert run.stdout.count("rccl_two_ranks OK") == 2
