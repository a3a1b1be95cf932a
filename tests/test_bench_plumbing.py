"""CPU: bench.py's launcher, sharding and collective plumbing.

``python bench.py --gpus N`` must start N ranks by itself; ``--mode scene`` shards ONE scene (BASELINE config 4) over
the ranks. ``--plumbing-only`` replaces the kernels with stand-ins so that everything around them -- self-launch under
torch.distributed.run, gloo rendezvous on 127.0.0.1, cyclic / 2-D block-cyclic ownership, the feature all-gather, the
ragged gather of the match lists, barrier + max-over-ranks timing, the one JSON line of rank 0 -- runs here."""

import json
import os
import subprocess
import sys

import pytest

from gtsfm_amd import parallel
from tests.conftest import REPO


def _run_bench(*flags, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    if "--details-file" not in flags:  # the tests' stand-in runs must not overwrite a measured run's record under gpurun_out/
        flags = (*flags, "--details-file", "")
    p = subprocess.run([sys.executable, str(REPO / "bench.py"), *flags], capture_output=True, text=True, timeout=600, env=e, cwd=str(REPO))
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_bench_self_launches_and_shards_a_scene_over_two_ranks():
    p, lines = _run_bench("--gpus", "2", "--mode", "scene", "--plumbing-only", "--images", "9", "--pairs", "30", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["plumbing_only"] is True and r["steps"] == 2 and r["warmup"] == 1
    assert r["config"]["mode"] == "scene" and r["config"]["images_per_gpu_per_step"] == 5
    assert r["config"]["pairs_per_gpu_per_step"] == len(parallel.partition_pairs_2d(parallel.exhaustive_pairs(9)[:30], 0, 2))
    assert r["value"] > 0 and r["unit"] == "image-pairs/s"


def test_bench_eight_ranks_scene_with_idle_ranks():
    """The launch the driver makes on an 8-GPU node (`bench.py --gpus 8`, here self-launched on gloo with stand-in kernels): a small
    scene on the 2 x 4 process grid leaves ranks without a pair -- they still take part in the feature all-gather and the ragged
    match gather -- and every pair of the scene comes back exactly once."""
    p, lines = _run_bench("--gpus", "8", "--mode", "scene", "--plumbing-only", "--images", "5", "--pairs", "7", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    r = json.loads(lines[0])
    pairs = parallel.exhaustive_pairs(5)[:7]
    parts = [parallel.partition_pairs_2d(pairs, rank, 8) for rank in range(8)]
    idle = sum(1 for part in parts if not part)
    assert idle >= 1 and sorted(q for part in parts for q in part) == sorted(pairs)
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["plumbing_only"] is True
    assert r["scene_check"] == {"pairs_gathered": 7, "pairs_of_the_scene": 7, "each_pair_exactly_once": True, "ranks_without_pairs": idle}
    assert r["distributed"]["world_size"] == 8 and "all_gather_into_tensor (ragged match lists)" in r["distributed"]["collectives"] and any("all_to_all_single" in c for c in r["distributed"]["collectives"])
    assert r["config"]["pairs_per_gpu_per_step"] == len(parts[0])
    # the step is the product class's, and the exchange ships an image only to the ranks whose pairs touch it
    assert r["exchange"]["class"].endswith("ShardedDetDescCorrespondenceGenerator") and r["exchange"]["images_in_rank0_table"] == len(parallel.images_touched(parts[0]))
    assert r["exchange"]["largest_table_over_ranks"] < 5


def test_bench_replica_mode_self_launch_weak_scaling():
    p, lines = _run_bench("--gpus", "2", "--plumbing-only", "--images", "5", "--pairs", "8", "--steps", "1", "--warmup", "0")
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["pairs_per_gpu_per_step"] == 8


def test_bench_eight_ranks_replica_default_is_the_drivers_scale_command():
    """`python bench.py --gpus 8` with no --mode: the command the driver's SCALE step issues on an 8-GPU node (replica = BASELINE config 3 per
    rank, weak scaling). Self-launched here on gloo with stand-in kernels: eight ranks rendezvous on 127.0.0.1, every rank owns its own image
    set / pair list, barrier + max-over-ranks timing, rank 0 prints ONE line whose value aggregates all ranks."""
    p, lines = _run_bench("--gpus", "8", "--plumbing-only", "--images", "5", "--pairs", "8", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["scaling"] == "weak" and r["plumbing_only"] is True and r["steps"] == 2 and r["warmup"] == 1
    assert r["distributed"]["world_size"] == 8 and r["distributed"]["backend"] == "gloo"
    assert r["config"]["mode"] == "replica" and r["config"]["pairs_per_gpu_per_step"] == 8 and r["config"]["images_per_gpu_per_step"] == 5
    assert r["config"]["parallelism"].startswith("dp8")
    # whole-job aggregate: 8 ranks x 8 pairs per step over the slowest rank's step time
    assert abs(r["value"] - 8 * 8 / (r["ms_per_step"] * 1e-3)) <= 0.01 * r["value"] + 0.01
    assert "scene_check" not in r and r["higher_is_better"] is True and r["unit"] == "image-pairs/s"


def test_bench_rejects_a_mismatched_launcher():
    p, lines = _run_bench("--gpus", "2", "--plumbing-only", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout) and not lines


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_scene_partitions_cover_every_pair_once(world):
    pairs = parallel.exhaustive_pairs(101)[:5000]
    parts = [parallel.partition_pairs_2d(pairs, r, world) for r in range(world)]
    assert sorted(p for part in parts for p in part) == sorted(pairs)
    assert max(len(p) for p in parts) <= 1.06 * len(pairs) / world  # balanced: BASELINE config 4's heaviest rank
    rows, cols = parallel.process_grid(world)
    assert rows * cols == world and rows <= cols
    if world == 8:  # 2 x 4 grid: a rank touches ~ n/2 + n/4 of the images, not all of them
        assert max(len(parallel.images_touched(p)) for p in parts) <= 76
    rows_of = sorted(parallel.table_index(i, 101, world) for i in range(101))
    assert len(set(rows_of)) == 101 and rows_of[-1] < world * (-(-101 // world))


def test_default_workload_helpers():
    """The headline's shape follows from the keypoint cap: fewest images whose exhaustive pairs cover the pair count (SURVEY.md
    section 8d: P exhaustive pairs <=> n images), a power-of-two pair chunk by keypoint count."""
    import bench

    for pairs in (1, 2, 3, 100, 250, 1000, 1035, 5000):
        n = bench.fewest_images_for(pairs)
        assert n * (n - 1) // 2 >= pairs and (n - 1) * (n - 2) // 2 < pairs
    assert (bench.fewest_images_for(250), bench.fewest_images_for(1000), bench.fewest_images_for(5000)) == (23, 46, 101)
    assert (bench.default_pair_chunk(5000), bench.default_pair_chunk(2048), bench.default_pair_chunk(512), bench.default_pair_chunk(20000)) == (16, 32, 32, 4)
    args = bench.parse_args([])
    # the default IS BASELINE config 3 as written: the first 1000 exhaustive pairs of 46 views, at the reference's 5000-keypoint cap
    assert (args.keypoints, args.pairs, args.images, args.pair_chunk, args.matcher, args.mode) == (5000, 1000, 46, 16, "lightglue", "replica")
    args = bench.parse_args(["--keypoints", "2048"])
    assert (args.pairs, args.images, args.pair_chunk) == (1000, 46, 32)
    args = bench.parse_args(["--mode", "scene"])
    assert (args.pairs, args.images, args.matcher) == (5000, 101, "superglue")


def test_the_printed_line_is_one_the_driver_can_read(tmp_path):
    """Round 5's line was 21.6 KB and the driver's 8 KB tail held no parseable object. The contract now: stdout's LAST
    and only JSON line is under 4 KB; the full record goes to --details-file. Checked on a stand-in run and on round 5's own full record."""
    import bench

    side = tmp_path / "details.json"
    p, lines = _run_bench("--plumbing-only", "--images", "5", "--pairs", "8", "--steps", "1", "--warmup", "0", "--details-file", str(side))
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout.strip().splitlines()
    assert len(lines) == 1 and out[-1] == lines[0] and len(out[-1]) < 4096
    head = json.loads(out[-1])
    full = json.loads(side.read_text())
    assert head["details"] == str(side) and full["value"] == head["value"] and full["config"] == head["config"]
    # a full run's record (round 5's, 21.6 KB): the headline keeps what the driver parses and drops the bulk
    recorded = json.loads((REPO / "profiles" / "r05_final_bench_default.json").read_text())
    line = json.dumps(bench.headline_of(recorded, bench.DETAILS_FILE))
    assert len(line) < 4096 < len(json.dumps(recorded))
    head = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline", "parity_check"):
        assert key in head, key
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(head["roofline"]) and head["roofline"]["frac"] == recorded["roofline"]["frac"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(head["cpu_baseline"])
    assert len(head["roofline_other"]) <= 3 and "secondary" not in head and "step_note" not in head
    assert head["secondary_rates"]["config2_superpoint_480x640"] == recorded["secondary"]["config2_superpoint_480x640"]["value"]
    # a record that still would not fit sheds its optional fields rather than its headline
    bloated = dict(recorded, secondary={f"leg{i}": {"value": float(i), "note": "x" * 50} for i in range(400)})
    shrunk = json.loads(json.dumps(bench.headline_of(bloated, bench.DETAILS_FILE)))
    assert len(json.dumps(shrunk)) <= bench.HEADLINE_BYTES and "roofline" in shrunk and "cpu_baseline" in shrunk and "secondary_rates" not in shrunk


def test_recorded_bench_line_is_hygienic():
    """The driver-command record kept for the latest round (profiles/r06_final_bench_details.json = the full record bench.py writes beside its < 4 KB
    line; round 5's single 21 KB line before that): no roofline fraction above 1 anywhere (an algorithmic fp32 rate is never divided by the fp32 roof
    for a kernel that executes bf16 MFMAs), every `traffic` figure cites a counter file collected THAT round, the headline carries `roofline` +
    `cpu_baseline`, and the < 4 KB line derived from the record keeps them."""
    import bench

    tag, path = "r06", REPO / "profiles" / "r06_final_bench_details.json"
    if not path.exists():
        tag, path = "r05", REPO / "profiles" / "r05_final_bench_default.json"
    if not path.exists():
        pytest.skip("no recorded bench line yet")
    text = path.read_text()
    line = json.loads(text) if tag == "r06" else json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])
    head = bench.headline_of(line, bench.DETAILS_FILE)
    assert len(json.dumps(head)) < 4096 and {"roofline", "cpu_baseline", "parity_check"} <= set(head)
    fracs, notes = [], []

    def walk(node, where):
        if isinstance(node, dict):
            if "frac" in node and isinstance(node["frac"], (int, float)):
                fracs.append((where, node["frac"]))
            if node.get("traffic_note"):
                notes.append((where, node["traffic_note"]))
            if node.get("traffic") is not None:
                assert node.get("traffic_note"), f"{where}: a traffic figure without its source"
            for k, v in node.items():
                walk(v, f"{where}.{k}")
        elif isinstance(node, list):
            for i, v in enumerate(node):
                walk(v, f"{where}[{i}]")

    walk(line, "line")
    assert fracs and all(0.0 < f <= 1.0 for _, f in fracs), [x for x in fracs if not 0.0 < x[1] <= 1.0]
    assert notes and all(f"profiles/{tag}_" in n for _, n in notes), [x for x in notes if f"profiles/{tag}_" not in x[1]]
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=2e-3)
    assert "cpu_baseline" in line and line["cpu_baseline"]["kind"] in ("port", "reference")
    assert "algorithmic_tflops" in line and "executed_tflops" in line and "tflops" not in line
    assert line["executed_tflops"] < line["algorithmic_tflops"] < 157.3
    kernels = " ".join(str(r.get("kernel")) for r in line["roofline_other"])
    assert "extract_rows" in kernels and "layernorm_gelu_kernel" in kernels  # the kernel furthest below its roof is in the line


def test_the_arithmetic_flag_labels_the_line_and_sets_both_switches():
    """``--arithmetic f16x2`` (round 6): a side measurement of the headline's protocol under both opt-in switches -- the line says so in `dtype` and
    `config.arithmetic`, the secondary legs (statements about the default arithmetic) are off, and the default run stays `f32`."""
    p, lines = _run_bench("--plumbing-only", "--images", "5", "--pairs", "8", "--steps", "1", "--warmup", "0", "--arithmetic", "f16x2")
    assert p.returncode == 0, p.stderr[-2000:]
    head = json.loads(lines[-1])
    assert head["config"]["arithmetic"] == "f16x2" and "2 x fp16" in head["dtype"] and "opt-in" in head["dtype"] and "secondary_rates" not in head
    p, lines = _run_bench("--plumbing-only", "--images", "5", "--pairs", "8", "--steps", "1", "--warmup", "0")
    head = json.loads(lines[-1])
    assert head["config"]["arithmetic"] == "f32" and head["dtype"] == "f32"
    # the switches are exported before anything launches (the ranks of a self-launch inherit them)
    src = (REPO / "bench.py").read_text()
    assert src.index('os.environ["GTSFM_ATTENTION_MATH"] = os.environ["GTSFM_GEMM_MATH"] = args.arithmetic') < src.index("relaunch_one_process_per_gpu(args.gpus))")
