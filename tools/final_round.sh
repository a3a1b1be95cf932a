#!/bin/bash
# Round-end GPU pass: parity tests, the bench lines that go to profiles/, kernel stats + PMC passes (tools/prof.sh).
set -u
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/final/pytest_gpu.txt
timeout 400 python bench.py > gpurun_out/final/bench_lightglue.json 2> gpurun_out/final/bench_lightglue.err; tail -c 600 gpurun_out/final/bench_lightglue.json
timeout 300 python bench.py --matcher superglue --sinkhorn 100 --no-cpu-baseline > gpurun_out/final/bench_superglue_sinkhorn100.json 2>/dev/null; cut -c1-150 gpurun_out/final/bench_superglue_sinkhorn100.json
timeout 300 python bench.py --matcher superglue --sinkhorn 20 --no-cpu-baseline > gpurun_out/final/bench_superglue_sinkhorn20.json 2>/dev/null; cut -c1-150 gpurun_out/final/bench_superglue_sinkhorn20.json
timeout 300 python bench.py --matcher none --images 64 --no-cpu-baseline > gpurun_out/final/bench_superpoint_only.json 2>/dev/null; cut -c1-150 gpurun_out/final/bench_superpoint_only.json
timeout 300 python bench.py --matcher none --height 480 --width 640 --images 256 --no-cpu-baseline > gpurun_out/final/bench_config2.json 2>/dev/null; cut -c1-150 gpurun_out/final/bench_config2.json
timeout 300 python bench.py --pair-definition independent --pairs 500 --steps 2 --no-cpu-baseline > gpurun_out/final/bench_independent.json 2>/dev/null; cut -c1-150 gpurun_out/final/bench_independent.json
