#!/bin/bash
# Round-end GPU pass: parity tests, smoke, every bench line kept under profiles/, GEMM shapes + cycle budget, kernel stats +
# PMC passes (tools/prof_r02.sh).   gpurun --timeout 2400 -- 'bash tools/final_round.sh'
set -u
OUT=gpurun_out/final
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ) > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
( time timeout 900 python bench.py --steps 5 --warmup 2 ) > $OUT/bench_lightglue.json 2> $OUT/bench_lightglue.err; head -c 250 $OUT/bench_lightglue.json; echo; tail -3 $OUT/bench_lightglue.err
timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 3 --warmup 1 --no-secondary > $OUT/bench_superglue_sinkhorn100.json 2>/dev/null; head -c 200 $OUT/bench_superglue_sinkhorn100.json; echo
timeout 600 python bench.py --matcher superglue --sinkhorn 20 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > $OUT/bench_superglue_sinkhorn20.json 2>/dev/null; head -c 200 $OUT/bench_superglue_sinkhorn20.json; echo
timeout 900 python bench.py --mode scene --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_scene_config4.json 2>/dev/null; head -c 200 $OUT/bench_scene_config4.json; echo
timeout 300 python bench.py --matcher none --height 480 --width 640 --images 256 --steps 3 --warmup 1 > $OUT/bench_config2.json 2>/dev/null; head -c 200 $OUT/bench_config2.json; echo
timeout 300 python bench.py --matcher none --images 64 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_superpoint_only.json 2>/dev/null; head -c 200 $OUT/bench_superpoint_only.json; echo
for shape in "131072 256 768" "131072 512 512" "131072 512 256 1" "131072 256 512" "131072 256 256" "32768 256 768" "5000 256 4800"; do timeout 60 tools/bin/gemm_dma_walk $shape | tail -1; done | tee $OUT/gemm_shapes.txt
for shape in "131072 256 768" "131072 512 512"; do timeout 60 tools/bin/gemm_dma_walk_trace $shape; done > $OUT/gemm_cycle_budget.txt 2>&1
bash tools/prof_r02.sh 2>&1 | tail -50
