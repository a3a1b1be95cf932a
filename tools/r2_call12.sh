#!/bin/bash
set -u
OUT=gpurun_out/r2c12
mkdir -p $OUT
( time timeout 900 python bench.py --steps 5 --warmup 2 ) > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -3 $OUT/bench_headline.err; head -c 300 $OUT/bench_headline.json; echo
GTSFM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --mode scene --images 33 --pairs 500 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_scene_small_rccl.json 2> $OUT/bench_scene_small.err; tail -2 $OUT/bench_scene_small.err; head -c 400 $OUT/bench_scene_small_rccl.json; echo
timeout 900 python bench.py --mode scene --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_scene_config4.json 2> $OUT/bench_scene.err; tail -2 $OUT/bench_scene.err; head -c 400 $OUT/bench_scene_config4.json; echo
timeout 600 python bench.py --matcher none --height 480 --width 640 --images 256 --steps 3 --warmup 1 > $OUT/bench_config2.json 2> $OUT/bench_config2.err; head -c 300 $OUT/bench_config2.json; echo
timeout 600 python bench.py --matcher superglue --sinkhorn 20 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | head -c 200; echo
python - <<PY
import json
d = json.load(open("$OUT/bench_headline.json".replace("real", ""))) if False else None
PY
