#!/bin/bash
set -u
OUT=gpurun_out/r2c14
mkdir -p $OUT
timeout 900 python -m pytest tests/test_matchers_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "superglue or sinkhorn or generator or pipeline" 2>&1 | tail -3
timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_sg100.err | tail -1 > $OUT/bench_sg100.json
python - <<PY
import json
d = json.load(open("$OUT/bench_sg100.json")); print(d["value"], [(r["kernel"][:24], r.get("avg_iteration_ms"), r["achieved"], r["frac"]) for r in d["roofline_other"] if r["bound"] == "hbm"])
PY
