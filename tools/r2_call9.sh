#!/bin/bash
set -u
OUT=gpurun_out/r2c9
mkdir -p $OUT
GTSFM_ATTENTION=dma timeout 900 python -m pytest tests/test_matchers_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -3
for a in mfma dma mfma dma; do
  GTSFM_ATTENTION=$a timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$a.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$a.json")); print("$a", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
