#!/bin/bash
set -u
OUT=gpurun_out/r2c10
mkdir -p $OUT
GTSFM_LIB=$PWD/gtsfm_amd/libgtsfm_amd_atint.so timeout 900 python -m pytest tests/test_matchers_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "attention or lightglue_matches or superglue_matches" 2>&1 | tail -3
for a in base atint base atint; do
  L=""; [ $a = atint ] && L=$PWD/gtsfm_amd/libgtsfm_amd_atint.so
  GTSFM_LIB=$L timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$a.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$a.json")); print("$a", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
