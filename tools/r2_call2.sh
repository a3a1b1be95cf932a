#!/bin/bash
# Round-2 GPU call 2: register-row sweep kernels (Sinkhorn, LightGlue double softmax, fused extraction)
set -u
OUT=gpurun_out/r2c2
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
for G in 160 0 64 100 220; do
  echo "== GTSFM_SINKHORN_GROUP_MB=$G"
  GTSFM_SINKHORN_GROUP_MB=$G timeout 300 python bench.py --matcher superglue --sinkhorn 100 --pairs 256 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_sg100_g$G.err | tail -1 > $OUT/bench_sg100_g$G.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_sg100_g$G.json"))
print(d["value"], [ (r["kernel"][:20], r.get("avg_iteration_ms"), r["achieved"], r["frac"]) for r in d["roofline_other"] if r["bound"] == "hbm"])
PY
done
echo "== LDS path"
GTSFM_SWEEP=lds timeout 300 python bench.py --matcher superglue --sinkhorn 100 --pairs 256 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary 2> $OUT/bench_lg.err | tail -1 > $OUT/bench_lg.json; cut -c1-200 $OUT/bench_lg.json
timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 2 --warmup 1 --no-secondary 2> $OUT/bench_sg100.err | tail -1 > $OUT/bench_sg100.json; cut -c1-200 $OUT/bench_sg100.json
python - <<PY
import json
for f in ("bench_lg", "bench_sg100"):
    d = json.load(open("$OUT/%s.json" % f))
    print(f, d["value"], d.get("parity_check"), d["config"]["matches_per_pair"])
PY
