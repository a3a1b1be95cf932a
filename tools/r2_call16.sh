#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c16
mkdir -p $OUT
for nb in 0 1 0 1; do
  GTSFM_GEMM_NB=$nb timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_nb$nb.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_nb$nb.json")); print("nb=$nb", d["value"], [r["frac"] for r in d["roofline_other"][:3]])
PY
done
GTSFM_GEMM_NB=1 timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | head -c 120; echo
cd /tmp && export TMPDIR=/tmp
for shape in "256 768" "512 512" "512 256"; do
  TAG=$(echo $shape | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    GTSFM_GEMM_NB=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py gemm $shape > $OUT/pmc.log 2>&1
    python - <<PY
import csv, glob
tot, n = 0.0, 0
for path in glob.glob("$OUT/pmc_${TAG}_$C/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "gemm_dma_walk" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print("nb=1 gemm $shape $C avg per dispatch (KB):", tot / max(n, 1), "dispatches", n)
PY
    rm -rf $OUT/pmc_${TAG}_$C
  done
done
