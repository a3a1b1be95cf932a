#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c15
mkdir -p $OUT
for nb in 0 1 2 3 4; do
  for shape in "131072 256 768" "131072 512 512" "131072 512 256 1" "131072 256 512"; do
    GTSFM_GEMM_NB=$nb timeout 60 tools/bin/gemm_dma_walk $shape | tail -1 | sed "s/^/nb=$nb: /"
  done
done 2>&1 | tee $OUT/gemm_nb.txt
cd /tmp && export TMPDIR=/tmp
for nb in 3 2; do
  for C in FETCH_SIZE; do
    GTSFM_GEMM_NB=$nb rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_nb${nb}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py gemm 256 768 > $OUT/pmc_nb${nb}_$C.log 2>&1
    python - <<PY
import csv, glob
tot, n = 0.0, 0
for path in glob.glob("$OUT/pmc_nb${nb}_$C/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "gemm_dma_walk" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print("nb=$nb $C avg per dispatch (KB):", tot / max(n, 1), "dispatches", n)
PY
    rm -rf $OUT/pmc_nb${nb}_$C
  done
done
