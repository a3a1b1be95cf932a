#!/bin/bash
# Profiling recipe (GPU box): kernel trace + stats of the default bench, then separate PMC passes (HBM traffic) on a
# short SuperPoint-only run. Outputs land in gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -o full -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/full.log 2>&1
rm -f $OUT/full/*kernel_trace.csv
cat > /tmp/pmc_workload.sh <<'SH'
python $GRAFT_REPO_ROOT/bench.py --matcher none --images 4 --steps 1 --warmup 1 --no-cpu-baseline
python $GRAFT_REPO_ROOT/tools/gpu_time_lg.py
SH
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- bash /tmp/pmc_workload.sh > $OUT/pmc_$C.log 2>&1
  python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/pmc_$C/*counter_collection.csv")
agg = collections.defaultdict(lambda: [0, 0.0])
for path in f:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:70]
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
with open("$OUT/pmc_${C}_summary.csv", "w") as o:
    o.write("kernel,dispatches,sum_$C,avg_$C\n")
    for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f'"{k}",{n},{s},{s/n}\n')
PY
  rm -rf $OUT/pmc_$C
done
ls -la $OUT
