#!/bin/bash
# Round-2 profiles (GPU box): kernel-trace stats of the two bench workloads (one stream, eager launches, so that per-kernel
# durations are the isolated ones) and separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the hot kernels at workload shape.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lg -o lg -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --streams 1 --graphs 0 --no-cpu-baseline --no-secondary > $OUT/lg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sg -o sg -- python $GRAFT_REPO_ROOT/bench.py --matcher superglue --sinkhorn 100 --pairs 256 --steps 1 --warmup 1 --streams 1 --graphs 0 --no-cpu-baseline --no-secondary > $OUT/sg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vf -o vf -- python $GRAFT_REPO_ROOT/tools/prof_verifier.py > $OUT/vf.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
for W in "attention" "gemm 256 768" "gemm 512 512" "gemm 512 256" "sinkhorn"; do
  TAG=$(echo $W | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W > $OUT/pmc_${TAG}_$C.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob, json, os
out = "gpurun_out/prof_r02"
summary = {}
for d in sorted(glob.glob(out + "/pmc_*_*SIZE")):
    tag = os.path.basename(d)[4:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"][:60]
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    summary[tag] = {k: {"dispatches": n, "avg": s / n} for k, (n, s) in agg.items() if "gtsfm" in k or "kernel" in k}
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
for tag, ks in summary.items():
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["avg"])[:4]:
        print(tag, k[:50], v["dispatches"], round(v["avg"]))
PY
for f in $OUT/lg/*/*_kernel_stats.csv $OUT/lg/*_kernel_stats.csv $OUT/sg/*/*_kernel_stats.csv $OUT/sg/*_kernel_stats.csv $OUT/vf/*/*_kernel_stats.csv $OUT/vf/*_kernel_stats.csv; do [ -f $f ] && { echo "== $f"; head -12 $f | cut -c1-150; }; done
rm -rf $OUT/pmc_*_*SIZE
