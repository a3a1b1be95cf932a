#!/bin/bash
# Round-3 profiles (GPU box). (1) kernel-trace stats of the headline workload (5000-keypoint cap) and of SuperGlue/20 at the cap: one
# stream, eager launches, --no-roofline / --no-secondary / --no-cpu-baseline so that the CSV holds the workload's launches only.
# (2) separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the hot kernels at their launch shapes at the cap. (3) one SQ pass: matrix-pipe
# busy cycles, CU busy cycles, LDS bank conflicts, instruction-wait cycles of the kernels that ship.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --streams 1 --graphs 0 --no-cpu-baseline --no-secondary --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lg -o lg -- $B --pairs 100 > $OUT/lg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sg -o sg -- $B --matcher superglue --sinkhorn 20 --pairs 64 > $OUT/sg.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
for W in "attention 5000 16" "gemm 163840 256 768" "gemm 163840 512 512" "gemm 163840 512 256" "sinkhorn 5000 16"; do
  TAG=$(echo $W | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W > $OUT/pmc_${TAG}_$C.log 2>&1
  done
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/sq -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py all > $OUT/sq.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob, json, os
out = "gpurun_out/prof_r03"
summary = {}
for d in sorted(glob.glob(out + "/pmc_*_*SIZE")):
    tag = os.path.basename(d)[4:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    summary[tag] = {k: {"dispatches": n, "avg": s / n} for k, (n, s) in agg.items() if "kernel" in k and "elementwise" not in k and "distribution" not in k}
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
for tag, ks in summary.items():
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["avg"])[:3]:
        print(tag, k[:60], v["dispatches"], round(v["avg"]))
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:60] + "|grid" + r.get("Grid_Size", "?")
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
trace = {}
for path in glob.glob(out + "/sq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:60] + "|grid" + r.get("Grid_Size", "?")
        trace.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = sorted({c for v in rows.values() for c in v})
with open(out + "/sq_summary.csv", "w") as o:
    o.write("kernel,dispatches,avg_ns," + ",".join(names) + "\n")
    for k, v in sorted(rows.items()):
        if not any(t in k for t in ("attention", "gemm_dma", "conv3x3", "sinkhorn")): continue
        n = max(len(x) for x in v.values())
        ns = sum(trace.get(k, [0])) / max(1, len(trace.get(k, [0])))
        o.write(f'"{k}",{n},{ns:.0f},' + ",".join(f"{sum(v[c])/max(1,len(v[c])):.0f}" for c in names) + "\n")
print(open(out + "/sq_summary.csv").read())
PY
for f in $OUT/lg/*/*_kernel_stats.csv $OUT/lg/*_kernel_stats.csv $OUT/sg/*/*_kernel_stats.csv $OUT/sg/*_kernel_stats.csv; do [ -f $f ] && { echo "== $f"; head -14 $f | cut -c1-160; }; done
rm -rf $OUT/pmc_*_*SIZE $OUT/sq
tail -2 $OUT/lg.log | cut -c1-400
