#!/bin/bash
# SQ / GRBM counters of the three MFMA kernels at their workload shapes (one PMC pass, kernel trace only):
# matrix-pipe busy cycles, wave stall buckets, LDS bank conflicts, and GRBM_GUI_ACTIVE (-> effective clock).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/raw -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py > $OUT/sq.log 2>&1
python - <<PY
import csv, collections, glob
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:48] + "|grid" + r.get("Grid_Size", "?")
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
trace = {}
for path in glob.glob("$OUT/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:48] + "|grid" + r.get("Grid_Size", "?")
        trace.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$OUT/sq_summary.csv", "w") as o:
    names = sorted({c for v in rows.values() for c in v})
    o.write("kernel,dispatches,avg_ns," + ",".join(names) + "\n")
    for k, v in rows.items():
        if "mfma" not in k: continue
        n = max(len(x) for x in v.values())
        ns = sum(trace.get(k, [0])) / max(1, len(trace.get(k, [0])))
        o.write(f'"{k}",{n},{ns:.0f},' + ",".join(f"{sum(v[c])/max(1,len(v[c])):.0f}" for c in names) + "\n")
print(open("$OUT/sq_summary.csv").read())
PY
rm -rf $OUT/raw
tail -3 $OUT/sq.log
