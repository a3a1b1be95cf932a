#!/bin/bash
# Round-4 profiles (GPU box). (1) kernel-trace stats of the headline workload (BASELINE config 3 as written: 1000 pairs / 46 views at the
# 5000-keypoint cap; one stream, eager launches, the workload's launches only) and of the same pair list's first 250 pairs under
# GTSFM_ATTENTION_MATH=bf16x3. (2) separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the conv stack (8 launches of a batch-8 forward at
# 1024x1024, the fused first layer in its shipped form) and of the bf16x3 attention at the cap's launch shape. (3) SQ counters of both
# attention arithmetics in one pass each (matrix-pipe busy cycles, wave cycles, GRBM_GUI_ACTIVE for the effective clock).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --streams 1 --graphs 0 --no-cpu-baseline --no-secondary --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lg -o lg -- $B > $OUT/lg.log 2>&1
GTSFM_ATTENTION_MATH=bf16x3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x3 -o x3 -- $B --pairs 250 > $OUT/x3.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
for W in "conv" "attention_x3 5000 16"; do
  TAG=$(echo $W | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W > $OUT/pmc_${TAG}_$C.log 2>&1
  done
done
for W in "attention 5000 16" "attention_x3 5000 16"; do
  TAG=$(echo $W | tr ' ' '_')
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/sq_$TAG -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W > $OUT/sq_$TAG.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob, json, os
out = "gpurun_out/prof_r04"
summary = {}
for d in sorted(glob.glob(out + "/pmc_*_*SIZE")):
    tag = os.path.basename(d)[4:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0][:70] + "|grid" + r.get("Grid_Size", "?")
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    summary[tag] = {k: {"dispatches": n, "avg": s / n, "sum": s} for k, (n, s) in agg.items() if ("kernel" in k) and "elementwise" not in k and "distribution" not in k}
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)
for tag, ks in summary.items():
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["sum"])[:10]:
        print("PMC", tag, k[:80], v["dispatches"], round(v["avg"]))
with open(out + "/sq_summary.csv", "w") as o:
    for d in sorted(glob.glob(out + "/sq_*")):
        if not os.path.isdir(d): continue
        rows = collections.defaultdict(lambda: collections.defaultdict(list)); trace = {}
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                rows[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                trace.setdefault(r["Kernel_Name"].split("(")[0][:60], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in sorted(rows.items()):
            if "attention" not in k: continue
            ns = sum(trace.get(k, [0])) / max(1, len(trace.get(k, [0])))
            vals = {c: sum(x) / len(x) for c, x in v.items()}
            gui = vals.get("GRBM_GUI_ACTIVE", 0) / 8
            line = f'"{k}",avg_ns={ns:.0f},clock_GHz={gui / max(ns, 1):.3f},mfma_util={vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, 1024 * gui):.3f},' + ",".join(f"{c}={x:.0f}" for c, x in sorted(vals.items()))
            o.write(line + "\n"); print("SQ", line)
PY
for f in $OUT/lg/*kernel_stats.csv $OUT/x3/*kernel_stats.csv; do [ -f $f ] && { echo "== $f"; head -12 $f | cut -c1-150; }; done
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/lg.log | cut -c1-300; tail -1 $OUT/x3.log | cut -c1-300
