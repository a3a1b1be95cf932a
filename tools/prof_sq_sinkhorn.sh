#!/bin/bash
# SQ counters of one Sinkhorn iteration at the cap (what bounds the sweep at 0.61 of the HBM roof) -- appended to the round's SQ summary.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r05_sk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/sq_sinkhorn_5000_16 -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py sinkhorn 5000 16 > $OUT/sq.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
out = "gpurun_out/prof_r05_sk"
rows, trace = collections.defaultdict(lambda: collections.defaultdict(list)), {}
for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for path in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        trace.setdefault(r["Kernel_Name"].split("(")[0][:70], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/sq_sinkhorn.csv", "w") as o:
    for k, v in sorted(rows.items()):
        if "sinkhorn" not in k and "fill" not in k: continue
        ns = sum(trace.get(k, [0])) / max(1, len(trace.get(k, [0])))
        vals = {c: sum(x) / len(x) for c, x in v.items()}
        gui = vals.get("GRBM_GUI_ACTIVE", 0) / 8
        wave = max(1.0, vals.get("SQ_WAVE_CYCLES", 0))
        line = (f'"{k}",avg_ns={ns:.0f},clock_GHz={gui / max(ns, 1):.3f},valu_issue_share_of_wave_cycles={vals.get("SQ_ACTIVE_INST_VALU", 0) / wave:.3f},'
                f'wait_any_share={vals.get("SQ_WAIT_ANY", 0) / wave:.3f},wait_inst_share={vals.get("SQ_WAIT_INST_ANY", 0) / wave:.3f},busy_cu_share={vals.get("SQ_BUSY_CU_CYCLES", 0) / max(1.0, 256 * gui):.3f},'
                + ",".join(f"{c}={x:.0f}" for c, x in sorted(vals.items())))
        o.write(line + "\n"); print("SQ", line[:420])
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
