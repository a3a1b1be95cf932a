#!/bin/bash
# SQ / GRBM counters of the attention launch (16 pairs at the cap) in the three arithmetics: matrix-pipe busy share, effective clock, VALU issue share, waits.
#   gpurun -- tools/prof_sq_attention_math.sh [tag]     -> gpurun_out/prof_sq_attention/sq_attention_math_<tag>.txt
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sq_attention
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in attention attention_x3 attention_f16x2; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/raw_$W -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W 5000 16 > $OUT/$W.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d $OUT/raw2_$W -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W 5000 16 > $OUT/${W}_2.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/sq_attention_math_$TAG.txt
import csv, glob, collections
out = "$OUT"
for W in ("attention", "attention_x3", "attention_f16x2"):
    dur, agg = {}, collections.defaultdict(lambda: collections.defaultdict(list))
    for raw in ("raw_", "raw2_"):
        for path in glob.glob(f"{out}/{raw}{W}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                dur[(raw, r["Dispatch_Id"])] = (r["Kernel_Name"].split("(")[0][:60], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for path in glob.glob(f"{out}/{raw}{W}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                key = (raw, r["Dispatch_Id"])
                if key in dur and "attention" in dur[key][0]:
                    agg[dur[key][0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    agg[dur[key][0]]["ns_" + raw].append(dur[key][1])
    for k, v in agg.items():
        vals = {c: sum(x) / len(x) for c, x in v.items()}
        gui = vals.get("GRBM_GUI_ACTIVE", 0) / 8
        ns = vals.get("ns_raw_", 1)
        wave = max(1.0, vals.get("SQ_WAVE_CYCLES", 0))
        print(f'{W}: "{k}" avg_ns={ns:.0f} clock_GHz={gui / ns:.3f} mfma_busy_share={vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, 1024 * gui):.3f} '
              f'valu_issue_share_of_wave_cycles={vals.get("SQ_ACTIVE_INST_VALU", 0) / wave:.3f} wait_any_share={vals.get("SQ_WAIT_ANY", 0) / wave:.3f} '
              f'wait_inst_share={vals.get("SQ_WAIT_INST_ANY", 0) / wave:.3f}')
        print("   " + " ".join(f"{c}={x:.0f}" for c, x in sorted(vals.items())))
PY
rm -rf $OUT/raw_* $OUT/raw2_*
