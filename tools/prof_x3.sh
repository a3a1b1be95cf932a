#!/bin/bash
# Round-4 profile of the bf16x3 attention (GPU box): kernel-trace stats (split pass vs main kernel) and one SQ counter pass.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_x3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py attention_x3 ${1:-5000} ${2:-16} > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/sq -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py attention_x3 ${1:-5000} ${2:-16} > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/sq2 -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py attention_x3 ${1:-5000} ${2:-16} > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob
out = "gpurun_out/prof_x3"
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
for sub in ("sq", "sq2"):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(out + f"/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in rows.items():
        if "attention" in k:
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
tail -3 $OUT/sq.log $OUT/sq2.log | cut -c1-300
