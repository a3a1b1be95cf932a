#!/bin/bash
# Effective shader clock of the hot kernels: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration from the same
# rocprofv3 run (kernel trace + one GRBM counter; MI355X_MICROARCH.md "DVFS give-back").
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/raw -o clk -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py all > $OUT/clk.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
out = "gpurun_out/prof_clock"
dur = {}
for path in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0][:60], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size", "?"))
agg = collections.defaultdict(list)
for path in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur: continue
        name, ns, grid = dur[r["Dispatch_Id"]]
        agg[f"{name}|grid{grid}"].append((float(r["Counter_Value"]), ns))
with open(out + "/clock_summary.csv", "w") as o:
    o.write("kernel|grid,dispatches,avg_ns,GRBM_GUI_ACTIVE,effective_clock_GHz\n")
    for k, v in sorted(agg.items()):
        if not any(t in k for t in ("attention", "gemm_dma", "conv3x3", "sinkhorn_rows")): continue
        v = v[1:] if len(v) > 1 else v  # the first dispatch warms up
        gui, ns = sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v)
        o.write(f'"{k}",{len(v)},{ns:.0f},{gui:.0f},{gui / 8 / ns:.3f}\n')
print(open(out + "/clock_summary.csv").read())
PY
rm -rf $OUT/raw
