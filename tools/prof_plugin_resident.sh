#!/bin/bash
# rocprofv3 --kernel-trace --stats of the per-call plugin path with ONE thread and resident images (tools/bench_plugin_resident.py): a single pair's
# kernels only. Prints the per-kernel averages and GPU-busy time per call (sum of kernel durations / calls) beside the wall time per call.
set -u
K=${1:-5000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_plugin_resident
mkdir -p $OUT
python $GRAFT_REPO_ROOT/tools/bench_plugin_resident.py --keypoints $K | tee $OUT/unprofiled_k$K.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k$K -o p -- python $GRAFT_REPO_ROOT/tools/bench_plugin_resident.py --keypoints $K --calls 24 > $OUT/k$K.log 2>&1
find $OUT/k$K -name "*kernel_trace.csv" -delete
f=$(find $OUT/k$K -name "*kernel_stats.csv" | head -1)
cp $f $OUT/plugin_resident_kernel_stats_k$K.csv
head -14 $f | cut -c1-150
grep keypoints $OUT/k$K.log
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
matcher = [r for r in rows if not any(s in r["Name"] for s in ("conv3x3", "softmax_d2s", "nms_", "kp_", "sample_desc", "gemm_mfma"))]
total = sum(float(r["TotalDurationNs"]) for r in matcher)
calls = 24 + 6  # timed calls + the 6 that made the images resident (same kernels)
launches = sum(int(r["Calls"]) for r in matcher)
print(f"matcher kernels: {total / calls / 1e6:.3f} ms of GPU time and {launches / calls:.0f} launches per match() call")
PY
