set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sp -o sp -- python $GRAFT_REPO_ROOT/bench.py --matcher none --images 256 --height 480 --width 640 --steps 2 --warmup 1 --no-secondary --no-roofline --no-cpu-baseline > $OUT/sp.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
head -12 $OUT/sp/sp_kernel_stats.csv | cut -c1-140
tail -1 $OUT/sp.log | cut -c1-200
