#!/bin/bash
# Round-6 profiles (GPU box), all on the tree as it is:
# (1) rocprofv3 --kernel-trace --stats of the headline workload (BASELINE config 3 as written: 1000 pairs / 46 views at the 5000-keypoint cap; one
#     stream, eager launches, the workload's launches only);
# (2) HBM-side traffic, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (guide: they do not fit one pass), of every kernel bench.py prints a
#     `traffic` figure for, at the headline's launch shapes (tools/pmc_kernels.py all) and of the bf16x3 attention; summarised by
#     tools/pmc_summarise.py into gpurun_out/prof_r06/r06_pmc_traffic.json in the form bench.py reads (copied to profiles/ by hand);
# (3) SQ counters of the match extraction (the kernel furthest below its roof) and of the exact attention: VALU / wait split, effective clock.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --streams 1 --graphs 0 --no-cpu-baseline --no-secondary --no-roofline"
if [ "${PROF_TRACE:-1}" = "1" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lg -o lg -- $B > $OUT/lg.log 2>&1
  find $OUT -name "*kernel_trace.csv" -delete
fi
for W in "all" "gemm 163840 256 512" "attention_x3 5000 16" "attention_f16x2 5000 16"; do
  TAG=$(echo $W | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W > $OUT/pmc_${TAG}_$C.log 2>&1
  done
done
for W in "lg_assign 5000 16" "attention 5000 16" "attention_f16x2 5000 16"; do
  TAG=$(echo $W | tr ' ' '_')
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/sq_$TAG -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $W > $OUT/sq_$TAG.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summarise.py $OUT r06
for f in $OUT/lg/*kernel_stats.csv; do [ -f $f ] && { echo "== $f"; head -14 $f | cut -c1-160; }; done
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
[ -f $OUT/lg.log ] && tail -1 $OUT/lg.log | cut -c1-300
