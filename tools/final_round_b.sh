#!/bin/bash
# Round-5 final evidence, second pass (after the sweep kernels' buffer-resource loads): counters first -- the traffic file they produce is put
# where bench.py reads it BEFORE the driver's bench command runs, so the line cites this tree's counters --, then the whole -m gpu suite, smoke,
# the driver's bench command, BASELINE config 4 as one scene (SuperGlue / 100 iterations at the cap), the sweeps' micro-benchmarks.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r05c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/prof_r05.sh > $OUT/prof.log 2>&1
grep "^TRAFFIC\|^SQ" $OUT/prof.log | cut -c1-200
cp gpurun_out/prof_r05/r05_pmc_traffic.json profiles/r05_pmc_traffic.json
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $OUT/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.log 2>&1
python bench.py --mode scene --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --dump-matches 1 > $OUT/bench_scene_config4_cap5000.log 2>&1
python tools/bench_assign.py > $OUT/bench_assign.txt 2>&1
python tools/bench_sweeps.py 1024 2048 5000:16 5000 5000:1 > $OUT/bench_sweeps.txt 2>&1
for f in gpu_tests.txt smoke.txt; do echo "== $f"; tail -3 $OUT/$f; done
for f in bench_default bench_scene_config4_cap5000; do echo "== $f"; grep "^{" $OUT/$f.log | cut -c1-300; done
grep real $OUT/bench_default.log
grep -v "waves=8" $OUT/bench_assign.txt | cut -c1-200; cut -c1-260 $OUT/bench_sweeps.txt
