#!/bin/bash
# Round-4 final evidence (GPU box): GPU tests, smoke, the driver's bench command (BASELINE config 3 as written, configs 2 / 4 and the opt-in
# arithmetic as secondary legs), BASELINE config 4 as ONE scene on one GPU at the 5000-keypoint cap (scene mode, SuperGlue / 100 iterations);
# everything lands in gpurun_out/final_r04 and is copied to profiles/ by hand.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r04
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.log 2>&1
python bench.py --mode scene --steps 1 --warmup 1 --no-secondary --no-cpu-baseline > $OUT/bench_scene_config4_cap5000.log 2>&1
for f in gpu_tests.txt smoke.txt; do echo "== $f"; tail -3 $OUT/$f; done
for f in bench_default bench_scene_config4_cap5000; do echo "== $f"; grep "^{" $OUT/$f.log | cut -c1-420; done
grep real $OUT/bench_default.log
