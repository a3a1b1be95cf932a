#!/bin/bash
# Round-3 final evidence (GPU box): GPU tests, smoke, the driver's bench command, BASELINE config 4 on one GPU (scene mode), config 2
# (SuperPoint only), SuperGlue headline variants; everything lands in gpurun_out/final_r03 and is copied to profiles/ by hand.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r03
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.log 2>&1
python bench.py --matcher superglue --sinkhorn 20 --steps 2 --warmup 1 --no-secondary > $OUT/bench_superglue20_cap5000.log 2>&1
python bench.py --mode scene --keypoints 2048 --steps 1 --warmup 1 --no-secondary --no-cpu-baseline > $OUT/bench_scene_config4_top2048.log 2>&1
python bench.py --matcher none --images 256 --height 480 --width 640 --steps 3 --warmup 1 --no-secondary > $OUT/bench_config2_superpoint_480x640.log 2>&1
for f in gpu_tests.txt smoke.txt; do echo "== $f"; tail -3 $OUT/$f; done
for f in bench_default bench_superglue20_cap5000 bench_scene_config4_top2048 bench_config2_superpoint_480x640; do echo "== $f"; grep "^{" $OUT/$f.log | cut -c1-420; done
grep real $OUT/bench_default.log
