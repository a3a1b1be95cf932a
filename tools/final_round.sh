#!/bin/bash
# Round-5 final evidence (GPU box), everything on the tree as it is: the whole -m gpu suite, smoke, the driver's bench command (BASELINE config 3 as
# written; configs 2 / 4 and the opt-in arithmetic as secondary legs), BASELINE config 4 as ONE scene through ShardedDetDescCorrespondenceGenerator
# (scene mode, SuperGlue / 100 iterations at the cap) plain and under a one-rank RCCL group, the profiles of tools/prof_r05.sh (kernel trace of the
# headline, HBM-side traffic and SQ counters). Everything lands in gpurun_out/final_r05 (+ prof_r05) and is copied to profiles/ by hand.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $OUT/gpu_tests.txt
python -m pytest tests/test_lightglue_hf_golden_gpu.py tests/test_lightglue_fp64_arbiter_gpu.py -m gpu -q -s -k "cap or float64" 2>&1 | grep "HFCAP\|ARBITER\|passed\|failed" > $OUT/lightglue_pins.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.log 2>&1
python bench.py --mode scene --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --dump-matches 1 > $OUT/bench_scene_config4_cap5000.log 2>&1
GTSFM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --mode scene --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --dump-matches 1 > $OUT/bench_scene_config4_cap5000_rccl_one_rank.log 2>&1
for f in gpu_tests.txt smoke.txt; do echo "== $f"; tail -3 $OUT/$f; done
for f in bench_default bench_scene_config4_cap5000 bench_scene_config4_cap5000_rccl_one_rank; do echo "== $f"; grep "^{" $OUT/$f.log | cut -c1-420; done
grep real $OUT/bench_default.log
python tools/bench_assign.py > $OUT/bench_assign.txt 2>&1
python tools/bench_sweeps.py 1024 2048 5000:16 5000 5000:1 > $OUT/bench_sweeps.txt 2>&1
grep -v "waves=8" $OUT/bench_assign.txt | cut -c1-200; cut -c1-260 $OUT/bench_sweeps.txt
bash tools/prof_r05.sh > $OUT/prof.log 2>&1
grep "^TRAFFIC\|^SQ" $OUT/prof.log | cut -c1-260
