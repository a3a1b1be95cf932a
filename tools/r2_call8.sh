#!/bin/bash
set -u
OUT=gpurun_out/r2c8
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
bash tools/prof_r02.sh 2>&1 | tail -60
