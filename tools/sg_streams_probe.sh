#!/bin/bash
# Developer probe: SuperGlue / 100 Sinkhorn iterations at the cap (256 pairs) by number of HIP streams and arithmetic -- how much of the HBM-bound
# Sinkhorn overlaps the matrix-bound graph network of the other streams' chunks.   [GTSFM_LIB=variant.so] tools/sg_streams_probe.sh "f32 f16x2" "1 2 3"
for arith in ${1:-f32 f16x2}; do for st in ${2:-1 2 3 4}; do echo -n "arith=$arith streams=$st: "; python bench.py --matcher superglue --sinkhorn 100 --pairs 256 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --streams $st --arithmetic $arith --details-file gpurun_out/r06c/tmp_details.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('pair_chunk'))"; done; done
