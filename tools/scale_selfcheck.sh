#!/bin/bash
# Self-check of the multi-GPU path on the box it runs on, in one command: `tools/scale_selfcheck.sh N` with N <= the GPUs visible.
#   1. N >= 2: tests/rccl_two_ranks.py on N ranks, one per GPU -- broadcast, all_gather, the ragged all_to_all_single, the ragged match gather on RCCL;
#   2. `bench.py --gpus N` (replica = BASELINE config 3 per rank, weak scaling), short: the line must parse and carry n_gpus = N;
#   3. `bench.py --mode scene --gpus N --dump-matches 1` (BASELINE config 4's shape, reduced: 12 views / 40 pairs, top-2048, 20 Sinkhorn iterations)
#      against the SAME scene on 1 GPU: the SHA-1 over every pair's match array must be identical -- sharding must not change a single match.
# Prints one PASS / FAIL line per step and exits non-zero on the first failure. Nothing here reads /root/reference.
set -u
N=${1:-1}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
OUT=${SCALE_SELFCHECK_OUT:-gpurun_out/scale_selfcheck}
mkdir -p "$OUT"
fail() { echo "FAIL: $1"; exit 1; }
port() { python -c "import socket; s = socket.socket(); s.bind(('127.0.0.1', 0)); print(s.getsockname()[1])"; }
field() { python -c "import json, sys; line = [l for l in open(sys.argv[1]) if l.startswith('{')][-1]; print(json.loads(line)[sys.argv[2]])" "$1" "$2"; }

if [ "$N" -ge 2 ]; then
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$(port)" tests/rccl_two_ranks.py \
        > "$OUT/rccl_ranks.log" 2>&1 || fail "tests/rccl_two_ranks.py on $N ranks (see $OUT/rccl_ranks.log)"
    [ "$(grep -c 'rccl_two_ranks OK' "$OUT/rccl_ranks.log")" -eq "$N" ] || fail "not every rank reported OK"
    echo "PASS: the front-end's collectives on $N RCCL ranks, one per GPU"
fi

SMALL="--keypoints 2048 --no-secondary --no-cpu-baseline --no-roofline --details-file $OUT/details.json"
timeout 900 python bench.py --gpus "$N" --steps 2 --warmup 1 --pairs 40 $SMALL > "$OUT/replica_n$N.log" 2>&1 || fail "bench.py --gpus $N (replica) -- see $OUT/replica_n$N.log"
[ "$(field "$OUT/replica_n$N.log" n_gpus)" = "$N" ] || fail "the replica line does not carry n_gpus = $N"
echo "PASS: bench.py --gpus $N (replica): $(field "$OUT/replica_n$N.log" value) image-pairs/s"

SCENE="--mode scene --images 12 --pairs 40 --sinkhorn 20 --steps 1 --warmup 1 --dump-matches 1 $SMALL"
timeout 900 python bench.py --gpus 1 $SCENE > "$OUT/scene_n1.log" 2>&1 || fail "bench.py --mode scene --gpus 1 -- see $OUT/scene_n1.log"
want=$(field "$OUT/scene_n1.log" match_digest)
if [ "$N" -ge 2 ]; then
    timeout 900 python bench.py --gpus "$N" $SCENE > "$OUT/scene_n$N.log" 2>&1 || fail "bench.py --mode scene --gpus $N -- see $OUT/scene_n$N.log"
    got=$(field "$OUT/scene_n$N.log" match_digest)
else  # one GPU: the sharded class under a one-rank RCCL process group (init, broadcast, all_to_all, gathers all run) against the plain run
    GTSFM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT="$(port)" timeout 900 python bench.py --gpus 1 $SCENE > "$OUT/scene_n1_rccl.log" 2>&1 \
        || fail "bench.py --mode scene under a one-rank RCCL group -- see $OUT/scene_n1_rccl.log"
    got=$(field "$OUT/scene_n1_rccl.log" match_digest)
fi
[ -n "$want" ] && [ "$want" = "$got" ] || fail "match digests differ: 1 GPU $want, $N rank(s) $got"
# the same scene through replica mode's pipeline (no generator class, no exchange): the digest must not depend on the path either
timeout 900 python bench.py --gpus 1 --matcher superglue --images 12 --pairs 40 --sinkhorn 20 --steps 1 --warmup 1 --dump-matches 1 --share-first-layer 1 $SMALL \
    > "$OUT/replica_same_scene.log" 2>&1 || fail "replica run of the same scene -- see $OUT/replica_same_scene.log"
rep=$(field "$OUT/replica_same_scene.log" match_digest)
[ "$rep" = "$want" ] || fail "match digests differ between --mode scene ($want) and replica ($rep) on the same scene"
echo "PASS: --mode scene on $N rank(s) == 1 GPU == replica pipeline, match digest $want"
