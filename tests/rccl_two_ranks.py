"""The front-end's collectives on RCCL, one rank per GPU when the box has as many GPUs as ranks (tools/scale_selfcheck.sh on a multi-GPU node),
otherwise every rank on device 0 (tests/test_rccl_gpu.py on a one-GPU box, launched under torch.distributed.run).
Exit code 77 = RCCL refuses several ranks on one GPU (it normally does: "Duplicate GPU detected"); 0 = the collectives ran and agree."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gtsfm_amd import parallel  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import datetime  # noqa: E402

distinct = torch.cuda.device_count() >= world
index = int(os.environ.get("LOCAL_RANK", rank)) if distinct else 0
torch.cuda.set_device(index)
dev = torch.device("cuda", index)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=120))
    probe = torch.full((4,), float(rank), device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: RCCL refused {world} ranks on {'distinct devices' if distinct else 'one device'}: {str(e)[:200]}", flush=True)
    os._exit(1 if distinct else 77)  # on distinct devices a refusal is a failure, not a skip
blob = torch.arange(1000, dtype=torch.float32, device=dev) if rank == 0 else None
got = parallel.broadcast_packed_weights(blob, 1000, dev)
assert torch.equal(got.cpu(), torch.arange(1000, dtype=torch.float32))
n, k = 5, 16
mine = parallel.partition_images(n, rank, world)
local = {"count": torch.tensor([k - i for i in mine], dtype=torch.int32, device=dev), "xy": torch.stack([torch.full((k, 2), float(i), device=dev) for i in mine]),
         "scores": torch.stack([torch.full((k,), float(i), device=dev) for i in mine]), "descriptors": torch.stack([torch.full((k, 256), float(i), device=dev) for i in mine])}
table = parallel.all_gather_feature_table(local, n)
for i in range(n):
    row = parallel.table_index(i, n, world)
    assert int(table["count"][row]) == k - i and float(table["descriptors"][row, 3, 7]) == float(i)
plan = parallel.ScenePlan(n, parallel.exhaustive_pairs(n), rank, world)  # the sharded generator's exchange: all_to_all_single on RCCL
mine_table = parallel.exchange_feature_rows(plan, local)
for row, i in enumerate(plan.table_images):
    assert int(mine_table["count"][row]) == k - i and float(mine_table["xy"][row, 2, 1]) == float(i) and float(mine_table["descriptors"][row, 3, 7]) == float(i)
pairs = parallel.partition_pairs_2d(parallel.exhaustive_pairs(n), rank, world)
gathered = parallel.gather_matches({p: np.full((p[0] + p[1], 2), p[0], dtype=np.int64) for p in pairs}, dev)
assert sorted(gathered) == parallel.exhaustive_pairs(n) and all(v.shape == (p[0] + p[1], 2) for p, v in gathered.items())
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank}: rccl_two_ranks OK on {dev} ({'one rank per GPU' if distinct else 'ranks share device 0'})", flush=True)
