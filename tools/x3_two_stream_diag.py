"""Diagnostic for tests/test_attention_bf16x3_gpu.py::test_bf16x3_two_stream_pipeline_is_deterministic: where and by how much a repetition differs.
    GTSFM_ATTENTION_MATH=bf16x3|f16x2 python tools/x3_two_stream_diag.py [reps] [streams]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gtsfm_amd.runtime import matcher_engine as ME  # noqa: E402
from gtsfm_amd.runtime.pipeline import FrontEndPipeline  # noqa: E402
from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine  # noqa: E402
from gtsfm_amd.utils import synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), dev)
eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=3), dev)
views = synthetic.synthetic_overlapping_views(5, 192, 256, seed=31)
pairs = [(0, 1), (0, 2), (1, 2), (2, 3), (0, 3), (3, 4)]
pipe = FrontEndPipeline(det, eng, max_keypoints=256, pair_chunk=2, num_streams=streams, use_graphs=False, share_first_layer=False)
feats = pipe.detect(torch.from_numpy(views).to(dev))
print("counts", feats["count"].tolist() if "count" in feats else None)
ref = pipe.match(feats, pairs, [(192, 256)] * 5)
torch.cuda.synchronize()
bad = 0
for it in range(reps):
    out = pipe.match(feats, pairs, [(192, 256)] * 5)
    torch.cuda.synchronize()
    for ci, (x, y) in enumerate(zip(ref, out)):
        a, b = x["mscores"].cpu().numpy(), y["mscores"].cpu().numpy()
        if not np.array_equal(a, b):
            bad += 1
            idx = np.flatnonzero(a != b)
            rows = np.cumsum([0] + [int(v) for pr in zip(x["n0"], x["n1"]) for v in pr])
            print(f"rep {it} chunk {ci}: {len(idx)} of {len(a)} scores differ; first {idx[:6]}; a {a[idx[:4]]} b {b[idx[:4]]}; max |d| {np.abs(a[idx] - b[idx]).max():.3e}; set boundaries {rows}; "
                  f"matches equal {np.array_equal(x['matches'].cpu().numpy(), y['matches'].cpu().numpy())}")
print("repetitions with a difference:", bad, "of", reps * len(ref))
