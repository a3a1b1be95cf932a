"""Developer diagnostic (GPU box): SuperGlue batch timing (32 pairs, N = 2048, 100 Sinkhorn iterations)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from gtsfm_amd.utils import synthetic
from gtsfm_amd.runtime.matcher_engine import SuperGlueEngine
dev = torch.device("cuda:0"); T = torch.from_numpy
eng = SuperGlueEngine(synthetic.synthetic_superglue_state_dict())
n, P, iters = 2048, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 100
k0, s0, d0, k1, s1, d1, gt = synthetic.synthetic_pair_features(n, n, (1024, 1024), (1024, 1024), seed=1)
kp = T(np.concatenate([k0, k1] * P)).to(dev); sc = T(np.concatenate([s0, s1] * P)).to(dev); de = T(np.concatenate([d0, d1] * P)).to(dev)
args = (kp, sc, de, [n] * P, [n] * P, [[1024, 1024, 1024, 1024]] * P, iters)
for _ in range(3): out = eng.match_batch(*args)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): out = eng.match_batch(*args)
torch.cuda.synchronize(); dt = (time.time() - t0) / 3
print(f"superglue N={n} P={P} iters={iters}: {dt*1e3:.2f} ms/batch = {P/dt:.1f} pairs/s")
