"""Developer diagnostic (GPU box): LightGlue end-to-end vs the oracle (early stop / pruning variants) + timing."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from gtsfm_amd.utils import synthetic
from gtsfm_amd.runtime.matcher_engine import LightGlueEngine
from oracle import lightglue_oracle as lgo
torch.set_num_threads(16)
dev = torch.device("cuda:0")
T = torch.from_numpy
cases = [
    (dict(), 300, 280, None), (dict(), 300, 280, -1),
    (dict(conf_bias=2.0, conf_gain=4.0), 300, 280, None),
    (dict(conf_bias=1.0, conf_gain=6.0, match_bias=-2.0, match_gain=8.0), 300, 280, -1),
    (dict(conf_bias=1.0, conf_gain=6.0, match_bias=-2.0, match_gain=8.0), 700, 650, 256),
    (dict(), 1, 3, None), (dict(), 129, 128, None),
]
for kw, n0, n1, pth in cases:
    sd = synthetic.synthetic_lightglue_state_dict(**kw)
    eng = LightGlueEngine(sd)
    k0, s0, d0, k1, s1, d1, gt = synthetic.synthetic_pair_features(n0, n1, (480, 640), (480, 640), seed=3)
    res = eng.match_pair(k0, d0, k1, d1, (480, 640), (480, 640), pruning_threshold=pth, return_sim=True)
    with torch.no_grad():
        ora = lgo.lightglue_forward(sd, T(k0)[None], T(k1)[None], T(d0)[None], T(d1)[None], (480, 640), (480, 640), pruning_threshold=pth, return_intermediates=True)
    m0 = ora["matches0"][0].numpy(); m1 = ora["matches1"][0].numpy()
    kept = (ora["ind0"].shape[1], ora["ind1"].shape[1])
    simerr = float("nan")
    if res["sim"][: kept[0], : kept[1]].shape == tuple(ora["sim"][0].shape):
        simerr = np.abs(res["sim"][: kept[0], : kept[1]] - ora["sim"][0].numpy()).max()
    print(f"lightglue {kw} n=({n0},{n1}) prune={pth}: stop hip={res['stop']} oracle={ora['stop']}; kept hip={res['kept'].tolist()} oracle={kept}; "
          f"matches0 equal {np.array_equal(res['matches0'], m0)} ({int((m0>-1).sum())}), matches1 equal {np.array_equal(res['matches1'], m1)}, "
          f"list equal {np.array_equal(res['matches'], ora['matches'].numpy())}; mscores0 err {np.abs(res['matching_scores0']-ora['matching_scores0'][0].numpy()).max():.2e}; sim err {simerr:.2e}")

sd = synthetic.synthetic_lightglue_state_dict()
eng = LightGlueEngine(sd)
for (n, P, kw) in [(1024, 1, {}), (2048, 1, {}), (2048, 4, {}), (1024, 16, {}), (2048, 8, {}), (2048, 8, dict(depth_confidence=-1, pruning_threshold=None))]:
    k0, s0, d0, k1, s1, d1, gt = synthetic.synthetic_pair_features(n, n, (1024, 1024), (1024, 1024), seed=1)
    kp = T(np.concatenate([k0, k1] * P)).to(dev); de = T(np.concatenate([d0, d1] * P)).to(dev)
    args = (kp, de, [n] * P, [n] * P, [[1024, 1024, 1024, 1024]] * P)
    for _ in range(2): out = eng.match_batch(*args, **kw)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): out = eng.match_batch(*args, **kw)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    layers = out["stop"].float().mean().item()
    fl = P * (layers * 2 * (2490368 * n + 1792 * n * n) + 262144 * n + 512 * n * n)
    print(f"lightglue N={n} P={P} {kw}: {dt*1e3:.2f} ms/batch = {P/dt:.1f} pairs/s ({fl/dt/1e12:.1f} TFLOP/s at {layers:.1f} layers); kept {out['kept'][:2].tolist()}")
