"""Developer diagnostic (GPU box): attention kernel and SuperGlue end-to-end vs the oracle."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from gtsfm_amd.utils import synthetic
from gtsfm_amd.runtime import lib as L
from gtsfm_amd.runtime.matcher_engine import SuperGlueEngine
from oracle import superglue_oracle as sgo
torch.set_num_threads(16)
dev = torch.device("cuda:0"); lib = L.load()
stream = lambda: torch.cuda.current_stream().cuda_stream

# attention
for (nq, nk) in [(100, 70), (128, 64), (257, 300), (1, 1), (513, 2049)]:
    q = torch.randn(nq, 256) ; k = torch.randn(nk, 256); v = torch.randn(nk, 256)
    qh = q.view(nq, 4, 64).transpose(0, 1); kh = k.view(nk, 4, 64).transpose(0, 1); vh = v.view(nk, 4, 64).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(1, 2) * 0.125, -1) @ vh).transpose(0, 1).reshape(nq, 256)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    out = torch.full((nq, 256), float("nan"), device=dev)
    probs = torch.tensor([[0, 0, 0, 1]], dtype=torch.int32, device=dev); counts = torch.tensor([nq, nk], dtype=torch.int32, device=dev)
    L.check(lib.gtsfm_attention_f32(qd.data_ptr(), 256, kd.data_ptr(), 256, vd.data_ptr(), 256, out.data_ptr(), 256, probs.data_ptr(), counts.data_ptr(), 1, nq, 4, 0.125, stream()), "attn")
    torch.cuda.synchronize()
    print(f"attention nq={nq} nk={nk}: max err {float((out.cpu()-ref).abs().max()):.3e} nan={bool(torch.isnan(out).any())}")

sd = synthetic.synthetic_superglue_state_dict()
eng = SuperGlueEngine(sd)
T = torch.from_numpy
for (n0, n1, shp0, shp1, seed, iters) in [(96, 80, (240, 320), (200, 300), 11, 20), (257, 300, (480, 640), (480, 640), 12, 100), (1, 5, (64, 64), (64, 64), 13, 20), (1024, 900, (1024, 1024), (1024, 1024), 14, 20)]:
    k0, s0, d0, k1, s1, d1, gt = synthetic.synthetic_pair_features(n0, n1, shp0, shp1, seed=seed)
    res = eng.match_pair(k0, s0, d0, k1, s1, d1, shp0, shp1, sinkhorn_iterations=iters, return_ot=True)
    with torch.no_grad():
        ora = sgo.superglue_forward(sd, T(k0)[None], T(k1)[None], T(s0)[None], T(s1)[None], T(d0).T[None].contiguous(), T(d1).T[None].contiguous(), shp0, shp1, sinkhorn_iterations=iters, return_intermediates=True)
    m0 = ora["matches0"][0].numpy(); m1 = ora["matches1"][0].numpy()
    print(f"superglue n=({n0},{n1}) it={iters}: matches0 equal {np.array_equal(res['matches0'], m0)} ({int((m0>-1).sum())} matches), matches1 equal {np.array_equal(res['matches1'], m1)}; "
          f"mscores0 err {np.abs(res['matching_scores0']-ora['matching_scores0'][0].numpy()).max():.2e}; OT max err {np.abs(res['ot']-ora['ot'][0].numpy()).max():.2e}")

# timing
for (n, P, iters) in [(1024, 1, 20), (2048, 1, 20), (2048, 1, 100), (2048, 4, 20), (1024, 16, 20)]:
    k0, s0, d0, k1, s1, d1, gt = synthetic.synthetic_pair_features(n, n, (1024, 1024), (1024, 1024), seed=1)
    kp = T(np.concatenate([k0, k1] * P)).to(dev); sc = T(np.concatenate([s0, s1] * P)).to(dev); de = T(np.concatenate([d0, d1] * P)).to(dev)
    args = (kp, sc, de, [n] * P, [n] * P, [[1024, 1024, 1024, 1024]] * P, iters)
    for _ in range(2): eng.match_batch(*args)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): eng.match_batch(*args)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    fl = P * (2*217280*n + 36*(1310720*n + 1024*n*n) + 262144*n + 512*n*n)
    print(f"superglue N={n} P={P} iters={iters}: {dt*1e3:.2f} ms/batch = {P/dt:.1f} pairs/s ({fl/dt/1e12:.1f} TFLOP/s)")
