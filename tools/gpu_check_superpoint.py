"""Developer diagnostic (GPU box): stage-by-stage comparison of the SuperPoint HIP kernels with the oracle."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch.nn.functional as F
from gtsfm_amd.utils import synthetic
from gtsfm_amd.runtime import lib as L
from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine
from oracle import superpoint_oracle as spo

torch.set_num_threads(8)
dev = torch.device("cuda:0")
lib = L.load()
stream = lambda: torch.cuda.current_stream().cuda_stream
sd = synthetic.synthetic_superpoint_state_dict()

def pack_conv(w):
    cout, cin = w.shape[:2]
    out = np.empty(lib.gtsfm_packed_conv3x3_floats(cin, cout), np.float32)
    wc = np.ascontiguousarray(w.numpy())
    L.check(lib.gtsfm_pack_conv3x3(wc.ctypes.data, cin, cout, out.ctypes.data), "pack")
    return torch.from_numpy(out).to(dev)

def padbias(b):
    n = (b.numel() + 63) // 64 * 64
    o = torch.zeros(n); o[: b.numel()] = b
    return o.to(dev)

# --- conv3x3 stage test
for (name, H, W, pool) in [("conv1b", 37, 53, 0), ("conv1b", 37, 53, 1), ("conv3b", 24, 40, 1), ("convPa", 16, 16, 0)]:
    w, b = sd[f"{name}.weight"], sd[f"{name}.bias"]
    cout, cin = w.shape[:2]
    x = torch.randn(2, cin, H, W)
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    if pool: ref = F.max_pool2d(ref, 2, 2)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.full((2, Ho, Wo, cout), float("nan"), device=dev)
    L.check(lib.gtsfm_conv3x3_f32(xd.data_ptr(), cin, 0, out.data_ptr(), cout, 0, pack_conv(w).data_ptr(), padbias(b).data_ptr(), 2, H, W, cin, cout, 1, pool, stream()), "conv")
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2)
    print(f"conv3x3 {name} {H}x{W} pool={pool}: max abs err {float((got-ref).abs().max()):.3e} (ref max {float(ref.abs().max()):.3f}) nan={bool(torch.isnan(got).any())}")

# --- end to end
eng = SuperPointEngine(sd)
for (H, W, seed) in [(120, 160, 1), (123, 157, 2), (240, 320, 3), (480, 640, 4)]:
    gray = synthetic.synthetic_gray_image(H, W, seed)
    img = torch.from_numpy(gray).to(dev)[None]
    out = eng.forward(img, return_score_maps=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        ora = spo.superpoint_forward(sd, spo.gray_u8_to_tensor(gray), return_intermediates=True)
    k = int(out["count"][0]); kraw = int(out["count_raw"][0])
    ds = out["dense_scores"][0].cpu(); nm = out["nms_scores"][0].cpu()
    print(f"e2e {H}x{W}: K hip={k} raw={kraw} oracle={ora['keypoints'].shape[0]}; dense score max err {float((ds-ora['dense_scores'][0]).abs().max()):.3e}; nms map mismatches {int(((nm>0)!=(ora['nms_scores'][0]>0)).sum())}")
    xy = out["xy"][0, :k].cpu(); sc = out["scores"][0, :k].cpu(); de = out["descriptors"][0, :k].cpu()
    if k == ora["keypoints"].shape[0] and torch.equal(xy, ora["keypoints"]):
        print(f"   keypoints identical; score max err {float((sc-ora['scores']).abs().max()):.3e}; desc max err {float((de-ora['descriptors'].T).abs().max()):.3e}")
    else:
        a = set(map(tuple, xy.numpy().astype(int).tolist())); b_ = set(map(tuple, ora["keypoints"].numpy().astype(int).tolist()))
        print(f"   keypoint sets differ: only hip {len(a-b_)}, only oracle {len(b_-a)}")
    # nms kernel fed with the oracle's dense scores must be bit-exact
    dso = ora["dense_scores"].contiguous().to(dev)
    h8, w8 = dso.shape[1:]
    scratch = torch.empty(lib.gtsfm_sp_nms_scratch_bytes(1, h8, w8), dtype=torch.uint8, device=dev)
    nout = torch.empty_like(dso)
    L.check(lib.gtsfm_sp_simple_nms(dso.data_ptr(), 1, h8, w8, 4, scratch.data_ptr(), nout.data_ptr(), stream()), "nms")
    torch.cuda.synchronize()
    print(f"   nms(oracle dense) bit-exact: {bool(torch.equal(nout.cpu(), ora['nms_scores']))}")

# --- timing at 1024x1024
for B in (1, 4):
    img = torch.from_numpy(np.stack([synthetic.synthetic_gray_image(1024, 1024, s) for s in range(B)])).to(dev)
    for _ in range(2): out = eng.forward(img, capacity=20000)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): out = eng.forward(img, capacity=20000)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"1024x1024 batch {B}: {dt*1e3:.2f} ms/step = {B/dt:.1f} img/s ({177.85*B/dt/1e3:.1f} TFLOP/s); K={out['count'].tolist()} raw={out['count_raw'].tolist()}")
