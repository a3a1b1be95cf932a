"""GPU parity tests of the SuperPoint HIP path against the oracle and the golden vectors, through the C ABI.

Tolerances (BASELINE.json north_star): keypoints / indices bit-exact; descriptors and scores within 1e-4 fp32
(observed: ~3e-7 descriptors, ~6e-6 scores). Comparison-only kernels (simple-NMS, keypoint extraction) are fed the
oracle's upstream tensor and must be bit-exact.
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gtsfm_amd.utils import synthetic
from oracle import superpoint_oracle as spo
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

DESC_TOL = 1e-4
SCORE_TOL = 1e-4


@pytest.fixture(scope="module")
def sd():
    return synthetic.synthetic_superpoint_state_dict()


@pytest.fixture(scope="module")
def engine(gpu_device, sd):
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    return SuperPointEngine(sd, gpu_device)


@pytest.fixture(scope="module")
def lib(built_library):
    from gtsfm_amd.runtime import lib as L

    return L.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(lib, rc):
    assert rc == 0, lib.gtsfm_last_error().decode()


def _pack_conv(lib, w, dev):
    cout, cin = w.shape[:2]
    out = np.empty(lib.gtsfm_packed_conv3x3_floats(cin, cout), np.float32)
    wc = np.ascontiguousarray(w.numpy())
    _check(lib, lib.gtsfm_pack_conv3x3(wc.ctypes.data, cin, cout, out.ctypes.data))
    return torch.from_numpy(out).to(dev)


def _pack_linear(lib, w, dev, k_pad=None):
    n, k = w.shape
    k_pad = k_pad or (k + 7) // 8 * 8
    out = np.empty(lib.gtsfm_packed_linear_floats(k_pad, n), np.float32)
    wc = np.ascontiguousarray(w.numpy())
    _check(lib, lib.gtsfm_pack_linear(wc.ctypes.data, k, k_pad, n, out.ctypes.data))
    return torch.from_numpy(out).to(dev)


def _pad64(b, dev):
    o = torch.zeros((b.numel() + 63) // 64 * 64)
    o[: b.numel()] = b
    return o.to(dev)


# ------------------------------------------------------------------------------------------------------------------
# stage-wise parity
# ------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize(
    "name,batch,h,w,pool",
    [("conv1b", 2, 37, 53, 0), ("conv1b", 2, 37, 53, 1), ("conv2a", 1, 8, 16, 0), ("conv3a", 1, 24, 40, 0),
     ("conv3b", 2, 24, 41, 1), ("convPa", 1, 16, 16, 0), ("conv4a", 3, 5, 3, 0), ("conv1b", 1, 1, 1, 0)],
)
def test_conv3x3_matches_aten(lib, gpu_device, sd, name, batch, h, w, pool):
    """conv3x3 + bias + ReLU (+ fused 2x2 max-pool, floor) vs F.conv2d / F.max_pool2d (superpoint.py:148-161)."""
    wt, bs = sd[f"{name}.weight"], sd[f"{name}.bias"]
    cout, cin = wt.shape[:2]
    gen = torch.Generator().manual_seed(h * 1000 + w)
    x = torch.randn((batch, cin, h, w), generator=gen)
    ref = F.relu(F.conv2d(x, wt, bs, padding=1))
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    out = torch.full((batch, ho, wo, cout), float("nan"), device=gpu_device)
    wp, bp = _pack_conv(lib, wt, gpu_device), _pad64(bs, gpu_device)
    _check(lib, lib.gtsfm_conv3x3_f32(xd.data_ptr(), cin, 0, out.data_ptr(), cout, 0, wp.data_ptr(), bp.data_ptr(), batch,
                                      h, w, cin, cout, 1, pool, _stream()))
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2)
    assert not torch.isnan(got).any()
    assert float((got - ref).abs().max()) < 5e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("batch,h,w,pool,u8", [(2, 37, 53, 1, True), (1, 64, 48, 0, False), (1, 9, 200, 1, True), (3, 1, 1, 0, True)])
def test_fused_first_layer_matches_aten(lib, gpu_device, sd, batch, h, w, pool, u8):
    """relu(conv1a) recomputed inside conv1b's halo staging (gtsfm_conv1_fused_f32, the form gtsfm_sp_forward runs) vs ATen:
    conv1a's values are the 9-fma sums of the stand-alone first layer (u8 / 255 read from the gray patch staged once per tile),
    so the result matches F.conv2d of F.conv2d within the conv tolerance, at image borders and for ragged tiles."""
    gen = torch.Generator().manual_seed(h * 1000 + w)
    img = torch.randint(0, 256, (batch, h, w), generator=gen, dtype=torch.uint8) if u8 else torch.rand((batch, h, w), generator=gen)
    x = (img.float() / 255.0 if u8 else img)[:, None]
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x, sd["conv1a.weight"], sd["conv1a.bias"], padding=1)), sd["conv1b.weight"], sd["conv1b.bias"], padding=1))
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    out = torch.full((batch, ho, wo, 64), float("nan"), device=gpu_device)
    w1a = sd["conv1a.weight"].reshape(64, 9).t().contiguous().to(gpu_device)  # [tap][channel]
    b1a = sd["conv1a.bias"].contiguous().to(gpu_device)
    wp, bp = _pack_conv(lib, sd["conv1b.weight"], gpu_device), _pad64(sd["conv1b.bias"], gpu_device)
    imd = img.contiguous().to(gpu_device)
    _check(lib, lib.gtsfm_conv1_fused_f32(imd.data_ptr(), int(u8), w1a.data_ptr(), b1a.data_ptr(), wp.data_ptr(), bp.data_ptr(), batch, h, w, pool,
                                          out.data_ptr(), _stream()))
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2)
    assert not torch.isnan(got).any()
    assert float((got - ref).abs().max()) < 5e-5 * max(1.0, float(ref.abs().max()))


def test_conv3x3_strided_channel_views(lib, gpu_device, sd):
    """Reads a channel window of a wider NHWC buffer and writes into a channel window of another."""
    wt, bs = sd["conv2a.weight"], sd["conv2a.bias"]
    x = torch.randn((1, 64, 9, 20))
    ref = F.conv2d(x, wt, bs, padding=1)
    wide = torch.randn((1, 9, 20, 192))
    wide[..., 64:128] = x.permute(0, 2, 3, 1)
    out = torch.full((1, 9, 20, 128), 7.0, device=gpu_device)
    wp, bp = _pack_conv(lib, wt, gpu_device), _pad64(bs, gpu_device)
    wd = wide.to(gpu_device)
    _check(lib, lib.gtsfm_conv3x3_f32(wd.data_ptr(), 192, 64, out.data_ptr(), 128, 64, wp.data_ptr(), bp.data_ptr(), 1, 9, 20,
                                      64, 64, 0, 0, _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.all(got[..., :64] == 7.0)
    assert float((got[..., 64:].permute(0, 3, 1, 2) - ref).abs().max()) < 5e-5


@pytest.mark.parametrize("m,k,n,relu,res", [(300, 256, 65, 0, 0), (129, 256, 256, 1, 0), (1, 512, 256, 0, 1), (700, 32, 64, 1, 1),
                                            (64, 8, 32, 0, 0)])
def test_linear_matches_aten(lib, gpu_device, m, k, n, relu, res):
    """1x1 conv / Conv1d(k=1) / Linear (superpoint.py:162,191; superglue.py:49-60) incl. residual epilogue."""
    gen = torch.Generator().manual_seed(m + k + n)
    a = torch.randn((m, k), generator=gen)
    w = torch.randn((n, k), generator=gen) / k**0.5
    b = torch.randn((n,), generator=gen)
    r = torch.randn((m, n), generator=gen)
    ref = F.linear(a, w, b) * 0.5
    if relu:
        ref = F.relu(ref)
    if res:
        ref = r + ref
    ad, rd = a.to(gpu_device), r.to(gpu_device)
    out = torch.full((m, n + 3), -5.0, device=gpu_device)
    wp, bp = _pack_linear(lib, w, gpu_device), _pad64(b, gpu_device)
    # alpha scales (A W^T + bias)
    _check(lib, lib.gtsfm_linear_f32(ad.data_ptr(), k, m, None, k, wp.data_ptr(), bp.data_ptr(), n, out.data_ptr(), n + 3, 2,
                                     rd.data_ptr() if res else None, n, 0.5, relu, _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.all(got[:, :2] == -5.0) and torch.all(got[:, n + 2 :] == -5.0)
    assert float((got[:, 2 : n + 2] - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("m,k,n,relu,res,m_live,n_live", [(300, 256, 65, 0, 0, 300, 65), (129, 256, 256, 1, 0, 129, 256), (1, 512, 256, 0, 1, 1, 256),
                                                          (700, 32, 64, 1, 1, 700, 64), (513, 64, 200, 0, 0, 400, 200), (260, 256, 300, 0, 0, 260, 257)])
def test_linear_rowmajor_matches_aten(lib, gpu_device, m, k, n, relu, res, m_live, n_live):
    """The LDS-DMA GEMM (row-major weights; the matchers' projections and score products): same contract as
    gtsfm_linear_f32, plus the device-side column count used after LightGlue's point pruning."""
    gen = torch.Generator().manual_seed(m + k + n)
    a = torch.randn((m, k), generator=gen)
    w = torch.randn((n, k), generator=gen) / k**0.5
    b = torch.randn((n,), generator=gen)
    r = torch.randn((m, n), generator=gen)
    ref = F.linear(a, w, b) * 0.5
    if relu:
        ref = F.relu(ref)
    if res:
        ref = r + ref
    ad, wd, rd = a.to(gpu_device), w.to(gpu_device), r.to(gpu_device)
    out = torch.full((m, n + 3), -5.0, device=gpu_device)  # ldc = n + 3 and c_coff = 2: scalar stores
    bp = _pad64(b, gpu_device)
    md = torch.tensor([m_live], dtype=torch.int32, device=gpu_device)
    nd = torch.tensor([n_live], dtype=torch.int32, device=gpu_device)
    _check(lib, lib.gtsfm_linear_rowmajor_f32(ad.data_ptr(), k, m, md.data_ptr() if m_live < m else None, k, wd.data_ptr(), k, bp.data_ptr(), n,
                                              nd.data_ptr() if n_live < n else None, out.data_ptr(), n + 3, 2, rd.data_ptr() if res else None, n,
                                              0.5, relu, _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.all(got[:, :2] == -5.0) and torch.all(got[:, n_live + 2 :] == -5.0) and torch.all(got[m_live:] == -5.0)
    assert float((got[:m_live, 2 : n_live + 2] - ref[:m_live, :n_live]).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("m,k,n,relu,res,m_live,n_live", [(300, 256, 68, 0, 0, 300, 68), (1000, 512, 512, 1, 1, 1000, 512), (260, 256, 300, 0, 1, 200, 296),
                                                          (4097, 32, 128, 1, 0, 4097, 128), (131, 256, 768, 0, 0, 131, 768), (640, 512, 256, 0, 1, 640, 132)])
def test_linear_rowmajor_aligned_layout(lib, gpu_device, m, k, n, relu, res, m_live, n_live):
    """Same kernel with 16-byte aligned rows (ldc % 4 == 0, live column count % 4 == 0): the epilogue that transposes each
    accumulator tile through LDS and writes full 128-byte lines, incl. partial row / column tiles, several column blocks per
    workgroup, a single 32-deep stage, bias + scale + ReLU + residual."""
    gen = torch.Generator().manual_seed(m * 3 + k + n)
    a = torch.randn((m, k), generator=gen)
    w = torch.randn((n, k), generator=gen) / k**0.5
    b = torch.randn((n,), generator=gen)
    r = torch.randn((m, n + 4), generator=gen)
    ref = F.linear(a, w, b) * 0.5
    if relu:
        ref = F.relu(ref)
    if res:
        ref = r[:, :n] + ref
    ad, wd, rd = a.to(gpu_device), w.to(gpu_device), r.to(gpu_device)
    out = torch.full((m, n + 8), -5.0, device=gpu_device)  # ldc = n + 8, c_coff = 4
    bp = _pad64(b, gpu_device)
    md = torch.tensor([m_live], dtype=torch.int32, device=gpu_device)
    nd = torch.tensor([n_live], dtype=torch.int32, device=gpu_device)
    _check(lib, lib.gtsfm_linear_rowmajor_f32(ad.data_ptr(), k, m, md.data_ptr() if m_live < m else None, k, wd.data_ptr(), k, bp.data_ptr(), n,
                                              nd.data_ptr() if n_live < n else None, out.data_ptr(), n + 8, 4, rd.data_ptr() if res else None, n + 4,
                                              0.5, relu, _stream()))
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.all(got[:, :4] == -5.0) and torch.all(got[:, n_live + 4 :] == -5.0) and torch.all(got[m_live:] == -5.0)
    assert float((got[:m_live, 4 : n_live + 4] - ref[:m_live, :n_live]).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("m,k,n,relu,res,alpha,m_live", [(4096, 256, 768, 0, 0, 1.0, 4096), (4096, 512, 256, 0, 1, 1.0, 4096), (1000, 512, 512, 1, 1, 0.5, 1000),
                                                         (10240, 256, 512, 0, 0, 1.0, 10240), (70, 256, 256, 1, 0, 0.25, 70), (1, 32, 64, 0, 1, 1.0, 1),
                                                         (2500, 512, 132, 0, 0, 1.0, 2100)])
def test_linear_small_tiles_are_bit_identical_to_large_tiles(lib, gpu_device, monkeypatch, m, k, n, relu, res, alpha, m_live):
    """Launches that would not fill the chip with 128 x 128 tiles (one keypoint-set pair: the per-call plugin API) run on 64 x 64
    tiles (gemm_dma_small_kernel). Same k-ordered MFMA chain per output element and the same epilogue: the two tilings must agree
    BIT FOR BIT, so that which one runs stays a launch-geometry decision (batched == single-pair results)."""
    gen = torch.Generator().manual_seed(m + 7 * k + n)
    ad = torch.randn((m, k), generator=gen).to(gpu_device)
    wd = (torch.randn((n, k), generator=gen) / k**0.5).to(gpu_device)
    bp = _pad64(torch.randn((n,), generator=gen), gpu_device)
    rd = torch.randn((m, n + 4), generator=gen).to(gpu_device)
    md = torch.tensor([m_live], dtype=torch.int32, device=gpu_device)

    def run(small_below):
        monkeypatch.setenv("GTSFM_GEMM_SMALL_BELOW", str(small_below))
        out = torch.full((m, n + 8), -5.0, device=gpu_device)
        _check(lib, lib.gtsfm_linear_rowmajor_f32(ad.data_ptr(), k, m, md.data_ptr() if m_live < m else None, k, wd.data_ptr(), k, bp.data_ptr(), n, None,
                                                  out.data_ptr(), n + 8, 4, rd.data_ptr() if res else None, n + 4, alpha, relu, _stream()))
        torch.cuda.synchronize()
        return out.cpu()

    large, small = run(0), run(1 << 40)
    assert torch.equal(large, small)
    assert torch.all(small[:, :4] == -5.0) and torch.all(small[:, n + 4 :] == -5.0) and torch.all(small[m_live:] == -5.0)
    assert float(small[:m_live, 4 : n + 4].abs().max()) > 0.1


def test_linear_with_packed_activation_operand(lib, gpu_device):
    """A B^T of two activation matrices through pack_rows (score GEMM, superglue.py:257-258)."""
    a, b = torch.randn((150, 256)), torch.randn((90, 256))
    ref = (a @ b.T) / 16.0
    ad, bd = a.to(gpu_device), b.to(gpu_device)
    packed = torch.empty(lib.gtsfm_packed_linear_floats(256, 90), dtype=torch.float32, device=gpu_device)
    _check(lib, lib.gtsfm_pack_rows_f32(bd.data_ptr(), 256, 90, None, 256, packed.data_ptr(), _stream()))
    out = torch.empty((150, 90), device=gpu_device)
    _check(lib, lib.gtsfm_linear_f32(ad.data_ptr(), 256, 150, None, 256, packed.data_ptr(), None, 90, out.data_ptr(), 90, 0, None,
                                     0, 1.0 / 16.0, 0, _stream()))
    torch.cuda.synchronize()
    assert float((out.cpu() - ref).abs().max()) < 2e-5


def test_softmax_depth_to_space(lib, gpu_device):
    """superpoint.py:163-166."""
    logits = torch.randn((2, 7, 9, 65)) * 4
    s = F.softmax(logits.permute(0, 3, 1, 2), 1)[:, :-1]
    b, _, h, w = s.shape
    ref = s.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    ld = logits.to(gpu_device)
    out = torch.empty((2, 56, 72), device=gpu_device)
    _check(lib, lib.gtsfm_sp_softmax_d2s(ld.data_ptr(), 65, 2, 7, 9, out.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert float((out.cpu() - ref).abs().max()) < 1e-6


@pytest.mark.parametrize("h,w,radius", [(64, 96, 4), (33, 70, 4), (40, 40, 0), (50, 31, 2), (9, 9, 4), (130, 67, 8)])
def test_simple_nms_bit_exact(lib, gpu_device, h, w, radius):
    """superpoint.py:47-62: comparison-only -> bit-exact, incl. ties (quantised scores) and non-tile sizes."""
    gen = torch.Generator().manual_seed(h * w + radius)
    scores = torch.rand((2, h, w), generator=gen)
    scores[1] = torch.round(scores[1] * 8) / 8  # many exact ties
    ref = spo.simple_nms(scores, radius)
    sdv = scores.to(gpu_device)
    scratch = torch.empty(lib.gtsfm_sp_nms_scratch_bytes(2, h, w), dtype=torch.uint8, device=gpu_device)
    out = torch.empty_like(sdv)
    _check(lib, lib.gtsfm_sp_simple_nms(sdv.data_ptr(), 2, h, w, radius, scratch.data_ptr(), out.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("h,w,border,capacity", [(40, 72, 4, 4096), (40, 72, 4, 17), (16, 16, 0, 512), (8, 200, 4, 512), (1100, 70, 3, 9000)])
def test_extract_keypoints_bit_exact(lib, gpu_device, h, w, border, capacity):
    """superpoint.py:170-178,187: row-major nonzero order, border removal, (x, y) float flip; capacity clamp."""
    gen = torch.Generator().manual_seed(h + w)
    nms = torch.rand((3, h, w), generator=gen)
    nms[nms < 0.9] = 0.0
    nms[2] = 0.0  # an image with no keypoints
    thr = 0.92
    scratch = torch.empty((3 * h * 2 + 3,), dtype=torch.int32, device=gpu_device)
    count = torch.empty(3, dtype=torch.int32, device=gpu_device)
    raw = torch.empty(3, dtype=torch.int32, device=gpu_device)
    xy = torch.zeros((3, capacity, 2), device=gpu_device)
    sc = torch.zeros((3, capacity), device=gpu_device)
    nd = nms.to(gpu_device)
    _check(lib, lib.gtsfm_sp_extract_keypoints(nd.data_ptr(), 3, h, w, thr, border, capacity, scratch.data_ptr(), count.data_ptr(),
                                               raw.data_ptr(), xy.data_ptr(), sc.data_ptr(), _stream()))
    torch.cuda.synchronize()
    for b in range(3):
        kp = torch.nonzero(nms[b] > thr)
        s = nms[b][tuple(kp.t())]
        kp, s = spo.remove_borders(kp, s, border, h, w)
        kp = torch.flip(kp, [1]).float()
        assert int(raw[b]) == kp.shape[0]
        k = min(kp.shape[0], capacity)
        assert int(count[b]) == k
        assert torch.equal(xy[b, :k].cpu(), kp[:k]) and torch.equal(sc[b, :k].cpu(), s[:k])


def test_sample_descriptors(lib, gpu_device):
    """superpoint.py:80-92,192,195-196 with align_corners=True; corner cells and border keypoints included."""
    hc, wc = 9, 13
    dense = torch.randn((2, 256, hc, wc))
    kp = torch.stack([torch.randint(0, wc * 8, (2, 300)).float(), torch.randint(0, hc * 8, (2, 300)).float()], -1)
    kp[0, 0] = torch.tensor([0.0, 0.0])
    kp[0, 1] = torch.tensor([wc * 8 - 1.0, hc * 8 - 1.0])
    count = torch.tensor([300, 123], dtype=torch.int32)
    dd = dense.permute(0, 2, 3, 1).reshape(2, hc * wc, 256).contiguous().to(gpu_device)
    out = torch.full((2, 300, 256), float("nan"), device=gpu_device)
    kd, cd = kp.to(gpu_device), count.to(gpu_device)
    _check(lib, lib.gtsfm_sp_sample_descriptors(dd.data_ptr(), 256, 2, hc, wc, kd.data_ptr(), cd.data_ptr(), 300, out.data_ptr(), _stream()))
    torch.cuda.synchronize()
    for b in range(2):
        ref = spo.sample_descriptors(kp[b : b + 1], F.normalize(dense[b : b + 1], p=2, dim=1), 8)[0].T
        k = int(count[b])
        assert float((out[b, :k].cpu() - ref[:k]).abs().max()) < 2e-6
        assert torch.isnan(out[b, k:]).all()  # rows beyond the count are untouched


# ------------------------------------------------------------------------------------------------------------------
# end to end
# ------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("path", sorted(GOLDEN.glob("superpoint_*.npz")), ids=lambda p: p.stem)
def test_end_to_end_matches_golden(engine, path):
    """Whole model vs the golden vectors produced by the reference's own superpoint.py."""
    g = np.load(path)
    gray = synthetic.synthetic_gray_image(int(g["height"]), int(g["width"]), int(g["seed"]))
    xy, sc, de = engine.detect(gray)
    np.testing.assert_array_equal(xy.astype(np.int32), g["keypoints"])  # bit-exact keypoints, row-major order
    assert xy.dtype == np.float32 and np.array_equal(xy, g["keypoints"].astype(np.float32))
    np.testing.assert_allclose(sc, g["scores"], rtol=0, atol=SCORE_TOL)
    np.testing.assert_allclose(de, g["descriptors"], rtol=0, atol=DESC_TOL)


@pytest.mark.parametrize("h,w,seed", [(480, 640, 21), (203, 331, 22), (64, 64, 23)])
def test_end_to_end_matches_oracle(engine, sd, h, w, seed):
    """Config-2-shaped image (480x640) and ragged sizes vs the oracle, incl. dense score / NMS maps."""
    gray = synthetic.synthetic_gray_image(h, w, seed)
    out = engine.forward(torch.from_numpy(gray).to(engine.device)[None], return_score_maps=True)
    with torch.no_grad():
        ora = spo.superpoint_forward(sd, spo.gray_u8_to_tensor(gray), return_intermediates=True)
    k = int(out["count"][0])
    assert k == ora["keypoints"].shape[0] == int(out["count_raw"][0])
    assert torch.equal(out["xy"][0, :k].cpu(), ora["keypoints"])
    assert float((out["dense_scores"][0].cpu() - ora["dense_scores"][0]).abs().max()) < SCORE_TOL
    assert torch.equal(out["nms_scores"][0].cpu() > 0, ora["nms_scores"][0] > 0)
    assert float((out["scores"][0, :k].cpu() - ora["scores"]).abs().max()) < SCORE_TOL
    assert float((out["descriptors"][0, :k].cpu() - ora["descriptors"].T).abs().max()) < DESC_TOL


def test_batch_equals_single_and_u8_equals_float(engine):
    """Batched launch == per-image launches (bit-exact); uint8 input == astype(float32)/255 input (bit-exact)."""
    imgs = np.stack([synthetic.synthetic_gray_image(96, 136, s) for s in (31, 32, 33)])
    dev = engine.device
    batched = engine.forward(torch.from_numpy(imgs).to(dev))
    as_float = engine.forward(torch.from_numpy(imgs.astype(np.float32) / 255.0).to(dev))
    for key in ("count", "xy", "scores"):
        assert torch.equal(batched[key], as_float[key]) or key != "count"
    for i in range(3):
        single = engine.forward(torch.from_numpy(imgs[i : i + 1]).to(dev))
        k = int(single["count"][0])
        assert k == int(batched["count"][i]) == int(as_float["count"][i]) and k > 0
        for key in ("xy", "scores", "descriptors"):
            assert torch.equal(single[key][0, :k], batched[key][i, :k])
            assert torch.equal(single[key][0, :k], as_float[key][i, :k])


def test_edge_cases(engine):
    dev = engine.device
    # smaller than one 8x8 cell -> no keypoints
    out = engine.forward(torch.zeros((2, 7, 5), dtype=torch.uint8, device=dev))
    assert out["count"].tolist() == [0, 0]
    # capacity overflow: clamped count, true count reported, first rows identical
    gray = synthetic.synthetic_gray_image(160, 160, 41)
    full = engine.forward(torch.from_numpy(gray).to(dev)[None])
    k = int(full["count"][0])
    assert k > 20
    small = engine.forward(torch.from_numpy(gray).to(dev)[None], capacity=16)
    assert int(small["count"][0]) == 16 and int(small["count_raw"][0]) == k
    assert torch.equal(small["xy"][0], full["xy"][0, :16]) and torch.equal(small["descriptors"][0], full["descriptors"][0, :16])
    # workspace-size validation goes through the error string
    from gtsfm_amd.runtime import lib as L

    lib = L.load()
    img = torch.from_numpy(gray).to(dev)[None]
    tiny = torch.empty(1024, dtype=torch.uint8, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    buf = torch.empty((1, 16, 256), device=dev)
    rc = lib.gtsfm_sp_forward(engine.weights.data_ptr(), img.data_ptr(), 1, 1, 160, 160, 0.005, 4, 4, 16, 0, tiny.data_ptr(), tiny.numel(),
                              cnt.data_ptr(), None, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), None, None, _stream())
    assert rc == -3 and b"workspace" in lib.gtsfm_last_error()


def test_full_size_properties(engine):
    """BASELINE config-3 image size (1024x1024): size-independent properties of the detector output --
    determinism, row-major sortedness, border removal, NMS separation, unit-norm descriptors, threshold."""
    gray = synthetic.synthetic_gray_image(1024, 1024, 7)
    img = torch.from_numpy(gray).to(engine.device)[None]
    a = engine.forward(img)
    b = engine.forward(img)
    k = int(a["count"][0])
    assert k == int(b["count"][0]) and 1000 < k < 20000
    for key in ("xy", "scores", "descriptors"):
        assert torch.equal(a[key][0, :k], b[key][0, :k])  # deterministic (tests/repro_tests analogue)
    xy = a["xy"][0, :k].cpu().numpy().astype(np.int64)
    lin = xy[:, 1] * 1024 + xy[:, 0]
    assert np.all(np.diff(lin) > 0)  # torch.nonzero order
    assert xy.min() >= 4 and xy[:, 0].max() < 1020 and xy[:, 1].max() < 1020  # remove_borders(4)
    assert float(a["scores"][0, :k].min()) > 0.005
    norms = a["descriptors"][0, :k].norm(dim=1).cpu().numpy()
    np.testing.assert_allclose(norms, 1.0, atol=1e-5)
    # NMS radius 4: no two survivors within Chebyshev distance 4 unless their scores tie exactly
    grid = -np.ones((1024, 1024), dtype=np.int64)
    grid[xy[:, 1], xy[:, 0]] = np.arange(k)
    sc = a["scores"][0, :k].cpu().numpy()
    for dy in range(0, 5):
        for dx in range(-4, 5):
            if dy == 0 and dx <= 0:
                continue
            y2, x2 = xy[:, 1] + dy, xy[:, 0] + dx
            ok = (y2 < 1024) & (x2 >= 0) & (x2 < 1024)
            nb = grid[y2[ok], x2[ok]]
            hit = nb >= 0
            assert np.all(sc[np.flatnonzero(ok)[hit]] == sc[nb[hit]])


def test_plugin_matches_oracle_wrapper(gpu_device, sd, tmp_path):
    """SuperPointDetectorDescriptor.detect_and_describe vs the restated reference wrapper
    (gtsfm/frontend/detector_descriptor/superpoint.py:63-93), incl. mask filtering and top-k selection, plus the
    reference's API-contract checks (tests/frontend/detector_descriptor/test_detector_descriptor_base.py:29-42)."""
    from gtsfm_amd.common.image import Image
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor

    path = tmp_path / "superpoint_v1.pth"
    torch.save(sd, str(path))
    gray = synthetic.synthetic_gray_image(240, 320, 3)
    rgb = np.stack([gray, gray, gray], -1)  # gray-valued RGB: the fixed-point gray conversion is the identity
    mask = np.zeros((240, 320), dtype=np.uint8)
    mask[20:200, 30:300] = 1
    for max_kp, use_mask in [(5000, False), (200, False), (150, True)]:
        det = SuperPointDetectorDescriptor(max_keypoints=max_kp, weights_path=path)
        kps, desc = det.detect_and_describe(Image(value_array=rgb, mask=mask if use_mask else None))
        rc, rs, rd = spo.detect_and_describe(sd, gray, max_keypoints=max_kp, mask=mask if use_mask else None)
        assert len(kps) <= max_kp and len(kps) == desc.shape[0] == len(rc)
        assert kps.coordinates.dtype == np.float32 and desc.dtype == np.float32 and kps.scales is None
        assert (kps.coordinates[:, 0] >= 0).all() and (kps.coordinates[:, 0] < 320).all()
        assert (kps.coordinates[:, 1] >= 0).all() and (kps.coordinates[:, 1] < 240).all()
        # same selection as the reference wrapper (argpartition order is implementation-defined: compare as sets,
        # then row by row after sorting both by pixel index)
        oa = np.lexsort((kps.coordinates[:, 0], kps.coordinates[:, 1]))
        ob = np.lexsort((rc[:, 0], rc[:, 1]))
        np.testing.assert_array_equal(kps.coordinates[oa], rc[ob])
        np.testing.assert_allclose(kps.responses[oa], rs[ob], rtol=0, atol=SCORE_TOL)
        np.testing.assert_allclose(desc[oa], rd[ob], rtol=0, atol=DESC_TOL)


@pytest.mark.parametrize("k", [1, 100, 333, 5000])
def test_device_topk_equals_host_selection(engine, k):
    """The GPU-resident path keeps the top-k responses on the device (Keypoints.get_top_k on the host in the
    reference, gtsfm/common/keypoints.py:89-110): same SET as argpartition on the full detection, rows in detection
    order, values bit-identical to the unselected run."""
    imgs = np.stack([synthetic.synthetic_gray_image(200, 264, s) for s in (51, 52)])
    dev = engine.device
    full = engine.forward(torch.from_numpy(imgs).to(dev))
    sel = engine.forward(torch.from_numpy(imgs).to(dev), top_k=k)
    for b in range(2):
        n = int(full["count"][b])
        kk = int(sel["count"][b])
        assert kk == min(n, k) and int(sel["count_raw"][b]) == n
        fs = full["scores"][b, :n].cpu().numpy()
        fxy = full["xy"][b, :n].cpu().numpy().astype(np.int64)
        sxy = sel["xy"][b, :kk].cpu().numpy().astype(np.int64)
        lin_full = fxy[:, 1] * 264 + fxy[:, 0]
        lin_sel = sxy[:, 1] * 264 + sxy[:, 0]
        assert np.all(np.diff(lin_sel) > 0)
        pos = np.searchsorted(lin_full, lin_sel)
        assert np.array_equal(lin_full[pos], lin_sel)
        if kk < n:
            kth = np.sort(fs)[::-1][kk - 1]
            assert (fs[pos] >= kth).all() and (np.delete(fs, pos) <= kth).all()
        assert np.array_equal(sel["scores"][b, :kk].cpu().numpy(), fs[pos])
        assert torch.equal(sel["descriptors"][b, :kk].cpu(), full["descriptors"][b, :n].cpu()[pos])


def test_device_topk_with_ties(lib, gpu_device):
    """Ties at the k-th value are resolved by detection order; fewer candidates than k are all kept."""
    scores = torch.tensor([[0.5, 0.9, 0.5, 0.5, 0.1, 0.5, 0.7, 0.0], [0.3, 0.2, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]], device=gpu_device)
    xy = torch.arange(32, dtype=torch.float32, device=gpu_device).reshape(2, 8, 2)
    count = torch.tensor([7, 2], dtype=torch.int32, device=gpu_device)
    oxy = torch.full((2, 4, 2), -1.0, device=gpu_device)
    osc = torch.full((2, 4), -1.0, device=gpu_device)
    ocnt = torch.empty(2, dtype=torch.int32, device=gpu_device)
    _check(lib, lib.gtsfm_sp_select_topk(scores.data_ptr(), xy.data_ptr(), count.data_ptr(), 2, 8, 4, oxy.data_ptr(), osc.data_ptr(),
                                         ocnt.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert ocnt.tolist() == [4, 2]
    assert osc[0].tolist() == [0.5, pytest.approx(0.9), 0.5, pytest.approx(0.7)]  # 0.9, 0.7 and the first two 0.5s, in order
    assert oxy[0, :, 0].tolist() == [0.0, 2.0, 4.0, 12.0]
    assert osc[1, :2].tolist() == [pytest.approx(0.3), pytest.approx(0.2)]


def test_real_images_lund_door(engine):
    """BASELINE config 1 (plumbing): real photographs (two frames of the reference's tests/data/set1_lund_door, 568x380
    after reduction) vs the reference's own SuperPoint outputs: ~1 850 keypoints per frame, identical; scores and
    descriptors within tolerance."""
    g = np.load(GOLDEN / "lund_door_pair.npz")
    for i in (0, 1):
        xy, sc, de = engine.detect(g[f"gray{i}"])
        np.testing.assert_array_equal(xy.astype(np.int32), g[f"keypoints{i}"])
        np.testing.assert_allclose(sc, g[f"scores{i}"], rtol=0, atol=SCORE_TOL)
        np.testing.assert_allclose(de[:256], g[f"descriptors{i}_head"], rtol=0, atol=DESC_TOL)


def test_end_to_end_matches_golden_config2_shape(engine):
    """BASELINE config-2 shape (480x640): full keypoint list, scores and the first 256 descriptors vs the reference."""
    g = np.load(GOLDEN / "config2_superpoint_480x640_s4.npz")
    xy, sc, de = engine.detect(synthetic.synthetic_gray_image(480, 640, 4))
    np.testing.assert_array_equal(xy.astype(np.int32), g["keypoints"])
    np.testing.assert_allclose(sc, g["scores"], rtol=0, atol=SCORE_TOL)
    np.testing.assert_allclose(de[:256], g["descriptors_head"], rtol=0, atol=DESC_TOL)


def test_device_mask_equals_filter_by_mask_then_top_k(gpu_device):
    """gtsfm_sp_forward_masked: the image mask applied on the device ahead of the top-k against the wrapper's host sequence
    (Keypoints.filter_by_mask, then get_top_k; gtsfm/frontend/detector_descriptor/superpoint.py:76-91) on the unmasked output."""
    from gtsfm_amd.common.keypoints import Keypoints
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    eng = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device)
    imgs = np.stack([synthetic.synthetic_gray_image(200, 264, s) for s in (5, 6, 7)])
    masks = np.zeros((3, 200, 264), dtype=np.uint8)
    masks[0, 20:150, 30:200] = 1
    masks[1] = 1                      # everything valid
    masks[2, ::2, :] = 1              # every other row; values other than 1 are invalid as in the reference
    masks[2, 1::2, :] = 255
    x = torch.from_numpy(imgs).to(gpu_device)
    m = torch.from_numpy(masks).to(gpu_device)
    plain = eng.forward(x)
    masked = eng.forward(x, valid_masks=m)
    top = eng.forward(x, valid_masks=m, top_k=150)
    for b in range(3):
        c = int(plain["count"][b])
        xy = plain["xy"][b, :c].cpu().numpy()
        sc = plain["scores"][b, :c].cpu().numpy()
        de = plain["descriptors"][b, :c].cpu().numpy()
        kept, idx = Keypoints(coordinates=xy, responses=sc).filter_by_mask(masks[b])
        cm = int(masked["count"][b])
        assert cm == len(kept) and (b != 1 or cm == c) and cm > 0
        np.testing.assert_array_equal(masked["xy"][b, :cm].cpu().numpy(), kept.coordinates)
        np.testing.assert_array_equal(masked["scores"][b, :cm].cpu().numpy(), kept.responses)
        np.testing.assert_array_equal(masked["descriptors"][b, :cm].cpu().numpy(), de[idx])
        # top-k of the masked set, detection order (no score ties in these images)
        order = np.sort(synthetic.topk_detection_order(kept.responses, 150))
        ct = int(top["count"][b])
        assert ct == len(order)
        np.testing.assert_array_equal(top["xy"][b, :ct].cpu().numpy(), kept.coordinates[order])
        np.testing.assert_array_equal(top["descriptors"][b, :ct].cpu().numpy(), de[idx][order])


def test_detect_and_describe_from_several_threads_equals_one_at_a_time(gpu_device, sd, tmp_path):
    """Several Dask threads of one worker (``--threads_per_worker``, gtsfm/runner.py:155,436) call ``detect_and_describe`` on ONE scattered
    plugin object: the engine serves them on lanes of their own (staging buffers, workspace, stream; round 3 took turns behind one lock).
    Images of three different sizes from three threads, twice: exactly the keypoints / descriptors of the same calls made one after the other."""
    import threading

    from gtsfm_amd.common.image import Image
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor

    path = tmp_path / "superpoint_v1.pth"
    torch.save(sd, str(path))
    det = SuperPointDetectorDescriptor(max_keypoints=400, weights_path=path)
    images = [Image(value_array=synthetic.synthetic_gray_image(h, w, 60 + q)) for q, (h, w) in enumerate([(240, 320), (200, 264), (123, 157), (240, 320), (264, 200), (192, 256)])]
    serial = [det.detect_and_describe(im) for im in images]
    results, errors = {}, []

    def worker(tid):
        try:
            for rep in range(2):
                for q in range(tid, len(images), 3):
                    results[(rep, q)] = det.detect_and_describe(images[q])
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 2 * len(images) and 1 <= len(det._model._lanes) <= det._model.max_lanes
    for (rep, q), (kps, desc) in results.items():
        assert kps == serial[q][0]
        np.testing.assert_array_equal(desc, serial[q][1])
    det._model.release_lanes()
    assert len(det._model._lanes) == 1
    kps, desc = det.detect_and_describe(images[0])
    assert kps == serial[0][0] and np.array_equal(desc, serial[0][1])
