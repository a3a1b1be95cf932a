"""GPU parity tests of the matcher HIP paths (attention, SuperGlue, LightGlue) through the C ABI.

Tolerances (BASELINE.json north_star): match indices bit-exact; match scores within 1e-4 fp32 (observed ~2e-6 for
SuperGlue, ~2e-5 for LightGlue). SuperGlue is pinned on golden vectors produced by the reference's own superglue.py;
LightGlue parity is UNPINNED (source absent from the reference) and reads "HIP path == oracle/lightglue_oracle.py".
"""

import numpy as np
import pytest
import torch

from gtsfm_amd.utils import synthetic
from oracle import lightglue_oracle as lgo
from oracle import superglue_oracle as sgo
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4
T = torch.from_numpy


@pytest.fixture(scope="module")
def lib(built_library):
    from gtsfm_amd.runtime import lib as L

    return L.load()


@pytest.fixture(scope="module")
def sg_sd():
    return synthetic.synthetic_superglue_state_dict()


@pytest.fixture(scope="module")
def sg_engine(gpu_device, sg_sd):
    from gtsfm_amd.runtime.matcher_engine import SuperGlueEngine

    return SuperGlueEngine(sg_sd, gpu_device)


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------------


def _ref_attention(q, k, v, scale):
    nq, nk = q.shape[0], k.shape[0]
    qh, kh, vh = (t.view(-1, 4, 64).transpose(0, 1) for t in (q, k, v))
    return (torch.softmax(qh @ kh.transpose(1, 2) * scale, -1) @ vh).transpose(0, 1).reshape(nq, 256)


def test_attention_ragged_batch(lib, gpu_device):
    """softmax(q k^T / 8) v per head (superglue.py:85-89) on a ragged batch: self and cross problems, strided packed
    qkv buffer, counts smaller than the row capacity (rows beyond a count must stay untouched)."""
    counts = [100, 70, 257, 1, 513]
    caps = [128, 70, 300, 4, 513]
    offs = np.concatenate([[0], np.cumsum(caps)])[:-1]
    total = int(sum(caps))
    gen = torch.Generator().manual_seed(0)
    qkv = torch.randn((total, 768), generator=gen)
    qkv[:, :512] *= 2.0
    problems = [(0, 0), (1, 1), (0, 1), (1, 0), (2, 3), (3, 2), (4, 4), (2, 4)]
    prob_arr = torch.tensor([[offs[a], a, offs[b], b] for a, b in problems], dtype=torch.int32, device=gpu_device)
    cnt = torch.tensor(counts, dtype=torch.int32, device=gpu_device)
    d = qkv.to(gpu_device)
    for pi, (a, b) in enumerate(problems):
        out = torch.full((total, 256), float("nan"), device=gpu_device)
        rc = lib.gtsfm_attention_f32(d.data_ptr(), 768, d.data_ptr() + 256 * 4, 768, d.data_ptr() + 512 * 4, 768, out.data_ptr(), 256,
                                     prob_arr[pi : pi + 1].contiguous().data_ptr(), cnt.data_ptr(), 1, counts[a], 4, 0.125, _stream())
        assert rc == 0, lib.gtsfm_last_error()
        torch.cuda.synchronize()
        qa = qkv[offs[a] : offs[a] + counts[a], :256]
        kb = qkv[offs[b] : offs[b] + counts[b], 256:512]
        vb = qkv[offs[b] : offs[b] + counts[b], 512:]
        ref = _ref_attention(qa, kb, vb, 0.125)
        got = out.cpu()
        # logits reach +-45 here (ulp 4e-6 in fp32): both implementations carry ~1e-5 of round-off in the exponent
        assert float((got[offs[a] : offs[a] + counts[a]] - ref).abs().max()) < 2e-5
        untouched = torch.ones(total, dtype=torch.bool)
        untouched[offs[a] : offs[a] + counts[a]] = False
        assert torch.isnan(got[untouched]).all()
    # all problems in one launch == one at a time
    out_all = torch.zeros((total, 256), device=gpu_device)
    sel = prob_arr[[0, 4, 6]].contiguous()  # disjoint query ranges
    rc = lib.gtsfm_attention_f32(d.data_ptr(), 768, d.data_ptr() + 256 * 4, 768, d.data_ptr() + 512 * 4, 768, out_all.data_ptr(), 256,
                                 sel.data_ptr(), cnt.data_ptr(), 3, max(counts), 4, 0.125, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    ref0 = _ref_attention(qkv[:100, :256], qkv[:100, 256:512], qkv[:100, 512:], 0.125)
    assert float((out_all[:100].cpu() - ref0).abs().max()) < 2e-5


def test_attention_split_schedule_is_bit_identical_to_fused(lib, gpu_device):
    """Keys are processed in 1024-key segments, each from a fresh online-softmax state, merged in ascending order by one formula.
    The fused schedule (a workgroup walks all segments; what batches run) and the split schedule (one workgroup per segment + a
    combine kernel; what a single pair runs to fill the chip) must agree BIT FOR BIT: 1, 2, 3 and 5 segments, a last segment of
    one key, ragged problems in one launch, a problem without keys, late dominant keys (reference maximum moves across segments).
    Both are also held to the fp32 reference."""
    counts = [300, 1024, 1025, 2048, 2500, 5000, 0, 70]
    caps = [384, 1024, 1152, 2048, 2560, 5120, 128, 128]
    offs = np.concatenate([[0], np.cumsum(caps)])[:-1]
    total = int(sum(caps))
    gen = torch.Generator().manual_seed(3)
    qkv = torch.randn((total, 768), generator=gen)
    qkv[:, :512] *= 1.5
    qkv[offs[5] + 4990, 256:320] = qkv[offs[5] : offs[5] + 5000, :64].mean(0) * 40  # head 0 of the 5000-key set: a key of the LAST segment dominates
    qkv[offs[3] + 3, 256 + 64 : 256 + 128] = qkv[offs[3] : offs[3] + 2048, 64:128].mean(0) * 40  # head 1 of the 2048-key set: one of the FIRST segment
    # (queries, keys): self problems of every size, cross problems between different segment counts, no keys, few queries x many keys
    problems = [(0, 0), (1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (7, 6), (7, 5), (4, 2), (2, 4)]
    d = qkv.to(gpu_device)
    cnt = torch.tensor(counts, dtype=torch.int32, device=gpu_device)
    ws = torch.empty(int(lib.gtsfm_attention_split_workspace_bytes(len(problems), max(counts), max(counts), 4, total)), dtype=torch.uint8, device=gpu_device)

    def run(sel, mode):
        prob = torch.tensor([[offs[a], a, offs[b], b] for a, b in sel], dtype=torch.int32, device=gpu_device)
        out = torch.full((total, 256), float("nan"), device=gpu_device)
        rc = lib.gtsfm_attention_split_f32(d.data_ptr(), 768, d.data_ptr() + 256 * 4, 768, d.data_ptr() + 512 * 4, 768, out.data_ptr(), 256, prob.data_ptr(),
                                           cnt.data_ptr(), len(sel), max(counts[a] for a, _ in sel), max(counts), 4, 0.125, mode, total, ws.data_ptr(), ws.numel(),
                                           _stream())
        assert rc == 0, lib.gtsfm_last_error()
        torch.cuda.synchronize()
        return out.cpu()

    for sel in ([problems[0], problems[3], problems[5]], [problems[1], problems[2], problems[4], problems[6]], [problems[7]], [problems[8]], [problems[9]]):
        fused, split = run(sel, -1), run(sel, 1)
        for a, b in sel:
            rows = slice(offs[a], offs[a] + counts[a])
            assert torch.equal(fused[rows], split[rows]), (a, b)
            if counts[b] == 0:
                assert float(fused[rows].abs().max()) == 0.0
                continue
            ref = _ref_attention(qkv[rows, :256], qkv[offs[b] : offs[b] + counts[b], 256:512], qkv[offs[b] : offs[b] + counts[b], 512:], 0.125)
            assert float((fused[rows] - ref).abs().max()) < 3e-5
        touched = torch.zeros(total, dtype=torch.bool)
        for a, _ in sel:
            touched[offs[a] : offs[a] + counts[a]] = True
        assert torch.isnan(fused[~touched]).all() and torch.isnan(split[~touched]).all()


def test_attention_peaked_softmax(lib, gpu_device):
    """Online-softmax rescaling across key tiles: a late key dominates every row (max jumps after several tiles)."""
    nq, nk = 200, 700
    gen = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn((n, 256), generator=gen) for n in (nq, nk, nk))
    k[650] = 0
    k[650, :64] = q[:, :64].mean(0) * 50  # head 0: key 650 (in the 11th tile) wins by a large margin
    ref = _ref_attention(q, k, v, 0.125)
    qd, kd, vd = q.to(gpu_device), k.to(gpu_device), v.to(gpu_device)
    out = torch.empty((nq, 256), device=gpu_device)
    prob = torch.tensor([[0, 0, 0, 1]], dtype=torch.int32, device=gpu_device)
    cnt = torch.tensor([nq, nk], dtype=torch.int32, device=gpu_device)
    rc = lib.gtsfm_attention_f32(qd.data_ptr(), 256, kd.data_ptr(), 256, vd.data_ptr(), 256, out.data_ptr(), 256, prob.data_ptr(),
                                 cnt.data_ptr(), 1, nq, 4, 0.125, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert float((out.cpu() - ref).abs().max()) < 1e-5


# ------------------------------------------------------------------------------------------------------------------
# SuperGlue
# ------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("path", sorted(GOLDEN.glob("superglue_*.npz")), ids=lambda p: p.stem)
def test_superglue_matches_golden(sg_engine, path):
    """Whole model vs golden vectors produced by the reference's own superglue.py (20 and 100 Sinkhorn iterations)."""
    g = np.load(path)
    shp0, shp1 = tuple(int(v) for v in g["shape0"]), tuple(int(v) for v in g["shape1"])
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(int(g["n0"]), int(g["n1"]), shp0, shp1, seed=int(g["seed"]))
    res = sg_engine.match_pair(k0, s0, d0, k1, s1, d1, shp0, shp1, sinkhorn_iterations=int(g["iters"]), return_ot=True)
    np.testing.assert_array_equal(res["matches0"], g["matches0"])  # bit-exact indices
    np.testing.assert_array_equal(res["matches1"], g["matches1"])
    np.testing.assert_allclose(res["matching_scores0"], g["matching_scores0"], rtol=0, atol=SCORE_TOL)
    np.testing.assert_allclose(res["matching_scores1"], g["matching_scores1"], rtol=0, atol=SCORE_TOL)
    np.testing.assert_allclose(res["ot"][::3, ::3], g["ot_sample"], rtol=0, atol=2e-4)  # log-space OT matrix, |values| ~ 10..60


def test_superglue_ragged_batch_equals_single_pairs(sg_engine, sg_sd):
    """A ragged batch (different keypoint counts and image shapes per pair) gives exactly the per-pair results, and
    those match the oracle."""
    specs = [(96, 80, (240, 320), (200, 300), 11), (130, 257, (480, 640), (480, 640), 21), (1, 5, (64, 64), (64, 64), 13), (300, 129, (600, 400), (400, 600), 22)]
    feats = [synthetic.synthetic_pair_features(a, b, s0, s1, seed=sd) for a, b, s0, s1, sd in specs]
    dev = sg_engine.device
    kp = T(np.concatenate([np.concatenate([f[0], f[3]]) for f in feats])).to(dev)
    sc = T(np.concatenate([np.concatenate([f[1], f[4]]) for f in feats])).to(dev)
    de = T(np.concatenate([np.concatenate([f[2], f[5]]) for f in feats])).to(dev)
    n0, n1 = [s[0] for s in specs], [s[1] for s in specs]
    hw = [[s[2][0], s[2][1], s[3][0], s[3][1]] for s in specs]
    out = sg_engine.match_batch(kp, sc, de, n0, n1, hw, sinkhorn_iterations=20)
    m, ms = out["matches"].cpu().numpy(), out["mscores"].cpu().numpy()
    row = 0
    for (a, b, s0, s1, _), f in zip(specs, feats):
        single = sg_engine.match_pair(f[0], f[1], f[2], f[3], f[4], f[5], s0, s1, sinkhorn_iterations=20)
        np.testing.assert_array_equal(m[row : row + a], single["matches0"])
        np.testing.assert_array_equal(m[row + a : row + a + b], single["matches1"])
        np.testing.assert_array_equal(ms[row : row + a], single["matching_scores0"])
        with torch.no_grad():
            ora = sgo.superglue_forward(sg_sd, T(f[0])[None], T(f[3])[None], T(f[1])[None], T(f[4])[None], T(f[2]).T[None].contiguous(),
                                        T(f[5]).T[None].contiguous(), s0, s1, sinkhorn_iterations=20)
        np.testing.assert_array_equal(single["matches0"], ora["matches0"][0].numpy())
        np.testing.assert_array_equal(single["matches1"], ora["matches1"][0].numpy())
        np.testing.assert_allclose(single["matching_scores0"], ora["matching_scores0"][0].numpy(), rtol=0, atol=SCORE_TOL)
        np.testing.assert_allclose(single["matching_scores1"], ora["matching_scores1"][0].numpy(), rtol=0, atol=SCORE_TOL)
        row += a + b


@pytest.mark.parametrize("iters", [0, 1, 100])
def test_superglue_sinkhorn_iteration_counts(sg_engine, sg_sd, iters):
    """The iteration count is a parameter: GTSfM runs 20, the third-party default / BASELINE config 4 is 100
    (SURVEY.md F5); 0 and 1 exercise the u = v = 0 start (superglue.py:143)."""
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(150, 170, (480, 640), (480, 640), seed=31)
    res = sg_engine.match_pair(k0, s0, d0, k1, s1, d1, (480, 640), (480, 640), sinkhorn_iterations=iters, return_ot=True)
    with torch.no_grad():
        ora = sgo.superglue_forward(sg_sd, T(k0)[None], T(k1)[None], T(s0)[None], T(s1)[None], T(d0).T[None].contiguous(),
                                    T(d1).T[None].contiguous(), (480, 640), (480, 640), sinkhorn_iterations=iters, return_intermediates=True)
    np.testing.assert_array_equal(res["matches0"], ora["matches0"][0].numpy())
    np.testing.assert_allclose(res["ot"], ora["ot"][0].numpy(), rtol=0, atol=2e-4)


@pytest.mark.parametrize("shapes,iters", [([(257, 300), (1, 5), (300, 255), (64, 1)], 20), ([(2048, 2048), (2047, 1500)], 100),
                                           ([(700, 513)], 1), ([(2100, 2300)], 20), ([(1500, 5000), (33, 2049), (40, 3072), (300, 200)], 20),
                                           ([(64, 5120), (31, 5121), (40, 2048), (65, 7000)], 5), ([(40, 10240), (20, 10241)], 3)],
                         ids=["ragged_small", "n2048_it100", "one_iteration", "wider_than_2048", "cap5000_mixed_tiers", "eight_wave_tier",
                              "lds_path_beyond_10240"])
def test_sinkhorn_standalone_vs_oracle(lib, gpu_device, shapes, iters):
    """The sweep kernels on their own (gtsfm_sinkhorn_f32) against superglue.py:150-170 as restated by the oracle: ragged
    batches, widths on both sides of the 256-column register chunks, the workgroup-per-row tiers beyond 2048 columns (4 waves
    to 5120, 8 waves to 10240; GTSfM's cap of 5000 keypoints sits in the first), tier boundaries, batches that mix tiers, the
    LDS-staged path beyond 10240 columns."""
    from gtsfm_amd.runtime import lib as L

    rng = np.random.default_rng(5)
    m = np.array([s[0] for s in shapes], dtype=np.int32)
    n = np.array([s[1] for s in shapes], dtype=np.int32)
    scores = [(rng.standard_normal(s) * 6.0).astype(np.float32) for s in shapes]
    flat = []
    for (mm, nn), sc in zip(shapes, scores):
        ld = (nn + 1 + 3) // 4 * 4
        z = np.full((mm + 1, ld), np.nan, dtype=np.float32)  # padding and dustbins must not be read / are overwritten
        z[:mm, :nn] = sc
        flat.append(z.reshape(-1))
    z_dev = T(np.concatenate(flat)).to(gpu_device)
    ws = torch.empty(int(lib.gtsfm_sinkhorn_workspace_bytes(len(shapes), m.ctypes.data, n.ctypes.data)), dtype=torch.uint8, device=gpu_device)
    u = torch.zeros((len(shapes), int(m.max()) + 1), device=gpu_device)
    v = torch.zeros((len(shapes), int(n.max()) + 1), device=gpu_device)
    L.check(lib.gtsfm_sinkhorn_f32(z_dev.data_ptr(), len(shapes), m.ctypes.data, n.ctypes.data, 1.0, iters, ws.data_ptr(), ws.numel(),
                                   u.data_ptr(), v.data_ptr(), _stream()), "sinkhorn")
    u, v = u.cpu().numpy(), v.cpu().numpy()
    for p, ((mm, nn), sc) in enumerate(zip(shapes, scores)):
        with torch.no_grad():
            ref = sgo.log_optimal_transport(T(sc)[None], torch.tensor(1.0), iters)[0].numpy()
        couplings = np.full((mm + 1, nn + 1), 1.0, dtype=np.float32)
        couplings[:mm, :nn] = sc
        norm = -np.log(np.float32(mm + nn))
        got = couplings + u[p, : mm + 1, None] + v[p, None, : nn + 1] - norm
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4)


@pytest.mark.parametrize("shapes", [[(300, 200), (1, 5), (129, 128), (64, 700)], [(5000, 4800), (2049, 5000)], [(1400, 1300)] * 3],
                         ids=["ragged_small", "cap_super_tiles", "nine_to_eleven_column_groups"])
def test_score_matrices_batched_launch_vs_matmul(lib, gpu_device, shapes):
    """gtsfm_score_matrices_f32: the ragged one-launch score GEMM of a chunk (superglue.py:257-258) against per-pair float64 matmuls, and -- the
    tile ORDER being a launch-geometry decision only -- bit-identical between the row order of rounds 2-5 and round 6's super-tile order for wide
    products (more than 8 column groups) and the 64 x 64 tiling small single products take: every pair again through the stand-alone linear entry
    point, which picks its own tiling and order from its own geometry."""
    from gtsfm_amd.runtime import lib as L

    rng = np.random.default_rng(11)
    m = np.array([s[0] for s in shapes], dtype=np.int32)
    n = np.array([s[1] for s in shapes], dtype=np.int32)
    descs = [(rng.standard_normal((a + b, 256)) / 16.0).astype(np.float32) for a, b in shapes]
    mdesc = T(np.concatenate(descs)).to(gpu_device)
    sizes = [(a + 1) * ((b + 1 + 3) // 4 * 4) for a, b in shapes]
    z = torch.full((sum(sizes),), float("nan"), device=gpu_device)
    ws = torch.empty(int(lib.gtsfm_score_matrices_workspace_bytes(len(shapes))), dtype=torch.uint8, device=gpu_device)
    L.check(lib.gtsfm_score_matrices_f32(mdesc.data_ptr(), len(shapes), m.ctypes.data, n.ctypes.data, 0.0625, z.data_ptr(), ws.data_ptr(), ws.numel(),
                                         _stream()), "score_matrices")
    z = z.cpu().numpy()
    off = 0
    for (a, b), d, size in zip(shapes, descs, sizes):
        ld = (b + 1 + 3) // 4 * 4
        got = z[off : off + size].reshape(a + 1, ld)
        ref = (d[:a].astype(np.float64) @ d[a:].astype(np.float64).T) * 0.0625
        np.testing.assert_allclose(got[:a, :b], ref, rtol=0, atol=2e-6)
        assert np.isnan(got[a]).all() and np.isnan(got[:a, b:]).all()  # the dustbin row / column and the padding are not this launch's to write
        # the same product of ONE pair through the stand-alone linear entry point (same kernel, same tile order rule): bit-identical
        c = torch.empty((a, b), device=gpu_device)
        A, W = T(d[:a]).to(gpu_device), T(d[a:]).to(gpu_device)
        L.check(lib.gtsfm_linear_rowmajor_f32(A.data_ptr(), 256, a, None, 256, W.data_ptr(), 256, None, b, None, c.data_ptr(), b, 0, None, 0, 0.0625, 0, _stream()),
                "linear_rowmajor")
        np.testing.assert_array_equal(c.cpu().numpy(), got[:a, :b])  # (widths that are no multiple of 4 take scalar stores there: same arithmetic)
        off += size


def test_sinkhorn_batch_composition_does_not_change_a_pair(lib, gpu_device):
    """Which sweep kernel a pair takes depends on its own width only: a pair alone, next to a wider pair (which adds the
    workgroup-per-row launches and more register chunks per wave) and next to a narrower one gives the same u, v bit for bit."""
    from gtsfm_amd.runtime import lib as L

    rng = np.random.default_rng(9)

    def run(shapes, scores):
        m = np.array([s[0] for s in shapes], dtype=np.int32)
        n = np.array([s[1] for s in shapes], dtype=np.int32)
        flat = []
        for (mm, nn), sc in zip(shapes, scores):
            z = np.zeros((mm + 1, (nn + 1 + 3) // 4 * 4), dtype=np.float32)
            z[:mm, :nn] = sc
            flat.append(z.reshape(-1))
        z_dev = T(np.concatenate(flat)).to(gpu_device)
        ws = torch.empty(int(lib.gtsfm_sinkhorn_workspace_bytes(len(shapes), m.ctypes.data, n.ctypes.data)), dtype=torch.uint8, device=gpu_device)
        u = torch.zeros((len(shapes), int(m.max()) + 1), device=gpu_device)
        v = torch.zeros((len(shapes), int(n.max()) + 1), device=gpu_device)
        L.check(lib.gtsfm_sinkhorn_f32(z_dev.data_ptr(), len(shapes), m.ctypes.data, n.ctypes.data, 1.0, 7, ws.data_ptr(), ws.numel(),
                                       u.data_ptr(), v.data_ptr(), _stream()), "sinkhorn")
        return u.cpu().numpy(), v.cpu().numpy()

    shapes = [(90, 700), (70, 2600), (50, 4100), (40, 6000)]
    scores = [(rng.standard_normal(s) * 6.0).astype(np.float32) for s in shapes]
    u_all, v_all = run(shapes, scores)
    for q, (mm, nn) in enumerate(shapes):
        u1, v1 = run([shapes[q]], [scores[q]])
        np.testing.assert_array_equal(u_all[q, : mm + 1], u1[0, : mm + 1])
        np.testing.assert_array_equal(v_all[q, : nn + 1], v1[0, : nn + 1])


def test_sinkhorn_nontemporal_reads_are_bit_identical(lib, gpu_device, monkeypatch):
    """Launches whose score matrices exceed the Infinity Cache read them nontemporally (sw_zload<true>); a speed choice per launch, so
    the kernels are forced on for small batches of every tier (GTSFM_SWEEP_NT_MB=0) and compared with the plain reads bit for bit."""
    from gtsfm_amd.runtime import lib as L

    rng = np.random.default_rng(10)
    shapes = [(90, 200), (33, 700), (64, 2048), (70, 2600), (50, 5000), (40, 6000)]
    m = np.array([s[0] for s in shapes], dtype=np.int32)
    n = np.array([s[1] for s in shapes], dtype=np.int32)
    flat = []
    for mm, nn in shapes:
        z = np.zeros((mm + 1, (nn + 1 + 3) // 4 * 4), dtype=np.float32)
        z[:mm, :nn] = (rng.standard_normal((mm, nn)) * 6.0).astype(np.float32)
        flat.append(z.reshape(-1))
    z_host = np.concatenate(flat)

    def run():
        z_dev = T(z_host).to(gpu_device)
        ws = torch.empty(int(lib.gtsfm_sinkhorn_workspace_bytes(len(shapes), m.ctypes.data, n.ctypes.data)), dtype=torch.uint8, device=gpu_device)
        u = torch.zeros((len(shapes), int(m.max()) + 1), device=gpu_device)
        v = torch.zeros((len(shapes), int(n.max()) + 1), device=gpu_device)
        L.check(lib.gtsfm_sinkhorn_f32(z_dev.data_ptr(), len(shapes), m.ctypes.data, n.ctypes.data, 1.0, 9, ws.data_ptr(), ws.numel(),
                                       u.data_ptr(), v.data_ptr(), _stream()), "sinkhorn")
        return u.cpu().numpy(), v.cpu().numpy()

    monkeypatch.setenv("GTSFM_SWEEP_NT_MB", "1e9")
    u_plain, v_plain = run()
    monkeypatch.setenv("GTSFM_SWEEP_NT_MB", "0")
    u_nt, v_nt = run()
    assert np.abs(u_plain).max() > 0
    np.testing.assert_array_equal(u_nt, u_plain)
    np.testing.assert_array_equal(v_nt, v_plain)


def test_superglue_end_to_end_with_nontemporal_sinkhorn_reads(sg_engine, monkeypatch):
    """The score matrices the Sinkhorn kernels read nontemporally were written by the score GEMM one launch earlier (the benchmark's
    32-pair chunks are in that regime): forced on for a small ragged batch, matches, scores and the transport matrix are the plain
    reads' bit for bit."""
    specs = [(300, 257, (480, 640), (480, 640), 31), (96, 2300, (240, 320), (600, 400), 32)]
    feats = [synthetic.synthetic_pair_features(a, b, s0, s1, seed=sd) for a, b, s0, s1, sd in specs]
    dev = sg_engine.device
    kp = T(np.concatenate([np.concatenate([f[0], f[3]]) for f in feats])).to(dev)
    sc = T(np.concatenate([np.concatenate([f[1], f[4]]) for f in feats])).to(dev)
    de = T(np.concatenate([np.concatenate([f[2], f[5]]) for f in feats])).to(dev)
    n0, n1 = [s[0] for s in specs], [s[1] for s in specs]
    hw = [[s[2][0], s[2][1], s[3][0], s[3][1]] for s in specs]

    def run(mb):
        monkeypatch.setenv("GTSFM_SWEEP_NT_MB", mb)
        out = sg_engine.match_batch(kp, sc, de, n0, n1, hw, sinkhorn_iterations=20, return_ot=True)
        ot, off, mats = out["ot"].cpu().numpy(), 0, []
        for a, b in zip(n0, n1):  # the row stride is padded to 4 floats; the padding is never written
            ld = (b + 1 + 3) // 4 * 4
            mats.append(ot[off : off + (a + 1) * ld].reshape(a + 1, ld)[:, : b + 1].copy())
            off += (a + 1) * ld
        return [out["matches"].cpu().numpy(), out["mscores"].cpu().numpy(), *mats]

    plain = run("1e9")
    assert (plain[0] >= 0).sum() > 50
    for _ in range(3):
        for got, want in zip(run("0"), plain):
            np.testing.assert_array_equal(got, want)


def test_superglue_plugin_contract(gpu_device, sg_sd, tmp_path):
    """SuperGlueMatcher.match vs the restated reference wrapper (gtsfm/frontend/matcher/superglue_matcher.py:75-113)
    and the reference's contract tests (tests/frontend/matcher/test_matcher_base.py:51-107,
    test_superglue_matcher.py:24-41): dtype uint32, indices in range, one-to-one, empty input, error behaviour."""
    from gtsfm_amd.common.keypoints import Keypoints
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    path = tmp_path / "superglue_outdoor.pth"
    torch.save(sg_sd, str(path))
    matcher = SuperGlueMatcher(weights_path=path)
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(200, 180, (480, 640), (360, 500), seed=41)
    kp0, kp1 = Keypoints(k0, responses=s0), Keypoints(k1, responses=s1)
    m = matcher.match(kp0, kp1, d0, d1, im_shape_i1=(480, 640, 3), im_shape_i2=(360, 500, 3))
    ref = sgo.match(sg_sd, k0, k1, s0, s1, d0, d1, (480, 640, 3), (360, 500, 3), sinkhorn_iterations=20)
    assert isinstance(m, np.ndarray) and m.dtype == np.uint32 and m.shape[1] == 2
    np.testing.assert_array_equal(m, ref)
    assert m.shape[0] > 20 and m[:, 0].max() < 200 and m[:, 1].max() < 180
    assert len(set(m[:, 0].tolist())) == len(m) == len(set(m[:, 1].tolist()))
    empty = matcher.match(Keypoints(np.zeros((0, 2), np.float32), responses=np.zeros(0, np.float32)), kp1,
                          np.zeros((0, 256), np.float32), d1, (480, 640, 3), (360, 500, 3))
    assert empty.shape == (0, 2) and empty.dtype == np.uint32
    with pytest.raises(ValueError):
        matcher.match(Keypoints(k0), kp1, d0, d1, (480, 640, 3), (360, 500, 3))
    with pytest.raises(Exception):
        matcher.match(kp0, kp1, d0[:, :128], d1[:, :128], (480, 640, 3), (360, 500, 3))


def test_superglue_full_size_properties(sg_engine):
    """BASELINE config-4 scale (N = 2048 per image, 100 Sinkhorn iterations): size-independent properties --
    determinism, one-to-one mutual consistency, score thresholds, planted correspondences recovered."""
    n = 2048
    k0, s0, d0, k1, s1, d1, gt = synthetic.synthetic_pair_features(n, n, (1024, 1024), (1024, 1024), seed=5)
    a = sg_engine.match_pair(k0, s0, d0, k1, s1, d1, (1024, 1024), (1024, 1024), sinkhorn_iterations=100)
    b = sg_engine.match_pair(k0, s0, d0, k1, s1, d1, (1024, 1024), (1024, 1024), sinkhorn_iterations=100)
    for key in a:
        np.testing.assert_array_equal(a[key], b[key])
    m0, m1 = a["matches0"], a["matches1"]
    v0 = m0 > -1
    assert v0.sum() > 500
    assert np.array_equal(m1[m0[v0]], np.flatnonzero(v0))  # mutual
    assert len(set(m0[v0].tolist())) == v0.sum()  # one-to-one
    assert (a["matching_scores0"][v0] > 0.2).all() and (a["matching_scores0"] >= 0).all() and (a["matching_scores0"] <= 1.0 + 1e-5).all()
    planted = gt > -1
    assert (m0[planted] == gt[planted]).mean() > 0.9


# ------------------------------------------------------------------------------------------------------------------
# LightGlue (parity unpinned: HIP path == oracle restatement)
# ------------------------------------------------------------------------------------------------------------------

LG_CASES = [
    # (weight kwargs, n0, n1, pruning threshold, expects early stop, expects pruning)
    ({}, 300, 280, 1536, False, False),
    ({}, 300, 280, -1, False, False),
    ({"conf_bias": 2.0, "conf_gain": 4.0}, 300, 280, None, True, False),
    ({"conf_bias": 1.0, "conf_gain": 6.0, "match_bias": -2.0, "match_gain": 8.0}, 300, 280, -1, False, True),
    ({"conf_bias": 1.0, "conf_gain": 6.0, "match_bias": -2.0, "match_gain": 8.0}, 700, 650, 256, False, True),
    ({}, 1, 3, 1536, False, False),
    ({}, 129, 128, 1536, False, False),
]


@pytest.mark.parametrize("kw,n0,n1,pth,early,pruned", LG_CASES)
def test_lightglue_matches_oracle(gpu_device, kw, n0, n1, pth, early, pruned):
    from gtsfm_amd.runtime.matcher_engine import LightGlueEngine

    sd = synthetic.synthetic_lightglue_state_dict(**kw)
    eng = LightGlueEngine(sd, gpu_device)
    k0, _, d0, k1, _, d1, _ = synthetic.synthetic_pair_features(n0, n1, (480, 640), (400, 600), seed=3)
    res = eng.match_pair(k0, d0, k1, d1, (480, 640), (400, 600), pruning_threshold=pth)
    with torch.no_grad():
        ora = lgo.lightglue_forward(sd, T(k0)[None], T(k1)[None], T(d0)[None], T(d1)[None], (480, 640), (400, 600),
                                    pruning_threshold=pth, return_intermediates=True)
    assert res["stop"] == ora["stop"] and (ora["stop"] < 9) == early
    kept = (ora["ind0"].shape[1], ora["ind1"].shape[1])
    assert tuple(res["kept"].tolist()) == kept and ((kept[0] < n0) or (kept[1] < n1)) == pruned
    np.testing.assert_array_equal(res["matches0"], ora["matches0"][0].numpy())
    np.testing.assert_array_equal(res["matches1"], ora["matches1"][0].numpy())
    np.testing.assert_array_equal(res["matches"], ora["matches"].numpy())
    assert res["matches"].dtype == np.int64
    np.testing.assert_allclose(res["matching_scores0"], ora["matching_scores0"][0].numpy(), rtol=0, atol=SCORE_TOL)
    np.testing.assert_allclose(res["matching_scores1"], ora["matching_scores1"][0].numpy(), rtol=0, atol=SCORE_TOL)
    np.testing.assert_allclose(res["scores"], ora["scores"].numpy(), rtol=0, atol=SCORE_TOL)


def test_lightglue_batch_mixed_depths(gpu_device):
    """Pairs that stop at different layers and prune differently share one launch sequence (device-side control flow);
    each must equal its stand-alone result."""
    from gtsfm_amd.runtime.matcher_engine import LightGlueEngine

    sd = synthetic.synthetic_lightglue_state_dict(conf_bias=1.2, conf_gain=5.0, match_bias=-1.0, match_gain=8.0)
    eng = LightGlueEngine(sd, gpu_device)
    specs = [(300, 280, 3, 0.6), (150, 400, 4, 0.0), (513, 511, 5, 0.9), (64, 64, 6, 0.3)]
    feats = [synthetic.synthetic_pair_features(a, b, (480, 640), (480, 640), overlap=ov, seed=sdd) for a, b, sdd, ov in specs]
    dev = eng.device
    kp = T(np.concatenate([np.concatenate([f[0], f[3]]) for f in feats])).to(dev)
    de = T(np.concatenate([np.concatenate([f[2], f[5]]) for f in feats])).to(dev)
    n0, n1 = [s[0] for s in specs], [s[1] for s in specs]
    out = eng.match_batch(kp, de, n0, n1, [[480, 640, 480, 640]] * 4, pruning_threshold=100)
    m, ms = out["matches"].cpu().numpy(), out["mscores"].cpu().numpy()
    stops = out["stop"].cpu().numpy().tolist()
    row = 0
    singles = []
    for (a, b, _, _), f in zip(specs, feats):
        single = eng.match_pair(f[0], f[2], f[3], f[5], (480, 640), (480, 640), pruning_threshold=100)
        singles.append(single["stop"])
        np.testing.assert_array_equal(m[row : row + a], single["matches0"])
        np.testing.assert_array_equal(m[row + a : row + a + b], single["matches1"])
        np.testing.assert_array_equal(ms[row : row + a], single["matching_scores0"])
        row += a + b
    assert stops == singles


def test_lightglue_batch_mixed_depths_and_widths_beyond_2048_keypoints(gpu_device):
    """the ragged / pruned path at the sizes production hits. 28 pairs of 8 views from four canvases of different texture
    scale, top-2304 keypoints (past the one-wave sweeps' 2048 columns and LightGlue's 1536-keypoint pruning threshold), bench.py's ADAPTIVE_HEADS: pairs
    leave ONE launch sequence at different layers and BOTH images of a pair are pruned, to different widths. Every checked pair must equal its stand-alone
    result bit for bit (one pair per distinct stop layer), and two of them the oracle (stop layer, match indices, scores within 1e-4)."""
    import bench
    from gtsfm_amd import parallel
    from gtsfm_amd.runtime.matcher_engine import LightGlueEngine
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    k, n, hw = 2304, 8, (1024, 1024)
    sd = synthetic.synthetic_lightglue_state_dict(**bench.ADAPTIVE_HEADS)
    eng = LightGlueEngine(sd, gpu_device)
    pipe = FrontEndPipeline(SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device), eng, max_keypoints=k, pair_chunk=32, use_graphs=False)
    feats = pipe.detect(torch.from_numpy(synthetic.synthetic_mixed_scene(n, *hw, canvases=4)).to(gpu_device))
    assert feats["count"].tolist() == [k] * n
    pairs = parallel.exhaustive_pairs(n)
    res = pipe.match(feats, pairs, [hw] * n)
    stop = torch.cat([r["stop"] for r in res]).cpu().numpy()
    kept = torch.cat([r["kept"] for r in res]).cpu().numpy().reshape(-1, 2)
    print("MIXED stop layers", np.bincount(stop, minlength=10)[1:].tolist(), "kept share min / median / max", kept.min() / k, np.median(kept) / k, kept.max() / k)
    assert len(set(stop.tolist())) >= 3, stop  # pairs leave at different depths ...
    assert ((kept[:, 0] < k) & (kept[:, 1] < k)).sum() >= 5 and (kept[:, 0] != kept[:, 1]).any()  # ... and both images are pruned, to different widths
    xy, de = feats["xy"].cpu().numpy(), feats["descriptors"].cpu().numpy()
    flat = [(q, r) for r in res for q in range(len(r["pairs"]))]
    first_of_layer = {}
    for idx, (q, r) in enumerate(flat):
        first_of_layer.setdefault(int(stop[idx]), (idx, q, r))
    for checked, (layer, (idx, q, r)) in enumerate(sorted(first_of_layer.items())):
        i, j = r["pairs"][q]
        row = sum(a + b for a, b in zip(r["n0"][:q], r["n1"][:q]))
        m = r["matches"][row : row + 2 * k].cpu().numpy()
        ms = r["mscores"][row : row + 2 * k].cpu().numpy()
        single = eng.match_pair(xy[i], de[i], xy[j], de[j], hw, hw)
        assert single["stop"] == layer
        np.testing.assert_array_equal(m[:k], single["matches0"])
        np.testing.assert_array_equal(m[k:], single["matches1"])
        np.testing.assert_array_equal(ms[:k], single["matching_scores0"])
        np.testing.assert_array_equal(single["kept"].reshape(-1), kept[idx])
        if checked < 2:
            with torch.no_grad():
                ora = lgo.lightglue_forward(sd, T(xy[i])[None], T(xy[j])[None], T(de[i])[None], T(de[j])[None], hw, hw)
            assert int(ora["stop"]) == layer
            np.testing.assert_array_equal(m[:k], ora["matches0"][0].numpy())
            np.testing.assert_allclose(ms[:k], ora["matching_scores0"][0].numpy(), rtol=0, atol=SCORE_TOL)


@pytest.mark.parametrize("matcher", ["superglue", "lightglue"])
def test_batch_mixing_sweep_tiers_equals_single_pairs(gpu_device, matcher):
    """Score matrices of one batch on both sides of the 2048-column limit of the wave-per-row sweeps: the wide pairs take the
    workgroup-per-row kernels, the narrow ones the wave-per-row kernels, inside the same launch sequence; every pair equals its
    stand-alone result bit for bit, and the wide ones equal the oracle."""
    from gtsfm_amd.runtime import matcher_engine as ME

    specs = [(300, 2600, 51), (2600, 300, 52), (120, 100, 53), (2050, 2049, 54)]
    feats = [synthetic.synthetic_pair_features(a, b, (768, 1024), (768, 1024), seed=sdd) for a, b, sdd in specs]
    kp = T(np.concatenate([np.concatenate([f[0], f[3]]) for f in feats])).to(gpu_device)
    sc = T(np.concatenate([np.concatenate([f[1], f[4]]) for f in feats])).to(gpu_device)
    de = T(np.concatenate([np.concatenate([f[2], f[5]]) for f in feats])).to(gpu_device)
    n0, n1 = [s[0] for s in specs], [s[1] for s in specs]
    hw = [[768, 1024, 768, 1024]] * len(specs)
    if matcher == "superglue":
        sd = synthetic.synthetic_superglue_state_dict(num_layers=2)
        eng = ME.SuperGlueEngine(sd, gpu_device)
        out = eng.match_batch(kp, sc, de, n0, n1, hw, sinkhorn_iterations=20)
        single = lambda f: eng.match_pair(f[0], f[1], f[2], f[3], f[4], f[5], (768, 1024), (768, 1024), sinkhorn_iterations=20)  # noqa: E731
    else:
        sd = synthetic.synthetic_lightglue_state_dict(num_layers=2, match_bias=-2.0, match_gain=30.0)
        eng = ME.LightGlueEngine(sd, gpu_device)
        out = eng.match_batch(kp, de, n0, n1, hw, pruning_threshold=None)
        single = lambda f: eng.match_pair(f[0], f[2], f[3], f[5], (768, 1024), (768, 1024), pruning_threshold=None)  # noqa: E731
    m, ms = out["matches"].cpu().numpy(), out["mscores"].cpu().numpy()
    row = 0
    for q, ((a, b, _), f) in enumerate(zip(specs, feats)):
        one = single(f)
        np.testing.assert_array_equal(m[row : row + a], one["matches0"])
        np.testing.assert_array_equal(m[row + a : row + a + b], one["matches1"])
        np.testing.assert_array_equal(ms[row : row + a], one["matching_scores0"])
        np.testing.assert_array_equal(ms[row + a : row + a + b], one["matching_scores1"])
        row += a + b
        if q < 2:
            with torch.no_grad():
                if matcher == "superglue":
                    ora = sgo.superglue_forward(sd, T(f[0])[None], T(f[3])[None], T(f[1])[None], T(f[4])[None], T(f[2]).T[None].contiguous(),
                                                T(f[5]).T[None].contiguous(), (768, 1024), (768, 1024), sinkhorn_iterations=20)
                else:
                    ora = lgo.lightglue_forward(sd, T(f[0])[None], T(f[3])[None], T(f[2])[None], T(f[5])[None], (768, 1024), (768, 1024),
                                                pruning_threshold=None, return_intermediates=True)
            np.testing.assert_array_equal(one["matches0"], ora["matches0"][0].numpy())
            np.testing.assert_array_equal(one["matches1"], ora["matches1"][0].numpy())
            np.testing.assert_allclose(one["matching_scores0"], ora["matching_scores0"][0].numpy(), rtol=0, atol=SCORE_TOL)
            assert (one["matches0"] > -1).sum() > 20


@pytest.mark.parametrize("matcher", ["superglue", "lightglue"])
def test_extraction_tiering_does_not_change_matches(gpu_device, monkeypatch, matcher):
    """Arg-maxima do not depend on how a row is cut into slices, so the match extraction's launcher is free to pick its tiers (launch_extract_matches):
    by default one wave per row up to GTSFM_EXTRACT_NARROW_COLS = 1024 columns, four waves per row for 1025 .. 5120, eight beyond;
    GTSFM_EXTRACT_WAVES=8 sends every row above the narrow bound to the eight-wave tier, and GTSFM_EXTRACT_NARROW_COLS=2048 restores round 4's
    one-wave bound (the 8-chunk narrow instantiation). All three must give the same matches and scores bit for bit (a 1500-, a 2600- and a
    5000-column pair in one batch: the 1500-column pair changes tier under the second switch, the wider two under the first)."""
    from gtsfm_amd.runtime import matcher_engine as ME

    specs = [(300, 2600, 41), (200, 5000, 42), (250, 1500, 43)]
    feats = [synthetic.synthetic_pair_features(a, b, (480, 640), (480, 640), seed=sd) for a, b, sd in specs]
    kp = T(np.concatenate([np.concatenate([f[0], f[3]]) for f in feats])).to(gpu_device)
    sc = T(np.concatenate([np.concatenate([f[1], f[4]]) for f in feats])).to(gpu_device)
    de = T(np.concatenate([np.concatenate([f[2], f[5]]) for f in feats])).to(gpu_device)
    n0, n1, hw = [s[0] for s in specs], [s[1] for s in specs], [[480, 640, 480, 640]] * len(specs)
    if matcher == "superglue":
        eng = ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(num_layers=4), gpu_device)
        call = lambda: eng.match_batch(kp, sc, de, n0, n1, hw, sinkhorn_iterations=5)  # noqa: E731
    else:
        eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=2, match_bias=-2.0, match_gain=30.0), gpu_device)
        call = lambda: eng.match_batch(kp, de, n0, n1, hw, pruning_threshold=None)  # noqa: E731
    got = {}
    for name, env in (("default", {}), ("eight_waves", {"GTSFM_EXTRACT_WAVES": "8"}), ("narrow_2048", {"GTSFM_EXTRACT_NARROW_COLS": "2048"})):
        monkeypatch.delenv("GTSFM_EXTRACT_WAVES", raising=False)
        monkeypatch.delenv("GTSFM_EXTRACT_NARROW_COLS", raising=False)
        for key, value in env.items():
            monkeypatch.setenv(key, value)
        out = call()
        got[name] = (out["matches"].cpu().numpy(), out["mscores"].cpu().numpy())
    assert (got["default"][0] >= 0).sum() > 20
    for name in ("eight_waves", "narrow_2048"):
        np.testing.assert_array_equal(got["default"][0], got[name][0])
        np.testing.assert_array_equal(got["default"][1], got[name][1])


@pytest.mark.parametrize("matcher", ["superglue", "lightglue"])
def test_eight_wave_sweep_tier_vs_oracle(gpu_device, matcher):
    """Keypoint sets beyond 5120 (a user raising max_keypoints above GTSfM's default): the score matrix's rows are shared by the 8
    waves of a 512-thread workgroup (Sinkhorn / double softmax / extraction); one layer keeps the CPU oracle to seconds."""
    from gtsfm_amd.runtime import matcher_engine as ME

    n0, n1 = 260, 5300
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(n0, n1, (1024, 1024), (1024, 1024), seed=61)
    if matcher == "superglue":
        sd = synthetic.synthetic_superglue_state_dict(num_layers=1)
        res = ME.SuperGlueEngine(sd, gpu_device).match_pair(k0, s0, d0, k1, s1, d1, (1024, 1024), (1024, 1024), sinkhorn_iterations=20)
        with torch.no_grad():
            ora = sgo.superglue_forward(sd, T(k0)[None], T(k1)[None], T(s0)[None], T(s1)[None], T(d0).T[None].contiguous(), T(d1).T[None].contiguous(),
                                        (1024, 1024), (1024, 1024), sinkhorn_iterations=20)
    else:
        sd = synthetic.synthetic_lightglue_state_dict(num_layers=1, match_bias=-2.0, match_gain=30.0)
        res = ME.LightGlueEngine(sd, gpu_device).match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024), pruning_threshold=None)
        with torch.no_grad():
            ora = lgo.lightglue_forward(sd, T(k0)[None], T(k1)[None], T(d0)[None], T(d1)[None], (1024, 1024), (1024, 1024), pruning_threshold=None,
                                        return_intermediates=True)
    np.testing.assert_array_equal(res["matches0"], ora["matches0"][0].numpy())
    np.testing.assert_array_equal(res["matches1"], ora["matches1"][0].numpy())
    np.testing.assert_allclose(res["matching_scores0"], ora["matching_scores0"][0].numpy(), rtol=0, atol=SCORE_TOL)
    np.testing.assert_allclose(res["matching_scores1"], ora["matching_scores1"][0].numpy(), rtol=0, atol=SCORE_TOL)
    assert (res["matches0"] > -1).sum() > 10


def test_lightglue_plugin_contract(gpu_device, tmp_path):
    """LightGlueMatcher.match (gtsfm/frontend/matcher/lightglue_matcher.py:43-112): (K,2) int64, in range,
    one-to-one, empty input -> empty result, ValueError without responses."""
    from gtsfm_amd.common.keypoints import Keypoints
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher

    sd = synthetic.synthetic_lightglue_state_dict()
    path = tmp_path / "superpoint_lightglue.pth"
    torch.save(sd, str(path))
    matcher = LightGlueMatcher(features="superpoint", weights_path=path)
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(220, 190, (480, 640), (360, 500), seed=43)
    kp0, kp1 = Keypoints(k0, responses=s0), Keypoints(k1, responses=s1)
    m = matcher.match(kp0, kp1, d0, d1, im_shape_i1=(480, 640, 3), im_shape_i2=(360, 500, 3))
    ref = lgo.match(sd, k0, k1, d0, d1, (480, 640, 3), (360, 500, 3))
    assert m.dtype == np.int64 and m.shape[1] == 2
    np.testing.assert_array_equal(m, ref)
    assert m.shape[0] > 20 and m[:, 0].max() < 220 and m[:, 1].max() < 190 and np.all(np.diff(m[:, 0]) > 0)
    assert len(set(m[:, 1].tolist())) == len(m)
    empty = matcher.match(Keypoints(np.zeros((0, 2), np.float32), responses=np.zeros(0, np.float32)), kp1,
                          np.zeros((0, 256), np.float32), d1, (480, 640, 3), (360, 500, 3))
    assert empty.shape == (0, 2)
    with pytest.raises(ValueError):
        matcher.match(Keypoints(k0), kp1, d0, d1, (480, 640, 3), (360, 500, 3))


@pytest.mark.parametrize("which", ["superglue", "lightglue"])
def test_per_call_image_cache_is_bit_identical_and_counts_hits(gpu_device, sg_sd, which):
    """The per-call path keeps every image's uploaded keypoints and the output of the matcher's per-image first block on the device
    (``_MatcherBase._image_entries``): pairs (A,B), (A,C), (B,C) cost 2 + 1 + 0 uploads instead of the reference's 6
    (superglue_matcher.py:75-102 uploads both images per pair). Results must equal the uncached calls bit for bit; an array that is
    modified in place (a sampled row) is a new image; ``release_lanes()`` empties the cache."""
    from gtsfm_amd.runtime import matcher_engine as ME

    imgs = []
    for seed, n in enumerate((900, 1100, 640)):
        k, s, d, *_ = synthetic.synthetic_pair_features(n, 8, (480, 640), (480, 640), seed=200 + seed)
        imgs.append((k, s, d))
    if which == "superglue":
        cached, plain = ME.SuperGlueEngine(sg_sd, gpu_device), ME.SuperGlueEngine(sg_sd, gpu_device)
        call = lambda e, a, b: e.match_pair(a[0], a[1], a[2], b[0], b[1], b[2], (480, 640), (480, 640))  # noqa: E731
    else:
        sd = synthetic.synthetic_lightglue_state_dict()
        cached, plain = ME.LightGlueEngine(sd, gpu_device), ME.LightGlueEngine(sd, gpu_device)
        call = lambda e, a, b: e.match_pair(a[0], a[2], b[0], b[2], (480, 640), (480, 640))  # noqa: E731
    plain.image_cache_capacity = 0
    assert cached.image_cache_capacity == 64
    expect = [(2, 0), (3, 1), (3, 3)]  # cumulative (misses, hits)
    for (i, j), (miss, hit) in zip([(0, 1), (0, 2), (1, 2)], expect):
        a, b = call(cached, imgs[i], imgs[j]), call(plain, imgs[i], imgs[j])
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
            np.testing.assert_array_equal(a[key], b[key])
        assert (cached.image_cache_misses, cached.image_cache_hits) == (miss, hit)
    assert (plain.image_cache_misses, plain.image_cache_hits) == (0, 0)
    # the same buffer with other content is another image
    imgs[0][2][-1, 5] += 0.25
    a, b = call(cached, imgs[0], imgs[1]), call(plain, imgs[0], imgs[1])
    np.testing.assert_array_equal(a["matching_scores0"], b["matching_scores0"])
    assert cached.image_cache_misses == 4 and len(cached._image_cache) == 4
    cached.release_lanes()
    assert len(cached._image_cache) == 0 and cached._workspace is None
    np.testing.assert_array_equal(call(cached, imgs[1], imgs[2])["matches0"], call(plain, imgs[1], imgs[2])["matches0"])


@pytest.mark.parametrize("which", ["superglue", "lightglue"])
def test_image_cache_cannot_serve_an_array_overwritten_in_place(gpu_device, sg_sd, which):
    """the cache's lookup key samples 36 rows; an image whose descriptors are overwritten IN PLACE everywhere else keeps that
    key. Every hit is therefore verified against a checksum over all bytes of the arrays while the GPU works on the pair
    (``_entries_still_valid``): the stale entry is dropped, the pair is redone from the arrays, and the call returns what an engine without
    a cache returns for the new content -- by default, with no switch."""
    from gtsfm_amd.runtime import matcher_engine as ME

    n = 1500
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(n, 1400, (480, 640), (480, 640), seed=310)
    if which == "superglue":
        cached, plain = ME.SuperGlueEngine(sg_sd, gpu_device), ME.SuperGlueEngine(sg_sd, gpu_device)
        call = lambda e: e.match_pair(k0, s0, d0, k1, s1, d1, (480, 640), (480, 640))  # noqa: E731
        arrays = (k0, s0, d0)
    else:
        sd = synthetic.synthetic_lightglue_state_dict()
        cached, plain = ME.LightGlueEngine(sd, gpu_device), ME.LightGlueEngine(sd, gpu_device)
        call = lambda e: e.match_pair(k0, d0, k1, d1, (480, 640), (480, 640))  # noqa: E731
        arrays = (k0, d0)
    plain.image_cache_capacity = 0
    before = call(cached)
    assert (cached.image_cache_misses, cached.image_cache_hits, cached.image_cache_stale) == (2, 0, 0)
    key_before = ME._MatcherBase._image_key(arrays, (480, 640))
    sampled = np.unique(np.concatenate([np.arange(10), np.arange(n - 10, n), np.linspace(0, n - 1, 16).astype(np.int64)]))
    rest = np.setdiff1d(np.arange(n), sampled)
    d0[rest] = d0[rest][::-1].copy()  # every row the lookup key does not sample: reversed in place (unit rows stay unit rows)
    assert ME._MatcherBase._image_key(arrays, (480, 640)) == key_before  # the cheap key cannot tell
    after, fresh = call(cached), call(plain)
    assert cached.image_cache_stale == 1 and cached.image_cache_misses == 3  # found stale, redone from the arrays
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
        np.testing.assert_array_equal(after[key], fresh[key])
    assert not np.array_equal(after["matches0"], before["matches0"])
    again = call(cached)  # the redone entry is a valid one
    assert cached.image_cache_stale == 1 and cached.image_cache_misses == 3
    np.testing.assert_array_equal(again["matches0"], fresh["matches0"])


@pytest.mark.parametrize("which", ["superglue", "lightglue"])
def test_threads_matching_pairs_that_share_an_image_are_bit_identical(gpu_device, sg_sd, which):
    """an image entry made on one lane's stream is consumed on another lane's as soon as it is in the cache; the event
    consumers wait for must cover the entry's keypoint / score clones as well as the per-image block's output. Six pairs over four images
    -- every image in three pairs -- from three threads at once, from a cold cache, ten times: every result equals the single-threaded,
    uncached call bit for bit."""
    import threading

    from gtsfm_amd.runtime import matcher_engine as ME

    imgs = []
    for seed, n in enumerate((2100, 1800, 2048, 1500)):
        k, s, d, *_ = synthetic.synthetic_pair_features(n, 8, (480, 640), (480, 640), seed=400 + seed)
        imgs.append((k, s, d))
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    if which == "superglue":
        eng, plain = ME.SuperGlueEngine(sg_sd, gpu_device), ME.SuperGlueEngine(sg_sd, gpu_device)
        call = lambda e, a, b: e.match_pair(a[0], a[1], a[2], b[0], b[1], b[2], (480, 640), (480, 640))  # noqa: E731
    else:
        sd = synthetic.synthetic_lightglue_state_dict()
        eng, plain = ME.LightGlueEngine(sd, gpu_device), ME.LightGlueEngine(sd, gpu_device)
        call = lambda e, a, b: e.match_pair(a[0], a[2], b[0], b[2], (480, 640), (480, 640))  # noqa: E731
    plain.image_cache_capacity = 0
    want = {p: call(plain, imgs[p[0]], imgs[p[1]]) for p in pairs}
    for rep in range(10):
        eng.release_lanes()  # cold cache: every round produces the entries afresh on whichever lane gets there first
        got, errors = {}, []

        def worker(tid):
            try:
                for p in pairs[tid::3]:
                    got[p] = call(eng, imgs[p[0]], imgs[p[1]])
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for p in pairs:
            for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
                np.testing.assert_array_equal(got[p][key], want[p][key], err_msg=f"round {rep}, pair {p}, {key}")
    assert eng.image_cache_stale == 0


@pytest.mark.parametrize("which", ["superglue", "lightglue"])
def test_plugin_match_from_several_threads_equals_one_at_a_time(gpu_device, sg_sd, tmp_path, which):
    """GTSfM's ``--threads_per_worker`` (gtsfm/runner.py:155,436) lets several Dask threads call ``match`` on ONE scattered matcher
    object at once. The engine hands every concurrent call a lane of its own (workspace, staging buffers, stream; shared weights):
    ragged pairs matched from three threads, repeatedly, give exactly the arrays the same calls give one after the other, and no more
    than ``max_lanes`` lanes ever exist."""
    import threading

    from gtsfm_amd.common.keypoints import Keypoints
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    if which == "superglue":
        torch.save(sg_sd, str(tmp_path / "sg.pth"))
        matcher = SuperGlueMatcher(weights_path=tmp_path / "sg.pth")
    else:
        torch.save(synthetic.synthetic_lightglue_state_dict(), str(tmp_path / "lg.pth"))
        matcher = LightGlueMatcher(features="superpoint", weights_path=tmp_path / "lg.pth")
    jobs = []
    for seed, (n0, n1) in enumerate([(220, 190), (700, 40), (1, 300), (513, 512), (90, 2100), (300, 300), (64, 65), (1200, 900), (33, 7)]):
        k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(n0, n1, (480, 640), (360, 500), seed=70 + seed)
        jobs.append((Keypoints(k0, responses=s0), Keypoints(k1, responses=s1), d0, d1, (480, 640, 3), (360, 500, 3)))
    serial = [matcher.match(*j) for j in jobs]
    assert sum(len(m) for m in serial) > 100
    results, errors = {}, []

    def worker(tid):
        try:
            for rep in range(3):
                for q in range(tid, len(jobs), 3):
                    results[(rep, q)] = matcher.match(*jobs[q])
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 3 * len(jobs)
    for (rep, q), m in results.items():
        np.testing.assert_array_equal(m, serial[q])
    engine = matcher._model
    assert 1 <= len(engine._lanes) <= engine.max_lanes


def test_lightglue_full_size_properties(gpu_device):
    """BASELINE config-3 scale (N = 2048 per image): determinism, mutual one-to-one matches, planted
    correspondences recovered, early stopping never changes which layer's assignment head is reported."""
    from gtsfm_amd.runtime.matcher_engine import LightGlueEngine

    eng = LightGlueEngine(synthetic.synthetic_lightglue_state_dict(), gpu_device)
    n = 2048
    k0, _, d0, k1, _, d1, gt = synthetic.synthetic_pair_features(n, n, (1024, 1024), (1024, 1024), seed=9)
    a = eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024))
    b = eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024))
    for key in ("matches0", "matches1", "matching_scores0", "matches"):
        np.testing.assert_array_equal(a[key], b[key])
    m0, m1 = a["matches0"], a["matches1"]
    v0 = m0 > -1
    assert v0.sum() > 500 and np.array_equal(m1[m0[v0]], np.flatnonzero(v0))
    assert (a["matching_scores0"][v0] > 0.1).all()
    planted = gt > -1
    assert (m0[planted] == gt[planted]).mean() > 0.9


def test_resident_pipeline_equals_plugins(gpu_device, sg_sd, tmp_path):
    """GPU-resident detect -> top-k -> match pipeline (features never leave HBM) vs the per-image / per-pair plugin
    calls that go through host numpy like the reference's Dask tasks."""
    from gtsfm_amd.common.image import Image
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher
    from gtsfm_amd.runtime.matcher_engine import SuperGlueEngine
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    sp_sd = synthetic.synthetic_superpoint_state_dict()
    torch.save(sp_sd, str(tmp_path / "sp.pth"))
    torch.save(sg_sd, str(tmp_path / "sg.pth"))
    imgs = np.stack([synthetic.synthetic_gray_image(160, 200, s) for s in (61, 62, 63)])
    pipe = FrontEndPipeline(SuperPointEngine(sp_sd, gpu_device), SuperGlueEngine(sg_sd, gpu_device), max_keypoints=150, pair_chunk=2)
    feats = pipe.detect(torch.from_numpy(imgs).to(gpu_device))
    pairs = [(0, 1), (0, 2), (1, 2)]
    got = pipe.matches_to_numpy(pipe.match(feats, pairs, [(160, 200)] * 3), dtype=np.uint32)
    det = SuperPointDetectorDescriptor(max_keypoints=150, weights_path=tmp_path / "sp.pth")
    matcher = SuperGlueMatcher(weights_path=tmp_path / "sg.pth")
    host = [det.detect_and_describe(Image(value_array=im)) for im in imgs]
    for i, j in pairs:
        # the plugin's top-k order is argpartition's; sort both keypoint sets into detection order to compare
        oi = np.lexsort((host[i][0].coordinates[:, 0], host[i][0].coordinates[:, 1]))
        oj = np.lexsort((host[j][0].coordinates[:, 0], host[j][0].coordinates[:, 1]))
        ki = host[i][0].extract_indices(oi)
        kj = host[j][0].extract_indices(oj)
        np.testing.assert_array_equal(ki.coordinates, feats["xy"][i, : len(ki)].cpu().numpy())
        ref = matcher.match(ki, kj, host[i][1][oi], host[j][1][oj], (160, 200, 1), (160, 200, 1))
        np.testing.assert_array_equal(got[(i, j)], ref)


def test_batched_correspondence_generator(gpu_device, sg_sd, tmp_path):
    """BatchedDetDescCorrespondenceGenerator.generate_correspondences (contract of
    gtsfm/frontend/correspondence_generator/det_desc_correspondence_generator.py:33-87) vs per-image / per-pair plugin
    calls: mixed image sizes, one masked image, SuperGlue and LightGlue."""
    from gtsfm_amd.common.image import Image
    from gtsfm_amd.frontend.correspondence_generator.batched_det_desc_correspondence_generator import (
        BatchedDetDescCorrespondenceGenerator,
    )
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    torch.save(synthetic.synthetic_superpoint_state_dict(), str(tmp_path / "sp.pth"))
    torch.save(sg_sd, str(tmp_path / "sg.pth"))
    torch.save(synthetic.synthetic_lightglue_state_dict(), str(tmp_path / "lg.pth"))
    mask = np.zeros((120, 176), dtype=np.uint8)
    mask[10:100, 20:160] = 1
    images = [
        Image(value_array=synthetic.synthetic_gray_image(160, 200, 71)),
        Image(value_array=synthetic.synthetic_gray_image(160, 200, 72)),
        Image(value_array=np.stack([synthetic.synthetic_gray_image(120, 176, 73)] * 3, -1), mask=mask),
        Image(value_array=synthetic.synthetic_gray_image(120, 176, 74)),
    ]
    graph = [(0, 1), (0, 2), (1, 3), (2, 3)]
    det = SuperPointDetectorDescriptor(max_keypoints=120, weights_path=tmp_path / "sp.pth")
    for matcher in (SuperGlueMatcher(weights_path=tmp_path / "sg.pth"), LightGlueMatcher("superpoint", weights_path=tmp_path / "lg.pth")):
        gen = BatchedDetDescCorrespondenceGenerator(matcher, det, image_batch=2, pair_batch=3)
        kps, corr = gen.generate_correspondences(None, images, graph)
        assert len(kps) == 4 and sorted(corr) == sorted(graph)
        host = []
        for im in images:
            kp, d = det.detect_and_describe(im)
            order = np.lexsort((kp.coordinates[:, 0], kp.coordinates[:, 1]))
            host.append((kp.extract_indices(order), d[order]))
        for i in range(4):
            assert len(kps[i]) <= 120 and kps[i] == host[i][0]
        if images[2].mask is not None:
            rc = np.round(kps[2].coordinates).astype(int)
            assert (mask[rc[:, 1], rc[:, 0]] == 1).all()
        for i, j in graph:
            ref = matcher.match(host[i][0], host[j][0], host[i][1], host[j][1], images[i].shape + (1,) * (3 - len(images[i].shape)),
                                images[j].shape + (1,) * (3 - len(images[j].shape)))
            np.testing.assert_array_equal(corr[(i, j)], ref)
            assert corr[(i, j)].dtype == ref.dtype


@pytest.mark.parametrize("which", ["superglue", "lightglue"])
def test_batched_correspondence_generator_vs_oracle(gpu_device, sg_sd, tmp_path, which):
    """The resident generator (device top-k, ragged gathers, pair chunks alternating over two HIP streams) against the
    ORACLE directly -- detection + wrapper selection + matcher restated on the CPU -- on a small scene of overlapping views
    with ragged keypoint counts (the other generator tests compare the HIP pipeline with the HIP plugins)."""
    from gtsfm_amd.common.image import Image
    from gtsfm_amd.frontend.correspondence_generator.batched_det_desc_correspondence_generator import (
        BatchedDetDescCorrespondenceGenerator,
    )
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher
    from oracle import superpoint_oracle as spo

    sp_sd = synthetic.synthetic_superpoint_state_dict()
    lg_sd = synthetic.synthetic_lightglue_state_dict()
    torch.save(sp_sd, str(tmp_path / "sp.pth"))
    torch.save(sg_sd, str(tmp_path / "sg.pth"))
    torch.save(lg_sd, str(tmp_path / "lg.pth"))
    views = synthetic.synthetic_overlapping_views(4, 200, 264, seed=91)
    from oracle import imageprep_oracle as ipo

    rgb1 = np.stack([views[1], views[1][::-1, ::-1], 255 - views[1]], -1)  # an RGB image: gray conversion on the device
    grays = [views[0], ipo.rgb_to_gray_u8(rgb1), views[2][:168, :232], views[3]]  # one smaller image: fewer keypoints than the cap
    images = [Image(value_array=grays[0]), Image(value_array=rgb1), Image(value_array=grays[2]), Image(value_array=grays[3])]
    graph = [(0, 1), (0, 2), (1, 3), (2, 3), (0, 3)]
    cap = 300
    det = SuperPointDetectorDescriptor(max_keypoints=cap, weights_path=tmp_path / "sp.pth")
    matcher = SuperGlueMatcher(weights_path=tmp_path / "sg.pth") if which == "superglue" else LightGlueMatcher("superpoint", weights_path=tmp_path / "lg.pth")
    gen = BatchedDetDescCorrespondenceGenerator(matcher, det, image_batch=2, pair_batch=2)
    kps, corr = gen.generate_correspondences(None, images, graph)
    feats = []
    for g in grays:
        c, sc, d = spo.detect_and_describe(sp_sd, g, max_keypoints=1 << 30)
        sel = synthetic.topk_detection_order(sc, cap)
        feats.append((c[sel], sc[sel], d[sel]))
    assert len({len(f[0]) for f in feats}) > 1, "the scene should have ragged keypoint counts"
    for i in range(4):
        np.testing.assert_array_equal(kps[i].coordinates, feats[i][0])
        np.testing.assert_allclose(kps[i].responses, feats[i][1], rtol=0, atol=SCORE_TOL)
    total = 0
    for i, j in graph:
        (c0, s0, d0), (c1, s1, d1) = feats[i], feats[j]
        shp0, shp1 = grays[i].shape, grays[j].shape
        with torch.no_grad():
            if which == "superglue":
                ora = sgo.superglue_forward(sg_sd, T(c0)[None], T(c1)[None], T(s0)[None], T(s1)[None], T(d0).T[None].contiguous(),
                                            T(d1).T[None].contiguous(), shp0, shp1, sinkhorn_iterations=20)
            else:
                ora = lgo.lightglue_forward(lg_sd, T(c0)[None], T(c1)[None], T(d0)[None], T(d1)[None], shp0, shp1)
        m0 = ora["matches0"][0].numpy()
        valid = m0 > -1
        ref = np.stack([np.flatnonzero(valid), m0[valid]], -1)
        np.testing.assert_array_equal(corr[(i, j)], ref.astype(corr[(i, j)].dtype))
        assert corr[(i, j)].dtype == (np.uint32 if which == "superglue" else np.int64)
        total += len(ref)
    assert total > 100


def test_real_images_lund_door_pair(gpu_device, sg_engine):
    """Real-image pair end to end on the device (SuperPoint -> top-1024 -> SuperGlue, 20 iterations) vs the reference's
    outputs for the same two Lund-door frames."""
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    g = np.load(GOLDEN / "lund_door_pair.npz")
    sp = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device)
    out = sp.forward(torch.from_numpy(np.stack([g["gray0"], g["gray1"]])).to(gpu_device), top_k=1024)
    assert out["count"].tolist() == [1024, 1024]
    n = 1024
    kp = out["xy"].reshape(-1, 2).contiguous()
    sc = out["scores"].reshape(-1).contiguous()
    de = out["descriptors"].reshape(-1, 256).contiguous()
    res = sg_engine.match_batch(kp, sc, de, [n], [n], [[568, 380, 568, 380]], sinkhorn_iterations=20)
    m0 = res["matches"][:n].cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(m0, g["matches0"])
    np.testing.assert_allclose(res["mscores"][:n].cpu().numpy(), g["matching_scores0"], rtol=0, atol=SCORE_TOL)


@pytest.mark.parametrize("matcher", ["superglue", "lightglue"])
def test_max_keypoints_5000_vs_oracle(gpu_device, matcher):
    """GTSfM's default cap (max_keypoints = 5000, deep_front_end.yaml:29): 100 MB score matrices, ragged 5000 x 4800 pair,
    LightGlue above its pruning threshold (1536). Matches identical to the oracle, scores within tolerance. Shallow
    models (4 / 3 layers) keep the CPU oracle to a few seconds; depth is covered by the other tests."""
    n0, n1 = 5000, 4800
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(n0, n1, (1024, 1024), (1024, 1024), seed=77)
    if matcher == "superglue":
        from gtsfm_amd.runtime.matcher_engine import SuperGlueEngine

        sd = synthetic.synthetic_superglue_state_dict(num_layers=4)
        res = SuperGlueEngine(sd, gpu_device).match_pair(k0, s0, d0, k1, s1, d1, (1024, 1024), (1024, 1024), sinkhorn_iterations=20)
        with torch.no_grad():
            ora = sgo.superglue_forward(sd, T(k0)[None], T(k1)[None], T(s0)[None], T(s1)[None], T(d0).T[None].contiguous(),
                                        T(d1).T[None].contiguous(), (1024, 1024), (1024, 1024), sinkhorn_iterations=20)
    else:
        from gtsfm_amd.runtime.matcher_engine import LightGlueEngine

        sd = synthetic.synthetic_lightglue_state_dict(num_layers=3, match_bias=-2.0, match_gain=30.0)
        res = LightGlueEngine(sd, gpu_device).match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024), depth_confidence=-1.0)
        with torch.no_grad():
            ora = lgo.lightglue_forward(sd, T(k0)[None], T(k1)[None], T(d0)[None], T(d1)[None], (1024, 1024), (1024, 1024),
                                        depth_confidence=-1.0, return_intermediates=True)
        assert res["stop"] == ora["stop"]
        assert tuple(res["kept"].tolist()) == (ora["ind0"].shape[1], ora["ind1"].shape[1])
        assert res["kept"][0] < n0 and res["kept"][1] < n1  # point pruning was active on both images
    m0 = ora["matches0"][0].numpy()
    if matcher == "superglue":
        assert (m0 > -1).sum() > 1000
    np.testing.assert_array_equal(res["matches0"], m0)
    np.testing.assert_array_equal(res["matches1"], ora["matches1"][0].numpy())
    np.testing.assert_allclose(res["matching_scores0"], ora["matching_scores0"][0].numpy(), rtol=0, atol=SCORE_TOL)


@pytest.mark.parametrize("matcher", ["superglue", "lightglue"])
@pytest.mark.parametrize("cap,graphs", [(120, False), (2000, False), (256, True)])
def test_first_layer_shared_per_image_is_bit_identical(gpu_device, matcher, cap, graphs):
    """The block of the first matcher layer that sees one image (SuperGlue: keypoint encoder + first self layer; LightGlue:
    first self block) computed once per image (``prepare_images`` + ``first_layer_done``) against the plain per-pair forward:
    identical matches AND scores, bit for bit, for full tables (cap 120 / 256), ragged ones (cap 2000 > detections), graph
    replay, an odd image count, and an image that appears in a single pair."""
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device)
    if matcher == "superglue":
        eng, mk = ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(num_layers=4), gpu_device), {"sinkhorn_iterations": 20}
    else:
        eng, mk = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=3), gpu_device), {}
    views = synthetic.synthetic_overlapping_views(5, 192, 256, seed=31)
    pairs = [(0, 1), (0, 2), (1, 2), (2, 3), (0, 3), (3, 4)]
    shared = FrontEndPipeline(det, eng, max_keypoints=cap, pair_chunk=2, num_streams=2, use_graphs=graphs, share_first_layer=True)
    plain = FrontEndPipeline(det, eng, max_keypoints=cap, pair_chunk=2, num_streams=2, use_graphs=graphs, share_first_layer=False)
    feats = shared.detect(torch.from_numpy(views).to(gpu_device))
    a = shared.match(feats, pairs, [(192, 256)] * 5, **mk)
    b = plain.match(feats, pairs, [(192, 256)] * 5, **mk)
    torch.cuda.synchronize()
    assert shared.last_shared_images == 5 and plain.last_shared_images == 0
    total = 0
    for x, y in zip(a, b):
        assert x["pairs"] == y["pairs"]
        assert torch.equal(x["matches"], y["matches"]) and torch.equal(x["mscores"], y["mscores"])
        total += int((x["matches"] > -1).sum())
    assert total > 0
    # independent pairs (no image reused): nothing to share, the plain path runs
    shared.match(feats, [(0, 1), (2, 3)], [(192, 256)] * 5, **mk)
    assert shared.last_shared_images == 0


def test_graph_replay_survives_descriptor_cache_eviction_and_empty_pair_lists(gpu_device):
    """A captured chunk copies its descriptor block from the engine's cache by ADDRESS: the block must outlive any number of
    other batch shapes passing through the cache (ragged scenes add one per chunk). And a rank that owns no pair gets []."""
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device)
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=2), gpu_device)
    eng.desc_cache_capacity = 4
    pipe = FrontEndPipeline(det, eng, max_keypoints=128, pair_chunk=2, num_streams=1, use_graphs=True, share_first_layer=False)
    feats = pipe.detect(torch.from_numpy(synthetic.synthetic_overlapping_views(3, 192, 256, seed=3)).to(gpu_device))
    pairs, shapes = [(0, 1), (0, 2)], [(192, 256)] * 3
    assert pipe.match(feats, [], shapes) == []
    first = pipe.match(feats, pairs, shapes)
    torch.cuda.synchronize()
    assert len(pipe._graphs) == 1 and len(eng._desc_pinned) >= 1
    k0, _, d0, k1, _, d1, _ = synthetic.synthetic_pair_features(40, 40, (192, 256), (192, 256), seed=1)
    junk = []
    for n in range(20, 32):  # twelve other shapes through a cache of four
        eng.match_pair(k0[:n], d0[:n], k1[:n], d1[:n], (192, 256), (192, 256))
        junk.append(torch.full((4096,), 7, dtype=torch.int32, device=gpu_device))  # whatever the allocator hands out next gets overwritten
    assert all(key in eng._desc_cache for key in eng._desc_pinned) and len(eng._desc_cache) <= 4 + len(eng._desc_pinned)
    again = pipe.match(feats, pairs, shapes)
    torch.cuda.synchronize()
    assert torch.equal(first[0]["matches"], again[0]["matches"]) and torch.equal(first[0]["mscores"], again[0]["mscores"])
    assert int((first[0]["matches"] > -1).sum()) > 0


@pytest.mark.parametrize("math", [1, 2], ids=["bf16x3", "f16x2"])
def test_attention_bf16x3_arithmetic(lib, gpu_device, math):
    """The opt-in arithmetics (``gtsfm_attention_math_f32`` with math = 1, GTSFM_ATTENTION_MATH=bf16x3: K Q^T and P V on bf16 MFMA with
    every fp32 operand split exactly into three bf16 pieces; math = 2, =f16x2: two fp16 pieces, three products), fp32 accumulation. Against a FLOAT64 reference its error must be of the
    class of the exact-fp32 kernel's (both are measured; bf16x3 may carry at most twice the exact kernel's error + 2e-6), its fused and
    split schedules must agree bit for bit with each other (same segment merge), rows beyond a count stay untouched, and a problem
    without keys writes zeros. Same ragged problem set as the schedule test: 1, 2, 3 and 5 key segments, cross problems, late dominant keys."""
    counts = [300, 1024, 1025, 2048, 2500, 5000, 0, 70]
    caps = [384, 1024, 1152, 2048, 2560, 5120, 128, 128]
    offs = np.concatenate([[0], np.cumsum(caps)])[:-1]
    total = int(sum(caps))
    gen = torch.Generator().manual_seed(3)
    qkv = torch.randn((total, 768), generator=gen)
    qkv[:, :512] *= 1.5
    qkv[offs[5] + 4990, 256:320] = qkv[offs[5] : offs[5] + 5000, :64].mean(0) * 40
    qkv[offs[3] + 3, 256 + 64 : 256 + 128] = qkv[offs[3] : offs[3] + 2048, 64:128].mean(0) * 40
    problems = [(0, 0), (1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (7, 6), (7, 5), (4, 2), (2, 4)]
    d = qkv.to(gpu_device)
    cnt = torch.tensor(counts, dtype=torch.int32, device=gpu_device)
    ws = torch.empty(int(lib.gtsfm_attention_math_workspace_bytes(len(problems), max(counts), max(counts), 4, total, math)), dtype=torch.uint8, device=gpu_device)

    def run(sel, mode, math):
        prob = torch.tensor([[offs[a], a, offs[b], b] for a, b in sel], dtype=torch.int32, device=gpu_device)
        out = torch.full((total, 256), float("nan"), device=gpu_device)
        rc = lib.gtsfm_attention_math_f32(d.data_ptr(), 768, d.data_ptr() + 256 * 4, 768, d.data_ptr() + 512 * 4, 768, out.data_ptr(), 256, prob.data_ptr(),
                                          cnt.data_ptr(), len(sel), max(counts[a] for a, _ in sel), max(counts), 4, 0.125, mode, math, total, ws.data_ptr(), ws.numel(),
                                          _stream())
        assert rc == 0, lib.gtsfm_last_error()
        torch.cuda.synchronize()
        return out.cpu()

    q64 = qkv.double()
    worst = {"exact": 0.0, "x3": 0.0}
    for sel in ([problems[0], problems[3], problems[5]], [problems[1], problems[2], problems[4], problems[6]], [problems[7]], [problems[8]], [problems[9]]):
        exact, fused, split = run(sel, -1, 0), run(sel, -1, math), run(sel, 1, math)
        touched = torch.zeros(total, dtype=torch.bool)
        for a, b in sel:
            rows = slice(offs[a], offs[a] + counts[a])
            touched[rows] = True
            assert torch.equal(fused[rows], split[rows]), (a, b)
            if counts[b] == 0:
                assert float(fused[rows].abs().max()) == 0.0
                continue
            ref = _ref_attention(q64[rows, :256], q64[offs[b] : offs[b] + counts[b], 256:512], q64[offs[b] : offs[b] + counts[b], 512:], 0.125)
            e_exact, e_x3 = float((exact[rows].double() - ref).abs().max()), float((fused[rows].double() - ref).abs().max())
            worst["exact"], worst["x3"] = max(worst["exact"], e_exact), max(worst["x3"], e_x3)
            assert e_x3 <= 2.0 * e_exact + 2e-6, ((a, b), e_x3, e_exact)
            assert not torch.equal(exact[rows], fused[rows])  # (the switch did something)
        assert torch.isnan(fused[~touched]).all() and torch.isnan(split[~touched]).all()
    print(f"attention max |error| against float64: exact fp32 {worst['exact']:.3e}, {'bf16x3' if math == 1 else 'f16x2'} {worst['x3']:.3e}")


@pytest.mark.parametrize("v_scale,q_scale", [(1.0e4, 1.0), (1.0e-5, 1.0), (1.0, 6.0), (3.0e4, 1.0)], ids=["values_near_fp16_max", "values_below_fp16_normal", "peaked_softmax", "values_beyond_fp16_max"])
def test_attention_f16x2_at_the_edges_of_its_range(lib, gpu_device, v_scale, q_scale):
    """The documented domain of the f16x2 arithmetic (gtsfm_amd/csrc/f16x2.h) at kernel level, against float64: (1) values up to 4.5e4 (fp16's largest
    number is 65504): relative accuracy of the exact kernel's class; (2) values of 1e-5, whose fp16 pieces are subnormal: the ABSOLUTE floor -- 2^-25 per
    operand, so |error| <= 1e-7 -- instead of a relative bound (documented: the mode carries fp32-class RELATIVE accuracy for |x| >= 2^-3 only);
    (3) scores of +-60 (a one-hot softmax: the weights span the whole range the kernel's shifted reference has to cover); (4) values beyond 65504: the leading
    piece is inf and the affected outputs are NaN -- never a clamped number."""
    n, cap = 700, 768
    gen = torch.Generator().manual_seed(11)
    qkv = torch.randn((2 * cap, 768), generator=gen)
    qkv[:, :256] *= q_scale
    qkv[:, 512:] *= v_scale
    d = qkv.to(gpu_device)
    cnt = torch.tensor([n, n], dtype=torch.int32, device=gpu_device)
    prob = torch.tensor([[0, 0, cap, 1], [cap, 1, 0, 0]], dtype=torch.int32, device=gpu_device)
    ws = torch.empty(int(lib.gtsfm_attention_math_workspace_bytes(2, n, n, 4, 2 * cap, 2)), dtype=torch.uint8, device=gpu_device)

    def run(math):
        out = torch.zeros((2 * cap, 256), device=gpu_device)
        rc = lib.gtsfm_attention_math_f32(d.data_ptr(), 768, d.data_ptr() + 256 * 4, 768, d.data_ptr() + 512 * 4, 768, out.data_ptr(), 256, prob.data_ptr(), cnt.data_ptr(), 2, n, n, 4,
                                          0.125, 0, math, 2 * cap, ws.data_ptr(), ws.numel(), _stream())
        assert rc == 0, lib.gtsfm_last_error()
        torch.cuda.synchronize()
        return out.cpu()

    exact, split = run(0), run(2)
    q64 = qkv.double()
    worst_exact = worst_split = scale = 0.0
    for a, b in ((0, 1), (1, 0)):
        rows = slice(a * cap, a * cap + n)
        ref = _ref_attention(q64[rows, :256], q64[b * cap : b * cap + n, 256:512], q64[b * cap : b * cap + n, 512:], 0.125)
        if v_scale > 2.0e4:  # some |v| exceed 65504: their channels' outputs are NaN, everything else is finite -- and nothing is silently clamped
            bad = torch.isnan(split[rows])
            assert bad.any() and not torch.isinf(split[rows]).any()
            over = (q64[b * cap : b * cap + n, 512:].abs() > 65504).any(0)  # channels holding a value beyond fp16's range
            assert bool(over.any()) and not bad[:, ~over].any() and torch.isfinite(exact[rows]).all()
            continue
        worst_exact = max(worst_exact, float((exact[rows].double() - ref).abs().max()))
        worst_split = max(worst_split, float((split[rows].double() - ref).abs().max()))
        scale = max(scale, float(ref.abs().max()))
    if v_scale > 2.0e4:
        return
    print(f"f16x2 at the edge (v x {v_scale:g}, q x {q_scale:g}): max |error| against float64 {worst_split:.3e} (exact fp32 kernel {worst_exact:.3e}), outputs up to {scale:.3e}")
    if v_scale < 1e-3:
        assert worst_split <= 1e-7, worst_split                      # the absolute floor: 2^-25 per operand piece
    else:
        assert worst_split <= 2.0 * worst_exact + 2e-6 * scale, (worst_split, worst_exact)


@pytest.mark.parametrize("matcher", ["superglue", "lightglue"])
def test_single_pair_schedules_are_bit_identical_to_the_batch_schedules(gpu_device, monkeypatch, matcher):
    """One pair on its own takes the schedules that fill the chip -- 64 x 64 GEMM tiles, attention split over key segments -- while a
    batch takes 128 x 128 tiles and the fused attention. Forcing either set on the SAME single pair (rotary epilogue, residual
    epilogue, per-tile live counts after point pruning, 3 key segments) must give identical matches and scores, bit for bit."""
    from gtsfm_amd.runtime import matcher_engine as ME

    n0, n1 = 2300, 2100
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(n0, n1, (1024, 1024), (1024, 1024), seed=91)
    if matcher == "superglue":
        eng = ME.SuperGlueEngine(synthetic.synthetic_superglue_state_dict(num_layers=4), gpu_device)
        run = lambda: eng.match_pair(k0, s0, d0, k1, s1, d1, (1024, 1024), (1024, 1024), sinkhorn_iterations=10)  # noqa: E731
    else:
        sd = synthetic.synthetic_lightglue_state_dict(num_layers=3, match_bias=-2.0, match_gain=30.0)  # the recipe of the 5000-keypoint test: pruning fires
        eng = ME.LightGlueEngine(sd, gpu_device)
        run = lambda: eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024), depth_confidence=-1.0)  # noqa: E731
    monkeypatch.setenv("GTSFM_GEMM_SMALL_BELOW", "0")
    monkeypatch.setenv("GTSFM_ATTENTION_SPLIT", "0")
    batch_like = run()
    monkeypatch.setenv("GTSFM_GEMM_SMALL_BELOW", str(1 << 40))
    monkeypatch.setenv("GTSFM_ATTENTION_SPLIT", "1")
    single_like = run()
    monkeypatch.delenv("GTSFM_GEMM_SMALL_BELOW")
    monkeypatch.delenv("GTSFM_ATTENTION_SPLIT")
    default = run()
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
        np.testing.assert_array_equal(batch_like[key], single_like[key])
        np.testing.assert_array_equal(batch_like[key], default[key])
    assert (default["matches0"] > -1).sum() > 50
    if matcher == "lightglue":
        assert default["kept"][0] < n0  # point pruning was active: the GEMMs ran on per-tile live counts


@pytest.mark.parametrize("shapes", [[(300, 257)], [(129, 130), (2048, 1900), (17, 640)], [(5000, 4800), (2100, 5000)]])
def test_lightglue_assignment_standalone_vs_oracle(lib, gpu_device, shapes):
    """``gtsfm_lg_assignment_f32`` -- the forward's last stage alone (the two log-softmax sweeps, then one read of the matrix for the mutual
    arg-maxima and the 0.1 filter) -- on seeded similarity matrices and matchability logits against the oracle's
    ``sigmoid_log_double_softmax`` + ``filter_matches`` (restating upstream LightGlue): match indices identical on
    both sides, scores within 1e-5; ragged batches mix the one-wave, four-wave and eight-wave sweep tiers; ``stages`` 1 then 2 equals 3."""
    from gtsfm_amd.runtime import lib as L

    rng = np.random.default_rng(len(shapes) * 100 + shapes[0][0])
    m = np.array([s[0] for s in shapes], dtype=np.int32)
    n = np.array([s[1] for s in shapes], dtype=np.int32)
    cap = lambda v: -(-int(v) // 128) * 128  # noqa: E731
    rows = sum(cap(a) + cap(b) for a, b in shapes)
    sims, zs, sim_flat = [], [], []
    zlogit = np.zeros(rows, dtype=np.float32)
    row = 0
    for a, b in shapes:
        sim = rng.normal(0.0, 4.0, (a, b)).astype(np.float32)
        hit = rng.permutation(min(a, b))[: min(a, b) // 2]
        sim[hit, hit] += 25.0  # mutual maxima that pass the filter
        z0, z1 = rng.normal(1.0, 2.0, a).astype(np.float32), rng.normal(1.0, 2.0, b).astype(np.float32)
        ld = (b + 3) // 4 * 4
        padded = np.zeros((a, ld), dtype=np.float32)
        padded[:, :b] = sim
        sim_flat.append(padded.reshape(-1))
        zlogit[row : row + a] = z0
        zlogit[row + cap(a) : row + cap(a) + b] = z1
        sims.append(sim), zs.append((z0, z1, row))
        row += cap(a) + cap(b)
    sim_dev = torch.from_numpy(np.concatenate(sim_flat)).to(gpu_device)
    zl_dev = torch.from_numpy(zlogit).to(gpu_device)
    ws = torch.empty(int(lib.gtsfm_lg_assignment_workspace_bytes(len(shapes), m.ctypes.data, n.ctypes.data)), dtype=torch.uint8, device=gpu_device)
    stream = torch.cuda.current_stream(gpu_device).cuda_stream

    def run(stages):
        matches = torch.full((rows,), -7, dtype=torch.int32, device=gpu_device)
        ms = torch.full((rows,), -7.0, dtype=torch.float32, device=gpu_device)
        for st in stages:
            L.check(lib.gtsfm_lg_assignment_f32(sim_dev.data_ptr(), len(shapes), m.ctypes.data, n.ctypes.data, zl_dev.data_ptr(), 0.1, st, ws.data_ptr(), ws.numel(),
                                                matches.data_ptr(), ms.data_ptr(), stream), "gtsfm_lg_assignment_f32")
        return matches.cpu().numpy(), ms.cpu().numpy()

    got_m, got_s = run([3])
    two_m, two_s = run([1, 2])
    np.testing.assert_array_equal(got_m, two_m)
    np.testing.assert_array_equal(got_s, two_s)
    total = 0
    for (a, b), sim, (z0, z1, row) in zip(shapes, sims, zs):
        t = torch.from_numpy
        with torch.no_grad():
            scores = lgo.sigmoid_log_double_softmax(t(sim)[None], t(z0)[None, :, None], t(z1)[None, :, None])
            m0, m1, s0, s1 = lgo.filter_matches(scores, 0.1)
        np.testing.assert_array_equal(got_m[row : row + a], m0[0].numpy())
        np.testing.assert_array_equal(got_m[row + cap(a) : row + cap(a) + b], m1[0].numpy())
        np.testing.assert_allclose(got_s[row : row + a], s0[0].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(got_s[row + cap(a) : row + cap(a) + b], s1[0].numpy(), rtol=0, atol=1e-5)
        total += int((m0[0] > -1).sum())
    assert total > 10 * len(shapes)


def test_layernorm_gelu_standalone_vs_aten(lib, gpu_device):
    """``gtsfm_layernorm_gelu_f32`` (LightGlue's FFN normalisation + activation, in place) against ATen's layer_norm + exact gelu on the CPU."""
    from gtsfm_amd.runtime import lib as L

    torch.manual_seed(5)
    rows, ld = 1234, 520
    x = torch.randn((rows, ld)) * 3.0 + 0.5
    gamma, beta = torch.rand(512) + 0.5, torch.randn(512) * 0.2
    want = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x[:, :512], (512,), gamma, beta))
    xd, gd, bd = x.to(gpu_device), gamma.to(gpu_device), beta.to(gpu_device)
    scratch = torch.empty(64, dtype=torch.uint8, device=gpu_device)
    L.check(lib.gtsfm_layernorm_gelu_f32(xd.data_ptr(), ld, rows, gd.data_ptr(), bd.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream(gpu_device).cuda_stream),
            "gtsfm_layernorm_gelu_f32")
    got = xd.cpu()
    assert float((got[:, :512] - want).abs().max()) < 2e-5
    assert torch.equal(got[:, 512:], x[:, 512:])  # columns beyond 512 are not touched


@pytest.mark.parametrize("kw,n0,n1", [({}, 5000, 4800), ({}, 2048, 2000), ({}, 100, 333), ({"conf_bias": 3.0, "conf_gain": 6.0}, 1800, 2100),
                                      ({"conf_bias": 1.0, "conf_gain": 6.0, "match_bias": 2.0, "match_gain": 12.0}, 2560, 2304)])
def test_single_pair_two_stream_form_is_bit_identical(gpu_device, kw, n0, n1):
    """``gtsfm_lg_forward_streams``: ONE pair's launch sequence split over two streams (image 0's per-image work on the caller's stream, image 1's
    on the lane's side stream, event waits at the cross attention and once per layer) against the one-stream form: matches, scores, stop
    layer and kept counts identical bit for bit -- at the cap, for ragged / tiny sets, with heads that stop early and with pruning active --
    through the image cache (phase 2) and without it (phase 0), and repeatedly (the two sequences race differently every time)."""
    from gtsfm_amd.runtime import matcher_engine as ME

    sd = synthetic.synthetic_lightglue_state_dict(**kw)
    k0, _, d0, k1, _, d1, _ = synthetic.synthetic_pair_features(n0, n1, (1024, 1024), (768, 1024), seed=510 + n0 % 7)
    one, two = ME.LightGlueEngine(sd, gpu_device), ME.LightGlueEngine(sd, gpu_device)
    one.pair_streams, two.pair_streams = 1, 2
    for cache in (0, 64):
        one.image_cache_capacity = two.image_cache_capacity = cache
        want = one.match_pair(k0, d0, k1, d1, (1024, 1024), (768, 1024), pruning_threshold=-1 if "match_bias" in kw else ME.LIGHTGLUE_PRUNING_THRESHOLD)
        for rep in range(4):
            got = two.match_pair(k0, d0, k1, d1, (1024, 1024), (768, 1024), pruning_threshold=-1 if "match_bias" in kw else ME.LIGHTGLUE_PRUNING_THRESHOLD)
            for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "kept"):
                np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key}, cache {cache}, repetition {rep}")
            assert got["stop"] == want["stop"]
    assert two._side_stream is not None and one._side_stream is None
    assert (want["matches0"] > -1).sum() > min(n0, n1) // 20
    if "conf_bias" in kw and "match_bias" not in kw:
        assert want["stop"] < 9
    if "match_bias" in kw:
        assert int(want["kept"].min()) < min(n0, n1)
