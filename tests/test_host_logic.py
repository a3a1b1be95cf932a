"""CPU: host-side logic of the matcher path -- batch descriptor builder, weight preparation (BatchNorm folding, head
permutation), blob packing, and the multi-process plumbing (gloo, world_size 2)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from gtsfm_amd.utils import synthetic


def test_match_descriptor_block(built_library):
    from gtsfm_amd.runtime import lib as L

    lib = L.load()
    n0 = np.array([5, 130], dtype=np.int32)
    n1 = np.array([7, 1], dtype=np.int32)
    hw = np.array([[10, 20, 30, 40], [50, 60, 70, 80]], dtype=np.int32)
    for superglue in (1, 0):
        size = lib.gtsfm_match_desc_ints(superglue, 2, n0.ctypes.data, n1.ctypes.data)
        d = np.full(size, -7, dtype=np.int32)
        assert lib.gtsfm_match_build_desc(superglue, 2, n0.ctypes.data, n1.ctypes.data, hw.ctypes.data, d.ctypes.data) == 0
        live, orig = d[0:4], d[12:16]
        assert live.tolist() == orig.tolist() == [5, 7, 130, 1]
        assert d[20:22].tolist() == [-1, -1]  # stop layers
        seqs = d[22 : 22 + 24].reshape(4, 6)
        caps = [5, 7, 130, 1] if superglue else [128, 128, 256, 128]
        assert seqs[:, 5].tolist() == caps
        assert seqs[:, 0].tolist() == np.concatenate([[0], np.cumsum(caps)[:-1]]).tolist()  # internal row offsets
        assert seqs[:, 4].tolist() == [0, 5, 12, 142]  # caller-side (unpadded) offsets
        assert seqs[:, 2:4].tolist() == [[10, 20], [30, 40], [50, 60], [70, 80]]
        pairs = d[46 : 46 + 12].view(np.int64).reshape(2, 3)
        ext = 1 if superglue else 0
        ld0 = (7 + ext + 3) // 4 * 4
        assert pairs[0, 0] == 0 and pairs[1, 0] == (5 + ext) * ld0
        if not superglue:
            tiles = d[-2 * 5 :]
            assert tiles[:5].tolist() == [0, 1, 2, 2, 3] and tiles[5:].tolist() == [0, 0, 0, 128, 0]
    # empty keypoint sets are the caller's business (superglue.py:233-240)
    n0[0] = 0
    assert lib.gtsfm_match_build_desc(1, 2, n0.ctypes.data, n1.ctypes.data, hw.ctypes.data, d.ctypes.data) != 0
    assert b"empty" in lib.gtsfm_last_error()


def test_superglue_weight_preparation_is_equivalent(built_library):
    """BatchNorm folding + head permutation + q/k/v fusion + the merge projection folded into mlp.0 reproduce one
    propagation layer of the reference in float64 (superglue.py:92-119)."""
    from gtsfm_amd.runtime.matcher_engine import HEAD_PERM, superglue_entries
    from oracle import superglue_oracle as sgo

    sd = synthetic.synthetic_superglue_state_dict(num_layers=1)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    entries = superglue_entries(sd)
    assert len(entries) == 5 + 3 + 1
    x = torch.randn((1, 256, 40), dtype=torch.float64)
    src = torch.randn((1, 256, 33), dtype=torch.float64)
    ref = sgo._propagation(sd64, "gnn.layers.0", x, src)[0].T.numpy()
    (_, wqkv, bqkv), (_, w0, b0), (_, w1, b1) = entries[5:8]
    xt, st = x[0].T.numpy(), src[0].T.numpy()
    q = xt @ wqkv[:256].T + bqkv[:256]
    k = st @ wqkv[256:512].T + bqkv[256:512]
    v = st @ wqkv[512:].T + bqkv[512:]
    att = np.zeros((40, 256))
    for h in range(4):
        s = q[:, 64 * h : 64 * h + 64] @ k[:, 64 * h : 64 * h + 64].T / 8.0
        p = np.exp(s - s.max(1, keepdims=True))
        att[:, 64 * h : 64 * h + 64] = (p / p.sum(1, keepdims=True)) @ v[:, 64 * h : 64 * h + 64]
    hid = np.maximum(np.concatenate([xt, att], 1) @ w0.T + b0, 0)  # cat([x, attention output]): merge lives inside w0
    out = hid @ w1.T + b1
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)
    assert sorted(HEAD_PERM.tolist()) == list(range(256))


def test_lightglue_weight_preparation_regroups_qkv(built_library):
    from gtsfm_amd.runtime.matcher_engine import lightglue_entries

    sd = synthetic.synthetic_lightglue_state_dict(num_layers=2)
    entries, match_bias, conf_bias = lightglue_entries(sd)
    assert len(entries) == 1 + 2 * 12 + 1 and match_bias.shape == (2,) and conf_bias.shape == (2,)
    w = sd["transformers.0.self_attn.Wqkv.weight"].double().numpy()
    x = np.random.default_rng(0).standard_normal((5, 256))
    qkv = (x @ w.T).reshape(5, 4, 64, 3)  # upstream unflatten(-1, (heads, head_dim, 3))
    new = x @ entries[1][1].T
    np.testing.assert_allclose(new[:, :256].reshape(5, 4, 64), qkv[..., 0], atol=1e-12)
    np.testing.assert_allclose(new[:, 256:512].reshape(5, 4, 64), qkv[..., 1], atol=1e-12)
    np.testing.assert_allclose(new[:, 512:].reshape(5, 4, 64), qkv[..., 2], atol=1e-12)


def test_lightglue_loader_accepts_published_key_layout(built_library):
    """The published superpoint_lightglue.pth stores self_attn.{i}.* / cross_attn.{i}.*; upstream renames them to
    transformers.{i}.* at load time. Both layouts must pack to the same blob."""
    from gtsfm_amd.runtime.matcher_engine import lightglue_entries, lightglue_num_layers, normalize_lightglue_state_dict, pack_blob

    sd = synthetic.synthetic_lightglue_state_dict(num_layers=3)
    old_style = {}
    for k, v in sd.items():
        parts = k.split(".")
        if parts[0] == "transformers":
            k = ".".join([parts[2], parts[1]] + parts[3:])  # transformers.i.self_attn.x -> self_attn.i.x
        old_style[k] = v
    assert not any(k.startswith("transformers.") for k in old_style) and "self_attn.2.Wqkv.weight" in old_style
    assert lightglue_num_layers(old_style) == 3
    assert sorted(normalize_lightglue_state_dict(old_style)) == sorted(sd)
    e_new, mb_new, cb_new = lightglue_entries(sd)
    e_old, mb_old, cb_old = lightglue_entries(old_style)
    assert np.array_equal(pack_blob(e_new), pack_blob(e_old)) and np.array_equal(mb_new, mb_old) and np.array_equal(cb_new, cb_old)
    with pytest.raises(KeyError, match="neither"):
        lightglue_num_layers({"posenc.Wr.weight": sd["posenc.Wr.weight"]})


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, tmp: str):
    import torch.distributed as dist

    from gtsfm_amd import parallel

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    try:
        # weight broadcast
        blob = torch.arange(1000, dtype=torch.float32) * 0.5 if rank == 0 else None
        got = parallel.broadcast_packed_weights(blob, 1000, dev)
        assert torch.equal(got, torch.arange(1000, dtype=torch.float32) * 0.5)
        # ownership
        n_img = 5
        mine = parallel.partition_images(n_img, rank, world)
        pairs = parallel.exhaustive_pairs(n_img)
        my_pairs = parallel.partition_pairs(pairs, rank, world)
        # feature exchange between the detect and match phases
        local = {}
        for i in mine:
            k = 3 + i
            local[i] = (torch.full((k, 2), float(i)), torch.full((k,), i + 0.5), torch.full((k, 256), i + 0.25))
        feats = parallel.gather_features(local, n_img, 16, dev)
        assert sorted(feats) == list(range(n_img))
        for i, (xy, sc, de) in feats.items():
            assert xy.shape == (3 + i, 2) and float(xy[0, 0]) == i and float(sc[0]) == i + 0.5 and float(de[-1, -1]) == i + 0.25
        # variable-length results back to every rank (incl. an empty match list)
        res = {p: np.stack([np.arange(p[0] + p[1]), np.arange(p[0] + p[1]) + 1], 1).astype(np.int64) for p in my_pairs}
        if my_pairs:
            res[my_pairs[0]] = np.zeros((0, 2), dtype=np.int64)
        allres = parallel.gather_matches(res, dev)
        assert sorted(allres) == sorted(pairs)
        firsts = {parallel.partition_pairs(pairs, r, world)[0] for r in range(world)}
        for p, m in allres.items():
            if p in firsts:
                assert m.shape == (0, 2)
            else:
                assert m.shape == (p[0] + p[1], 2) and m[-1, 1] == p[0] + p[1]
        with open(os.path.join(tmp, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_multiprocess_plumbing_gloo_world2(tmp_path):
    """N > 1 path on CPU: gloo, world_size 2 -- weight broadcast, image / pair partitioning, feature all-gather, ragged
    result gather (the same calls run over RCCL on the GPUs)."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_partitions_cover_everything_once():
    from gtsfm_amd import parallel

    pairs = parallel.exhaustive_pairs(9)
    assert len(pairs) == 36
    for world in (1, 2, 3, 8):
        got = sum((parallel.partition_pairs(pairs, r, world) for r in range(world)), [])
        assert sorted(got) == pairs
        imgs = sum((parallel.partition_images(9, r, world) for r in range(world)), [])
        assert sorted(imgs) == list(range(9))


def test_bench_flop_accounting_matches_survey():
    """bench.py prices its TFLOP/s and rooflines with SURVEY.md section 8(a)/(d)'s algorithmic FLOP counts; pin them
    to the figures quoted there (they are what `roofline.achieved` is computed from)."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("bench_for_test", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # SuperPoint: 52.10 G @480x640, 177.85 G @1024x1024, 146.05 G @1135x760 (H x W = 760 x 1135 after the loader resize)
    assert abs(bench.superpoint_flops(480, 640) / 1e9 - 52.10) < 0.01
    assert abs(bench.superpoint_flops(1024, 1024) / 1e9 - 177.85) < 0.01
    assert abs(bench.superpoint_flops(760, 1135) / 1e9 - 146.05) < 0.01
    # SuperGlue dense FLOP per pair: 34.3 G @512, 88.2 G @1024, 254.8 G @2048, 1173.8 G @5000
    for n, g in ((512, 34.3), (1024, 88.2), (2048, 254.8), (5000, 1173.8)):
        assert abs(bench.matcher_flops("superglue", n, 18.0, 100) / 1e9 - g) < 0.06, n
    # LightGlue, full depth (9 layers): 31.7 G @512, 80.5 G @1024, 229.8 G @2048, 1044.6 G @5000
    for n, g in ((512, 31.7), (1024, 80.5), (2048, 229.8), (5000, 1044.6)):
        assert abs(bench.matcher_flops("lightglue", n, 9.0, 0) / 1e9 - g) < 0.06, n
    # early exit scales the per-layer term only
    assert bench.matcher_flops("lightglue", 2048, 4.5, 0) < 0.51 * bench.matcher_flops("lightglue", 2048, 9.0, 0) + 1.2e9


def test_matcher_engine_lanes_host_logic(monkeypatch):
    """Concurrent match() calls of one worker process each get an engine lane nobody else is using; at most max_lanes exist; a
    single-threaded caller only ever uses the first (matcher_engine._MatcherBase._lane; streams mocked: no GPU here)."""
    import contextlib
    import threading
    import time
    from collections import OrderedDict

    from gtsfm_amd.runtime import matcher_engine as ME

    monkeypatch.setattr(ME.torch.cuda, "Stream", lambda device=None: object())
    monkeypatch.setattr(ME.torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setenv("GTSFM_PLUGIN_LANES", "2")
    eng = object.__new__(ME._MatcherBase)
    eng.device, eng._lib, eng.weights, eng.num_layers = None, None, object(), 9
    eng._init_host_state()
    eng.some_per_call_attribute_added_later = []  # not in _SHARED_ATTRS: a lane must not inherit it (round 3 used copy.copy)
    assert eng.max_lanes == 2
    for _ in range(3):  # one caller: always the engine itself
        with eng._lane() as lane:
            assert lane is eng
    assert len(eng._lanes) == 1

    busy, seen, clashes, guard = set(), set(), [], threading.Lock()

    def worker():
        for _ in range(5):
            with eng._lane() as lane:
                with guard:
                    if id(lane) in busy:
                        clashes.append(id(lane))
                    busy.add(id(lane))
                    seen.add(id(lane))
                time.sleep(0.002)
                with guard:
                    busy.discard(id(lane))

    threads = [threading.Thread(target=worker) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not clashes
    assert len(eng._lanes) == 2 and len(seen) == 2
    sibling = next(lane for lane in eng._lanes if lane is not eng)
    assert sibling.weights is eng.weights and sibling._desc_cache is not eng._desc_cache and sibling._workspace is None
    assert not hasattr(sibling, "some_per_call_attribute_added_later") and sibling._root is eng
    assert eng._free_lanes.qsize() == 2
    # release_lanes(): idle siblings go, the engine keeps serving
    eng._workspace, eng._image_cache["k"] = object(), object()
    eng.release_lanes()
    assert eng._lanes == [eng] and eng._workspace is None and not eng._image_cache and eng._free_lanes.qsize() == 1
    with eng._lane() as lane:
        assert lane is eng


def test_image_key_tells_arrays_apart():
    """The per-call path's image cache key (matcher_engine._MatcherBase._image_key): same arrays -> same key; another buffer, another
    image shape or changed sampled rows -> another key."""
    from gtsfm_amd.runtime.matcher_engine import _MatcherBase

    rng = np.random.default_rng(0)
    k, d = rng.random((500, 2), dtype=np.float32), rng.random((500, 256), dtype=np.float32)
    key = _MatcherBase._image_key((k, d), (480, 640))
    assert key == _MatcherBase._image_key((k, d), (480, 640))
    assert key != _MatcherBase._image_key((k, d), (640, 480))
    assert key != _MatcherBase._image_key((k, d.copy()), (480, 640))
    d[-1, 3] += 1.0  # in-place change of a sampled row
    assert key != _MatcherBase._image_key((k, d), (480, 640))
    assert _MatcherBase._image_key((k[:0], d[:0]), (8, 8))  # empty sets do not break it


def test_full_digest_sees_every_byte():
    """The image cache's validation checksum (matcher_engine._full_digest) covers all bytes of all arrays: a change in a row the lookup key
    does not sample, a permutation of rows, another dtype view -- all give another digest; equal content in another buffer gives the same."""
    from gtsfm_amd.runtime.matcher_engine import _MatcherBase, _full_digest

    rng = np.random.default_rng(1)
    k, d = rng.random((500, 2), dtype=np.float32), rng.random((500, 256), dtype=np.float32)
    base = _full_digest((k, d))
    assert base == _full_digest((k.copy(), d.copy())) and len(base) >= 8
    key = _MatcherBase._image_key((k, d), (480, 640))
    d[250, 17] += 1e-3  # row 250 is not among the 36 sampled rows of 500
    assert _MatcherBase._image_key((k, d), (480, 640)) == key and _full_digest((k, d)) != base
    d[250, 17] -= 1e-3
    d[[100, 101]] = d[[101, 100]]  # a permutation keeps every sum
    assert _full_digest((k, d)) != base
    assert _full_digest((k[:0], d[:0])) == _full_digest((k[:0], d[:0]))
