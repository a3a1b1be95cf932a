"""Expected outputs for ``tests/abi/abi_lightglue_from_c.c`` -- TEST INFRASTRUCTURE (never imported by the product path).

The C program drives the matcher side of the C ABI without Python: ``gtsfm_blob_floats`` / ``gtsfm_pack_blob`` (weights), ``gtsfm_match_desc_ints`` /
``gtsfm_match_build_desc`` (batch descriptor), ``gtsfm_lg_workspace_bytes``, ``gtsfm_lg_forward`` -- the entry points a foreign binding of
``gtsfm/frontend/matcher/lightglue_matcher.py:75-112`` would call. It builds a two-layer LightGlue from a counter-based integer hash DIRECTLY in the
logical form the blob packer takes (q | k | v head-major ``Wqkv``, fused ``to_qk | to_v``, the attention output projections folded into ``ffn.0``) and
two keypoint / descriptor sets from the same hash. This script builds the same numbers in numpy, turns them into an UPSTREAM-layout ``state_dict`` whose
preparation (``matcher_engine.lightglue_entries``) reproduces the logical entries bit for bit -- ``out_proj`` / ``to_out`` are identities with zero bias, so
the float64 fold is exact; ``Wqkv`` is un-permuted -- runs ``oracle/lightglue_oracle.py`` on it and writes ``tests/abi/abi_lightglue_expected.h``: every
keypoint's match index and matching score. Adaptive depth / width are off (the C program passes the same settings).

Run (build container):  python oracle/make_abi_lightglue_expectation.py"""

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import lightglue_oracle as lgo  # noqa: E402
from oracle.make_abi_model_expectation import hash32  # noqa: E402

LAYERS, N0, N1, MATCHED = 2, 300, 280, 200
H, W = 480, 640
SEED = 777


def unit(t: int, n: int, seed: int) -> np.ndarray:
    return (hash32(t, np.arange(n), seed) >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0) - np.float32(0.5)


def entry_specs():
    """(kind, n, k, weight scale, bias scale, offset) of the blob entries, in the packer's order (matcher_engine.lightglue_entries). kind 0 = linear
    (W [n][k] + bias [n]), kind 1 = raw vector of n floats (value = offset + unit * scale)."""
    s = lambda fan: np.float32(2.0 * np.sqrt(3.0 / fan))  # noqa: E731 - uniform with unit output variance for unit inputs
    specs = [(1, 64, 0, np.float32(8.0), 0.0, 0.0)]  # posenc.Wr: 32 x 2 frequencies
    for l in range(LAYERS):
        specs += [(0, 768, 256, s(256), 0.05, 0.0),                                    # Wqkv (q | k | v, head-major)
                  (0, 512, 512, s(512), 0.05, 0.0), (1, 512, 0, np.float32(0.2), 0.0, 1.0), (1, 512, 0, np.float32(0.1), 0.0, 0.0),
                  (0, 256, 512, s(512) * np.float32(0.5), 0.05, 0.0),                    # self ffn.0 (out_proj folded), LN gamma, beta, ffn.3
                  (0, 512, 256, s(256), 0.05, 0.0),                                    # cross to_qk | to_v
                  (0, 512, 512, s(512), 0.05, 0.0), (1, 512, 0, np.float32(0.2), 0.0, 1.0), (1, 512, 0, np.float32(0.1), 0.0, 0.0),
                  (0, 256, 512, s(512) * np.float32(0.5), 0.05, 0.0),                    # cross ffn.0 (to_out folded), LN gamma, beta, ffn.3
                  (0, 256, 256, s(256) * np.float32(4.0), 0.05, 0.0),                    # log_assignment.final_proj
                  (1, 256, 0, np.float32(0.5), 0.0, 0.0)]                               # matchability weight
        if l < LAYERS - 1:
            specs.append((1, 256, 0, np.float32(0.5), 0.0, 0.0))                        # token_confidence weight
    return specs


def entries(seed: int):
    out = []
    for t, (kind, n, k, ws, bs, off) in enumerate(entry_specs()):
        if kind == 0:
            w = (unit(2 * t, n * k, seed) * ws).reshape(n, k)
            b = unit(2 * t + 1, n, seed) * np.float32(bs)
            out.append((0, w, b))
        else:
            out.append((1, np.float32(off) + unit(2 * t, n, seed) * ws, None))
    return out


MATCH_BIAS = [np.float32(0.5), np.float32(1.0)]
CONF_BIAS = [np.float32(-0.25), np.float32(0.0)]


def features(seed: int):
    perm = (np.arange(N1) * 7 + 3) % N0
    k0 = np.stack([np.floor((unit(210, N0, seed) + np.float32(0.5)) * np.float32(600.0)) + np.float32(4.0),
                   np.floor((unit(211, N0, seed) + np.float32(0.5)) * np.float32(430.0)) + np.float32(10.0)], 1).astype(np.float32)
    k1 = np.stack([np.floor((unit(212, N1, seed) + np.float32(0.5)) * np.float32(600.0)) + np.float32(4.0),
                   np.floor((unit(213, N1, seed) + np.float32(0.5)) * np.float32(430.0)) + np.float32(10.0)], 1).astype(np.float32)
    k1[:MATCHED] = k0[perm[:MATCHED]] + np.array([7.0, -5.0], dtype=np.float32)
    d0 = (unit(200, N0 * 256, seed) * np.float32(0.125)).reshape(N0, 256)
    d1 = (unit(202, N1 * 256, seed) * np.float32(0.125)).reshape(N1, 256)
    d1[:MATCHED] = d0[perm[:MATCHED]] + (unit(201, MATCHED * 256, seed) * np.float32(0.01)).reshape(MATCHED, 256)
    return k0, d0, k1, d1


def state_dict(seed: int):
    """Upstream cvg/LightGlue names and layouts whose load-time preparation gives exactly `entries(seed)`."""
    e = entries(seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    sd = {"posenc.Wr.weight": t(e[0][1].reshape(32, 2))}
    qkv_perm = np.array([(r % 256 // 64) * 192 + (r % 64) * 3 + r // 256 for r in range(768)])  # packed row r <- upstream row qkv_perm[r]
    eye, zero = np.eye(256, dtype=np.float32), np.zeros(256, dtype=np.float32)
    i = 1
    for l in range(LAYERS):
        p = f"transformers.{l}"
        wq, bq = np.empty((768, 256), np.float32), np.empty(768, np.float32)
        wq[qkv_perm], bq[qkv_perm] = e[i][1], e[i][2]
        sd[f"{p}.self_attn.Wqkv.weight"], sd[f"{p}.self_attn.Wqkv.bias"] = t(wq), t(bq)
        for blk, proj, j in (("self_attn", "out_proj", i + 1), ("cross_attn", "to_out", i + 6)):
            sd[f"{p}.{blk}.{proj}.weight"], sd[f"{p}.{blk}.{proj}.bias"] = t(eye), t(zero)
            sd[f"{p}.{blk}.ffn.0.weight"], sd[f"{p}.{blk}.ffn.0.bias"] = t(e[j][1]), t(e[j][2])
            sd[f"{p}.{blk}.ffn.1.weight"], sd[f"{p}.{blk}.ffn.1.bias"] = t(e[j + 1][1]), t(e[j + 2][1])
            sd[f"{p}.{blk}.ffn.3.weight"], sd[f"{p}.{blk}.ffn.3.bias"] = t(e[j + 3][1]), t(e[j + 3][2])
        sd[f"{p}.cross_attn.to_qk.weight"], sd[f"{p}.cross_attn.to_qk.bias"] = t(e[i + 5][1][:256]), t(e[i + 5][2][:256])
        sd[f"{p}.cross_attn.to_v.weight"], sd[f"{p}.cross_attn.to_v.bias"] = t(e[i + 5][1][256:]), t(e[i + 5][2][256:])
        a = f"log_assignment.{l}"
        sd[f"{a}.final_proj.weight"], sd[f"{a}.final_proj.bias"] = t(e[i + 10][1]), t(e[i + 10][2])
        sd[f"{a}.matchability.weight"], sd[f"{a}.matchability.bias"] = t(e[i + 11][1].reshape(1, 256)), t(np.array([MATCH_BIAS[l]], np.float32))
        i += 12
        if l < LAYERS - 1:
            sd[f"token_confidence.{l}.token.0.weight"] = t(e[i][1].reshape(1, 256))
            sd[f"token_confidence.{l}.token.0.bias"] = t(np.array([CONF_BIAS[l]], np.float32))
            i += 1
    assert i == len(e)
    return sd


def check_preparation(seed: int) -> None:
    """The product's own weight preparation on the upstream-layout dict gives the C program's logical entries, bit for bit."""
    from gtsfm_amd.runtime.matcher_engine import lightglue_entries

    got, mb, cb = lightglue_entries(state_dict(seed))
    want = entries(seed)
    assert len(got) == len(want)
    for (gk, gw, gb), (wk, ww, wb) in zip(got, want):
        assert gk == wk and np.array_equal(np.asarray(gw, dtype=np.float32), ww), "weight entry differs"
        assert (gb is None) == (wb is None) and (gb is None or np.array_equal(np.asarray(gb, dtype=np.float32), wb)), "bias entry differs"
    assert np.array_equal(mb, np.array(MATCH_BIAS, np.float32)) and np.array_equal(cb[: LAYERS - 1], np.array(CONF_BIAS[: LAYERS - 1], np.float32))


def run_oracle(seed: int):
    k0, d0, k1, d1 = features(seed)
    t = torch.from_numpy
    with torch.no_grad():
        out = lgo.lightglue_forward(state_dict(seed), t(k0)[None], t(k1)[None], t(d0)[None], t(d1)[None], (H, W), (H, W),
                                    depth_confidence=-1.0, width_confidence=-1.0, pruning_threshold=None, return_intermediates=True)
    return out


def cfloat(v) -> str:
    """A C float literal that reads back to exactly this float32 (nine significant digits; a '.' so that "8f" cannot happen)."""
    t = f"{float(v):.9g}"
    return (t if any(ch in t for ch in ".en") else t + ".0") + "f"


def main() -> None:
    torch.set_num_threads(4)
    best = None
    for seed in range(SEED, SEED + 16):
        check_preparation(seed)
        out = run_oracle(seed)
        m0, s0, s1 = out["matches0"][0].numpy(), out["matching_scores0"][0].numpy(), out["matching_scores1"][0].numpy()
        scores = out["log_assignment"][0, :-1, :-1].numpy()
        # decision gaps: mutual candidates' exp(score) against the 0.1 filter, and every row's / column's best against its runner-up (log domain)
        thr_gap = float(np.abs(np.concatenate([s0[s0 > 0], s1[s1 > 0]]) - 0.1).min())
        top2r = np.sort(scores, axis=1)[:, -2:]
        top2c = np.sort(scores, axis=0)[-2:, :]
        arg_gap = float(min((top2r[:, 1] - top2r[:, 0]).min(), (top2c[1] - top2c[0]).min()))
        nm = int((m0 > -1).sum())
        print(f"seed {seed}: {nm} matches, threshold gap {thr_gap:.2e}, smallest arg-max gap (log domain) {arg_gap:.2e}")
        score = min(thr_gap, arg_gap * 0.1)
        if nm >= 100 and (best is None or score > best[0]):
            best = (score, seed, out)
    score, seed, out = best
    m0, m1 = out["matches0"][0].numpy(), out["matches1"][0].numpy()
    s0, s1 = out["matching_scores0"][0].numpy(), out["matching_scores1"][0].numpy()
    specs = entry_specs()
    lines = [
        "/* GENERATED by oracle/make_abi_lightglue_expectation.py -- expected outputs of gtsfm_lg_forward for the hash-built two-layer LightGlue and feature",
        " * sets of tests/abi/abi_lightglue_from_c.c, computed by oracle/lightglue_oracle.py (fp32) on the upstream-layout state_dict whose load-time",
        f" * preparation gives the same logical entries bit for bit. {int((m0 > -1).sum())} matches; smallest decision gap of the fixture {score:.2e}. */",
        f"#define ABI_LG_LAYERS {LAYERS}", f"#define ABI_LG_N0 {N0}", f"#define ABI_LG_N1 {N1}", f"#define ABI_LG_MATCHED {MATCHED}", f"#define ABI_LG_H {H}", f"#define ABI_LG_W {W}",
        f"#define ABI_LG_SEED {seed}u", f"#define ABI_LG_ENTRIES {len(specs)}",
        "/* kind, n, k of every blob entry, then the weight scale, bias scale and offset its values are built with */",
        "static const int abi_lg_entry_shape[ABI_LG_ENTRIES][3] = {" + ", ".join(f"{{{k}, {n}, {kk}}}" for k, n, kk, *_ in specs) + "};",
        "static const float abi_lg_entry_scale[ABI_LG_ENTRIES][3] = {" + ", ".join(f"{{{cfloat(ws)}, {cfloat(np.float32(bs))}, {cfloat(off)}}}" for _, _, _, ws, bs, off in specs) + "};",
        "static const float abi_lg_match_bias[ABI_LG_LAYERS] = {" + ", ".join(cfloat(v) for v in MATCH_BIAS) + "};",
        "static const float abi_lg_conf_bias[ABI_LG_LAYERS] = {" + ", ".join(cfloat(v) for v in CONF_BIAS) + "};",
        "static const short abi_lg_matches0[ABI_LG_N0] = {" + ", ".join(str(int(v)) for v in m0) + "};",
        "static const short abi_lg_matches1[ABI_LG_N1] = {" + ", ".join(str(int(v)) for v in m1) + "};",
        "static const float abi_lg_scores0[ABI_LG_N0] = {" + ", ".join(cfloat(v) for v in s0) + "};",
        "static const float abi_lg_scores1[ABI_LG_N1] = {" + ", ".join(cfloat(v) for v in s1) + "};",
        "",
    ]
    (REPO / "tests" / "abi" / "abi_lightglue_expected.h").write_text("\n".join(lines))
    print(f"chosen seed {seed}: wrote tests/abi/abi_lightglue_expected.h ({int((m0 > -1).sum())} matches, decision gap {score:.2e})")


if __name__ == "__main__":
    main()
