"""Expected outputs for ``tests/abi/abi_model_from_c.c`` -- TEST INFRASTRUCTURE (never imported by the product path).

The C program drives the MODEL-LEVEL entry points of the C ABI (``gtsfm_sp_pack_weights`` -> ``gtsfm_sp_workspace_bytes`` ->
``gtsfm_sp_forward``) without Python: it builds the 24 SuperPoint tensors and a gray image from a counter-based integer hash (so that C and
numpy produce the same bits with no file in between) and compares what comes back with the numbers THIS script writes into
``tests/abi/abi_model_expected.h``: the keypoint count, every keypoint's (x, y), every score and the first descriptor values, computed by
``oracle/superpoint_oracle.py`` (the restatement pinned bit-exact on the reference's ``superpoint.py``) from the same hash.
It also reports how far the fixture sits from a flip (score gaps at the 0.005 threshold and inside the 9 x 9 NMS windows): the seed below was
chosen so that no decision depends on the sixth digit of a score.

Run (build container):  python oracle/make_abi_model_expectation.py"""

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import superpoint_oracle as spo  # noqa: E402

H, W = 96, 128
SEED = 12345
LAYERS = [("conv1a", 64, 1, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3), ("conv3a", 128, 64, 3), ("conv3b", 128, 128, 3),
          ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3), ("convPa", 256, 128, 3), ("convPb", 65, 256, 1), ("convDa", 256, 128, 3), ("convDb", 256, 256, 1)]
GAIN = {"convPb": 6.0}  # spread the detector logits: keypoint decisions far from ties


def hash32(t: int, i: np.ndarray, seed: int) -> np.ndarray:
    """Counter-based: element i of tensor t. uint32 arithmetic (wraps), murmur3's finaliser."""
    x = (i.astype(np.uint64) * 2654435761 + t * 40503 + seed) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x.astype(np.uint32)


def unit(t: int, n: int, seed: int) -> np.ndarray:
    """[-0.5, 0.5) on a 2^-24 grid: exact in float32 on both sides."""
    return (hash32(t, np.arange(n), seed) >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0) - np.float32(0.5)


def scales():
    out = []
    for name, cout, cin, k in LAYERS:
        out.append(np.float32(2.0 * np.sqrt(6.0 / (cin * k * k)) * GAIN.get(name, 1.0)))  # Kaiming-uniform bound, as a float32 literal both sides use
        out.append(np.float32(0.1))
    return out


def tensors(seed: int):
    sc = scales()
    sd = {}
    for li, (name, cout, cin, k) in enumerate(LAYERS):
        w = unit(2 * li, cout * cin * k * k, seed) * sc[2 * li]
        b = unit(2 * li + 1, cout, seed) * sc[2 * li + 1]
        sd[f"{name}.weight"] = torch.from_numpy(w.reshape(cout, cin, k, k).copy())
        sd[f"{name}.bias"] = torch.from_numpy(b.copy())
    return sd


def image(seed: int) -> np.ndarray:
    """Blocky texture: an 8 x 8-pixel cell pattern plus pixel noise, uint8, integer arithmetic only."""
    y, x = np.mgrid[0:H, 0:W]
    cell = hash32(100, ((y // 8) * 64 + (x // 8)).reshape(-1), seed).reshape(H, W) >> 25       # 0 .. 127 per cell
    fine = hash32(101, (y * W + x).reshape(-1), seed).reshape(H, W) >> 26                        # 0 .. 63 per pixel
    ramp = ((x * 3 + y * 5) >> 2) & 63
    return ((cell + fine + ramp) & 255).astype(np.uint8)


def cfloat(v) -> str:
    """A C float literal that reads back to exactly this float32 (nine significant digits; a '.' so that "1f" cannot happen)."""
    t = f"{float(v):.9g}"
    return (t if any(ch in t for ch in ".en") else t + ".0") + "f"


def main() -> None:
    torch.set_num_threads(4)
    best = None
    for seed in range(SEED, SEED + 24):
        sd, img = tensors(seed), image(seed)
        with torch.no_grad():
            out = spo.superpoint_forward(sd, spo.gray_u8_to_tensor(img), return_intermediates=True)
        k = out["keypoints"].shape[0]
        dense = out["dense_scores"][0].numpy()
        nms = out["nms_scores"][0].numpy()
        sc = out["scores"].numpy()
        thr_gap = float(np.abs(nms[nms > 0] - 0.005).min())  # only maxima that survive the NMS meet the threshold
        # smallest gap between a surviving maximum and the runner-up inside its 9 x 9 window
        gaps = []
        for (xx, yy) in out["keypoints"].numpy().astype(int):
            win = dense[max(0, yy - 4) : yy + 5, max(0, xx - 4) : xx + 5].copy().reshape(-1)
            win.sort()
            gaps.append(float(win[-1] - win[-2]))
        # and between any two dense scores within a window that could tie (global check: relative gap of the closest pair among local maxima)
        print(f"seed {seed}: K = {k}, scores {sc.min():.4f} .. {sc.max():.4f}, threshold gap {thr_gap:.2e}, min NMS window gap {min(gaps):.2e}")
        score = min(thr_gap, min(gaps))
        if 40 <= k <= 400 and (best is None or score > best[0]):
            best = (score, seed, sd, img, out)
    score, seed, sd, img, out = best
    print(f"chosen seed {seed}: min decision gap {score:.2e}")
    kp = out["keypoints"].numpy().astype(np.int32)
    sc = out["scores"].numpy()
    de = out["descriptors"].numpy().T  # [K, 256]
    lines = [
        "/* GENERATED by oracle/make_abi_model_expectation.py -- expected outputs of gtsfm_sp_forward for the hash-built SuperPoint weights and image of",
        " * tests/abi/abi_model_from_c.c, computed by oracle/superpoint_oracle.py (restatement pinned bit-exact on the reference's superpoint.py).",
        f" * Smallest score gap any keypoint decision of this fixture depends on: {score:.2e} (the HIP path agrees with the oracle to ~3e-6). */",
        f"#define ABI_MODEL_H {H}", f"#define ABI_MODEL_W {W}", f"#define ABI_MODEL_SEED {seed}u", f"#define ABI_MODEL_K {len(kp)}",
        "static const float abi_model_scales[24] = {" + ", ".join(cfloat(s) for s in scales()) + "};",
        "static const int abi_model_shapes[12][3] = {" + ", ".join(f"{{{cout}, {cin}, {k}}}" for _, cout, cin, k in LAYERS) + "};  /* cout, cin, kernel */",
        "static const short abi_model_xy[ABI_MODEL_K][2] = {" + ", ".join(f"{{{int(a)}, {int(b)}}}" for a, b in kp) + "};",
        "static const float abi_model_scores[ABI_MODEL_K] = {" + ", ".join(cfloat(s) for s in sc) + "};",
        "static const float abi_model_desc_head[ABI_MODEL_K][4] = {" + ", ".join("{" + ", ".join(cfloat(v) for v in row[:4]) + "}" for row in de) + "};",
        "",
    ]
    (REPO / "tests" / "abi" / "abi_model_expected.h").write_text("\n".join(lines))
    print(f"wrote tests/abi/abi_model_expected.h: K = {len(kp)}")


if __name__ == "__main__":
    main()
