"""Golden vectors for the FRONT-END SLICE of BASELINE config 5 -- TEST INFRASTRUCTURE (never imported by the product path).

Config 5 is Tanks-and-Temples Barn through ``deep_front_end.yaml`` end to end (``gtsfm/configs/deep_front_end.yaml:22-49``: SuperPoint(5000) -> LightGlue ->
``TwoViewEstimator`` with ``Ransac(use_intrinsics_in_verification=True, estimation_threshold_px=4)`` -> two-view bundle adjustment in gtsam -> averaging / BA).
Everything behind the verifier needs gtsam, real checkpoints and a dataset that is not here; what IS here are the three Barn frames of the reference's own
fixture (``tests/data/tanks_and_temples_barn/Barn/00000{1,2,3}.jpg``, 1920x1080). This script takes them through the part of ``run_2view``
(``gtsfm/two_view_estimator.py:350-397``) that precedes gtsam, exactly as the config wires it:

* the COLMAP loader's reduction to ``max_resolution: 760`` (``gtsfm/configs/loader/colmap.yaml:7`` -> 760 x 1351; restated ``cv.INTER_CUBIC`` + ``RGB2GRAY``:
  unpinned, cv2 absent) and its intrinsics: focal length from EXIF ``FocalLengthIn35mmFilm`` = 21 mm (``gtsfm/common/image.py:108-111``: 21 / 35 x max(w, h)),
  principal point at the centre, rescaled like ``LoaderBase.__rescale_intrinsics`` (``gtsfm/loader/loader_base.py:224-233``);
* the REFERENCE SuperPoint (model file run here, restatement asserted bit-exact) + the wrapper's ``get_top_k(5000)`` (``np.argpartition``);
* LightGlue by ``oracle/lightglue_oracle.py`` with upstream's defaults (depth 0.95, width 0.99, pruning above 1536 keypoints) -- the restatement, NOT the
  reference (its source is absent: PARITY UNPINNED) -- and the wrapper's output marshalling (``lightglue_matcher.py:104-112``);
* the verifier by ``oracle/verifier_oracle.py`` (unpinned towards OpenCV's USAC by construction) with the per-pair seed the batched path uses.

Weights are the seeded synthetic ones (no checkpoints offline): the matches are matches of random-weight descriptors, NOT correspondences of the scene, so no
pose is compared with the COLMAP ground truth of the fixture. What the fixture pins is that the HIP plugins reproduce this chain on real photographs of a
second dataset: keypoints, match arrays, verified index arrays. Output: ``tests/golden/barn_config5_frontend.npz``.

Run (build container, ~1 min of CPU):  python oracle/make_barn_config5_golden.py"""

import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from gtsfm_amd.utils import synthetic  # noqa: E402
from oracle import imageprep_oracle, lightglue_oracle, superpoint_oracle, verifier_oracle  # noqa: E402
from oracle.validate_against_reference import REFERENCE, _force_align_corners, _png_bytes, reference_superpoint  # noqa: E402

MAX_RESOLUTION = 760   # gtsfm/configs/loader/colmap.yaml:7
MAX_KEYPOINTS = 5000   # gtsfm/configs/deep_front_end.yaml:29
THRESHOLD_PX = 4.0     # gtsfm/configs/deep_front_end.yaml:48
FOCAL_35MM = 21.0      # EXIF FocalLengthIn35mmFilm of the three frames
DESC_HEAD = 64


def intrinsics_after_loader(full_h: int, full_w: int, new_h: int, new_w: int):
    """(fx, fy, cx, cy) as the loader hands them to run_2view: image.py:108-111 at full resolution, loader_base.py:224-233 rescaled (Cal3Bundler: one f)."""
    f_full = FOCAL_35MM / 35.0 * max(full_w, full_h)
    scale_u, scale_v = new_w / full_w, new_h / full_h
    return (f_full * scale_u, f_full * scale_u, (full_w / 2) * scale_u, (full_h / 2) * scale_v)


def main() -> None:
    from PIL import Image as PILImage

    torch.set_num_threads(8)
    folder = REFERENCE / "tests" / "data" / "tanks_and_temples_barn" / "Barn"
    names = sorted(p.name for p in folder.glob("*.jpg"))
    assert names == ["000001.jpg", "000002.jpg", "000003.jpg"], names
    sp_sd = synthetic.synthetic_superpoint_state_dict()
    lg_sd = synthetic.synthetic_lightglue_state_dict()
    sp = reference_superpoint(sp_sd)
    out = {"names": np.array(names), "max_resolution": MAX_RESOLUTION, "threshold_px": THRESHOLD_PX}
    feats, intr = [], []
    for i, name in enumerate(names):
        rgb = np.asarray(PILImage.open(folder / name).convert("RGB"), dtype=np.uint8)
        new_h, new_w = imageprep_oracle.downsampled_size(rgb.shape[0], rgb.shape[1], MAX_RESOLUTION)
        gray = imageprep_oracle.rgb_to_gray_u8(imageprep_oracle.resize_inter_cubic_u8(rgb, new_h, new_w))
        img = superpoint_oracle.gray_u8_to_tensor(gray)
        with torch.no_grad(), _force_align_corners():
            ref = sp({"image": img})
            ora = superpoint_oracle.superpoint_forward(sp_sd, img)
        kp, sc, de = ref["keypoints"][0], ref["scores"][0], ref["descriptors"][0]
        assert torch.equal(kp, ora["keypoints"]) and torch.equal(sc, ora["scores"]) and torch.equal(de, ora["descriptors"])
        coords, resp, desc = kp.numpy(), sc.numpy(), de.numpy().T
        k_raw = coords.shape[0]
        sel = np.argpartition(-resp, MAX_KEYPOINTS)[:MAX_KEYPOINTS] if k_raw > MAX_KEYPOINTS else np.arange(k_raw)
        coords, resp, desc = coords[sel], resp[sel], desc[sel]
        print(f"barn {name}: {rgb.shape[0]}x{rgb.shape[1]} -> {gray.shape[0]}x{gray.shape[1]}, K_raw = {k_raw} -> {len(sel)}; restatement bit-exact with the reference model file")
        feats.append((coords, resp, desc))
        intr.append(intrinsics_after_loader(rgb.shape[0], rgb.shape[1], new_h, new_w))
        out[f"gray_png_{i}"] = _png_bytes(gray)
        out[f"k_raw_{i}"] = k_raw
        out[f"sel_{i}"] = sel.astype(np.uint16 if k_raw < 65536 else np.uint32)
        out[f"keypoints_{i}"] = coords.astype(np.int16)
        out[f"scores_{i}"] = resp
        out[f"descriptors_head_{i}"] = desc[:DESC_HEAD].copy()
        out["height"], out["width"] = gray.shape
    out["intrinsics"] = np.array(intr, dtype=np.float64)
    shape = (int(out["height"]), int(out["width"]), 3)
    for i, j in ((0, 1), (0, 2), (1, 2)):
        (k0, _, d0), (k1, _, d1) = feats[i], feats[j]
        t = torch.from_numpy
        with torch.no_grad():
            res = lightglue_oracle.lightglue_forward(lg_sd, t(k0)[None].float(), t(k1)[None].float(), t(d0)[None].float(), t(d1)[None].float(), shape[:2], shape[:2])
        matches = res["matches"].numpy().astype(np.int64)  # lightglue_matcher.py:107-112: rbd(...)["matches"], (K, 2), image-i1 keypoint order
        ver = verifier_oracle.verify(k0, k1, matches, intr[i], intr[j], THRESHOLD_PX, seed=(i << 32) | j)
        out[f"matches_{i}_{j}"] = matches.astype(np.int16)
        out[f"matches0_{i}_{j}"] = res["matches0"][0].numpy().astype(np.int16)
        out[f"matching_scores0_{i}_{j}"] = res["matching_scores0"][0].numpy().astype(np.float32)
        out[f"stop_{i}_{j}"] = int(res["stop"])
        out[f"v_corr_idxs_{i}_{j}"] = np.asarray(ver["v_corr_idxs"]).astype(np.int16).reshape(-1, 2)
        out[f"inlier_ratio_{i}_{j}"] = float(ver["inlier_ratio"])
        out[f"has_pose_{i}_{j}"] = ver["R"] is not None
        out[f"R_{i}_{j}"] = np.zeros((3, 3)) if ver["R"] is None else np.asarray(ver["R"], dtype=np.float64)
        out[f"t_{i}_{j}"] = np.zeros(3) if ver["t"] is None else np.asarray(ver["t"], dtype=np.float64).reshape(3)
        near = np.abs(res["matching_scores0"][0].numpy() - 0.1)
        print(f"barn pair ({i}, {j}): layers run {int(res['stop'])}, {len(matches)} matches (closest score to the 0.1 filter: {near[near > 0].min():.2e}), "
              f"{len(out[f'v_corr_idxs_{i}_{j}'])} verified, inlier ratio {out[f'inlier_ratio_{i}_{j}']:.3f}")
    path = REPO / "tests" / "golden" / "barn_config5_frontend.npz"
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({path.stat().st_size / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
