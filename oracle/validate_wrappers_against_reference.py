"""ORACLE tooling (build container only): the REFERENCE'S OWN PLUGIN CLASSES, run live on the CPU, against the oracle's restatement of them and
against the committed config-1 golden vectors.

``gtsfm/frontend/detector_descriptor/superpoint.py`` and ``gtsfm/frontend/matcher/superglue_matcher.py`` cannot normally be imported here: they pull
cv2 and gtsam through ``gtsfm.utils.images`` / ``gtsfm.common.image``. Neither package is touched when the input image is already
gray (``rgb_to_gray_cv`` returns a 2-D array as it is, gtsfm/utils/images.py:31-33), so this script imports the reference's modules from
/root/reference with the ABSENT THIRD-PARTY PACKAGES replaced by inert stand-ins (the finder of validate_cache_against_reference.py; nothing of
GTSfM itself is replaced), gives ``torch.load`` the seeded synthetic checkpoints (the real ones cannot be downloaded), forces
``grid_sample(align_corners=True)`` and lets the reference's wrapper code run:

1. ``SuperPointDetectorDescriptor.detect_and_describe`` (superpoint.py:63-93: /255, model, numpy, ``filter_by_mask``, ``get_top_k``) on synthetic
   gray frames with and without a mask, with ``max_keypoints`` below and above the raw count  ==  ``oracle/superpoint_oracle.detect_and_describe``,
   the restatement every SuperPoint parity test of this repository compares the HIP path with: same coordinates in the same order, same
   responses, same descriptors, bit for bit;
2. ``SuperGlueMatcher.match`` (superglue_matcher.py:47-113: numpy -> torch dict -> model -> ``(K, 2) uint32``) on ragged synthetic sets, its
   ValueError / Exception contracts included  ==  ``oracle/superglue_oracle.match``;
3. the same two wrappers on the first Lund-door frames of ``tests/golden/lund_door_config1.npz`` (BASELINE config 1, 1135x760, cap 5000)  ==  the
   stored keypoints / responses / descriptor rows / match arrays: the golden file the ``-m gpu`` config-1 test holds the HIP plugins to is what
   the reference's wrapper classes return, not only what a restatement of them returns.

Usage: python oracle/validate_wrappers_against_reference.py [--quick] [--frames N] [--pairs N]"""

from __future__ import annotations

import argparse
import io
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
REFERENCE = Path(os.environ.get("GTSFM_REFERENCE", "/root/reference"))
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "oracle"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="synthetic cases only (seconds); the CPU test suite runs this")
    ap.add_argument("--frames", type=int, default=3, help="Lund-door frames taken through the reference SuperPoint wrapper")
    ap.add_argument("--pairs", type=int, default=1, help="Lund-door pairs taken through the reference SuperGlue wrapper (~20 s each)")
    args = ap.parse_args()
    if not (REFERENCE / "gtsfm" / "frontend" / "matcher" / "superglue_matcher.py").exists():
        raise SystemExit(f"reference plugins not found under {REFERENCE}")
    from validate_cache_against_reference import _AbsentPackages

    sys.meta_path.insert(0, _AbsentPackages())
    sys.path.insert(0, str(REFERENCE))
    import torch

    import gtsfm.frontend.detector_descriptor.superpoint as ref_sp
    import gtsfm.frontend.matcher.superglue_matcher as ref_sg
    from gtsfm.common.image import Image
    from gtsfm.common.keypoints import Keypoints

    assert Path(ref_sp.__file__).is_relative_to(REFERENCE) and Path(ref_sg.__file__).is_relative_to(REFERENCE)
    import validate_against_reference as V
    from gtsfm_amd.utils import synthetic
    from oracle import superglue_oracle, superpoint_oracle

    sp_sd = synthetic.synthetic_superpoint_state_dict()
    sg_sd = synthetic.synthetic_superglue_state_dict()
    with tempfile.NamedTemporaryFile(suffix=".pth") as fake_weights:  # the wrapper checks that the file exists (superpoint.py:49-53)
        try:
            ref_sp.SuperPointDetectorDescriptor(weights_path="/nonexistent/superpoint_v1.pth")
            raise AssertionError("missing weights must raise FileNotFoundError at construction")
        except FileNotFoundError:
            pass
        detectors = {k: ref_sp.SuperPointDetectorDescriptor(max_keypoints=k, use_cuda=False, weights_path=fake_weights.name) for k in (5000, 150)}
        with V._patched_load(sp_sd):
            for d in detectors.values():
                d._ensure_model_loaded()
    with V._patched_load(sg_sd):
        matcher = ref_sg.SuperGlueMatcher(use_cuda=False)

    def detect(det, gray, mask=None):
        with V._force_align_corners():
            kps, desc = det.detect_and_describe(Image(value_array=gray, mask=mask))
        assert type(kps) is Keypoints and kps.scales is None and desc.dtype == np.float32 and desc.shape == (len(kps), 256)
        return kps, desc

    # 1. the SuperPoint wrapper == its restatement
    rng = np.random.default_rng(5)
    for h, w, seed, use_mask in [(120, 160, 31, False), (123, 157, 32, True), (240, 320, 33, True)]:
        gray = synthetic.synthetic_gray_image(h, w, seed)
        mask = (rng.random((h, w)) < 0.6).astype(np.uint8) if use_mask else None
        for cap, det in detectors.items():
            kps, desc = detect(det, gray, mask)
            with V._force_align_corners():
                c, r, d = superpoint_oracle.detect_and_describe(sp_sd, gray, max_keypoints=cap, mask=mask)
            assert kps.coordinates.dtype == c.dtype and np.array_equal(kps.coordinates, c), "coordinates"
            assert kps.responses.dtype == r.dtype and np.array_equal(kps.responses, r), "responses"
            assert np.array_equal(desc, d), "descriptors"
            print(f"SuperPointDetectorDescriptor {h}x{w} mask={use_mask} max_keypoints={cap}: K={len(kps)}, wrapper == restatement, bit for bit")

    # 2. the SuperGlue wrapper == its restatement
    for n0, n1, seed in [(140, 90, 41), (1, 60, 42), (333, 280, 43)]:
        shp0, shp1 = (240, 320), (200, 304)
        k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(n0, n1, shp0, shp1, seed=seed)
        got = matcher.match(Keypoints(k0, responses=s0), Keypoints(k1, responses=s1), d0, d1, shp0 + (3,), shp1 + (3,))
        want = superglue_oracle.match(sg_sd, k0, k1, s0, s1, d0, d1, shp0, shp1, sinkhorn_iterations=ref_sg.DEFAULT_NUM_SINKHORN_ITERATIONS)
        assert got.dtype == want.dtype == np.uint32 and got.shape == want.shape and np.array_equal(got, want)
        print(f"SuperGlueMatcher n=({n0},{n1}): {len(got)} matches, wrapper == restatement")
    for bad, exc in (((Keypoints(k0), Keypoints(k1, responses=s1), d0, d1), ValueError), ((Keypoints(k0, responses=s0), Keypoints(k1, responses=s1), d0[:, :128], d1), Exception)):
        try:
            matcher.match(*bad, shp0 + (3,), shp1 + (3,))
            raise AssertionError("the wrapper must refuse this input")
        except exc:
            pass
    print("SuperGlueMatcher: ValueError without responses, Exception on 128-dimensional descriptors")
    if args.quick:
        print("OK (quick)")
        return

    # 3. the config-1 golden vectors == what the reference's wrapper classes return
    from PIL import Image as PILImage

    g = np.load(REPO / "tests" / "golden" / "lund_door_config1.npz")
    assert ref_sg.DEFAULT_NUM_SINKHORN_ITERATIONS == 20 and int(g["max_keypoints"]) == 5000
    feats = []
    for i in range(max(args.frames, 2 if args.pairs else 0)):
        gray = np.asarray(PILImage.open(io.BytesIO(g[f"gray_png_{i}"].tobytes())), dtype=np.uint8)
        kps, desc = detect(detectors[5000], gray)
        assert np.array_equal(kps.coordinates, g[f"keypoints_{i}"].astype(np.float32)) and np.array_equal(kps.responses, g[f"scores_{i}"])
        assert np.array_equal(desc[: g[f"descriptors_head_{i}"].shape[0]], g[f"descriptors_head_{i}"])
        feats.append((kps, desc, gray.shape))
        print(f"lund door frame {i} ({gray.shape[1]}x{gray.shape[0]}): the reference wrapper returns the golden's {len(kps)} keypoints, responses and descriptor rows")
    pairs = [(i, j) for i in range(len(feats)) for j in range(i + 1, len(feats))][: args.pairs]
    for i, j in pairs:
        got = matcher.match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], feats[i][2] + (3,), feats[j][2] + (3,))
        assert got.dtype == np.uint32 and np.array_equal(got, g[f"match_indices_{i}_{j}"].astype(np.uint32))
        print(f"lund door pair ({i},{j}): the reference wrapper returns the golden's {len(got)} matches")
    print("OK")


if __name__ == "__main__":
    main()
