"""Camera intrinsics at the verifier boundary. GTSfM passes gtsam calibration objects (``CALIBRATION_TYPE``,
``gtsfm/common/types.py``; ``Cal3Bundler`` throughout the deep front-end); gtsam cannot be imported in the build container,
so a stand-in with the two methods the verifier path reads -- ``K()`` and ``calibrate(uv)`` -- is provided. The real gtsam
classes are accepted unchanged wherever GTSfM is installed."""

from __future__ import annotations

import numpy as np


class PinholeIntrinsics:
    """fx, fy, principal point; no distortion (``Cal3Bundler(fx, 0, 0, u0, v0)``)."""

    def __init__(self, fx: float = 1.0, u0: float = 0.0, v0: float = 0.0, fy: float = None):
        self.fx, self.fy, self.u0, self.v0 = float(fx), float(fx if fy is None else fy), float(u0), float(v0)

    def K(self) -> np.ndarray:  # noqa: N802 - gtsam's name
        return np.array([[self.fx, 0.0, self.u0], [0.0, self.fy, self.v0], [0.0, 0.0, 1.0]])

    def calibrate(self, uv: np.ndarray) -> np.ndarray:
        uv = np.asarray(uv, dtype=np.float64).reshape(2)
        return np.array([(uv[0] - self.u0) / self.fx, (uv[1] - self.v0) / self.fy])


# calibration classes whose ``calibrate`` is ((u - cx) / fx, (v - cy) / fy) once their distortion coefficients and skew are zero
# (gtsfm/common/types.py CALIBRATION_TYPE also lists Cal3Fisheye: an equidistant projection even with k1..k4 = 0 -- never pure)
_PINHOLE_FAMILY = ("PinholeIntrinsics", "Cal3Bundler", "Cal3_S2", "Cal3DS2")
_DISTORTION_GETTERS = ("k1", "k2", "k3", "k4", "p1", "p2")


def pinhole_parameters(intrinsics) -> tuple:
    """(fx, fy, cx, cy, is_pure_pinhole) of a gtsam calibration / the stand-in. Pure = a class of the pinhole family with no
    skew and every distortion coefficient it exposes equal to zero, i.e. ``calibrate`` is exactly ((u - cx) / fx, (v - cy) / fy)
    and can run on the device. Any other calibration type goes through its own ``calibrate`` on the host, as the reference does
    (``gtsfm/utils/features.py:41-51``)."""
    k = np.asarray(intrinsics.K(), dtype=np.float64)
    distortion = [getattr(intrinsics, name)() for name in _DISTORTION_GETTERS if callable(getattr(intrinsics, name, None))]
    pure = type(intrinsics).__name__ in _PINHOLE_FAMILY and k[0, 1] == 0.0 and all(float(d) == 0.0 for d in distortion)
    return float(k[0, 0]), float(k[1, 1]), float(k[0, 2]), float(k[1, 2]), bool(pure)
