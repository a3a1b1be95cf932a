"""CPU: libgtsfm_amd.so builds for gfx950, loads, and exports every symbol include/gtsfm_amd.h declares (no compute
calls without a GPU); host-side packers work."""

import ctypes
import re

import numpy as np

from tests.conftest import REPO


def _declared_functions():
    text = (REPO / "include" / "gtsfm_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gtsfm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared_functions()
    assert "gtsfm_sp_forward" in names and "gtsfm_last_error" in names and len(names) >= 15


def test_library_exports_every_declared_symbol(built_library):
    lib = ctypes.CDLL(str(built_library))
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/gtsfm_amd.h but not exported: {missing}"


def test_ctypes_binding_covers_header(built_library):
    from gtsfm_amd.runtime import lib as L

    assert sorted(L.SIGNATURES) == _declared_functions()
    lib = L.load()
    assert lib.gtsfm_abi_version() == 1
    # the last error is per THREAD and "" on a thread that has not seen one -- read on a fresh thread so that the order in which test files ran in
    # this process cannot matter (the main thread may carry the message of another test's deliberately bad call)
    import threading

    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.gtsfm_last_error()))
    t.start()
    t.join()
    assert seen == [b""]


def test_host_side_packers(built_library):
    from gtsfm_amd.runtime import lib as L

    lib = L.load()
    n, k = 70, 24
    w = np.arange(n * k, dtype=np.float32).reshape(n, k)
    out = np.full(lib.gtsfm_packed_linear_floats(k, n), -1, dtype=np.float32)
    assert out.size == 2 * (k // 8) * 512
    assert lib.gtsfm_pack_linear(w.ctypes.data, k, k, n, out.ctypes.data) == 0
    packed = out.reshape(2, k // 8, 2, 64, 4)  # [n_block][k_step][wave_n][lane][e]
    for nb, s, wn, lane, e in [(0, 0, 0, 0, 0), (0, 2, 1, 37, 3), (1, 1, 0, 5, 2), (1, 0, 0, 40, 1)]:
        row = nb * 64 + wn * 32 + (lane & 31)
        col = s * 8 + (lane >> 5) * 4 + e
        expect = w[row, col] if row < n else 0.0
        assert packed[nb, s, wn, lane, e] == expect
    # bad arguments are reported through the error string, not by crashing
    assert lib.gtsfm_pack_linear(w.ctypes.data, k, k + 1, n, out.ctypes.data) != 0
    assert b"pack_linear" in lib.gtsfm_last_error()


def test_superpoint_weight_packing(built_library):
    from gtsfm_amd.runtime.superpoint_engine import pack_superpoint_weights
    from gtsfm_amd.utils import synthetic

    sd = synthetic.synthetic_superpoint_state_dict()
    blob = pack_superpoint_weights(sd)
    assert blob.dtype == np.float32 and np.isfinite(blob).all()
    # conv1a section: [9][64] tap-major
    np.testing.assert_array_equal(blob[: 9 * 64].reshape(9, 64), sd["conv1a.weight"].numpy().reshape(64, 9).T)
    # total parameter mass is preserved (padding is zero)
    total = sum(float(v.double().abs().sum()) for v in sd.values())
    # the two 1x1 convolutions are stored twice: packed for the register-staged GEMM and row-major for the LDS-DMA GEMM
    total += sum(float(sd[k].double().abs().sum()) for k in ("convPb.weight", "convDb.weight"))
    assert abs(float(np.abs(blob.astype(np.float64)).sum()) - total) < 1e-6 * total
