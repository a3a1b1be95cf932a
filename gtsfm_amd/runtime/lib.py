"""ctypes binding of libgtsfm_amd.so (the C ABI declared in include/gtsfm_amd.h).

The library is loaded lazily, once per process, and never pickled: plugin objects hold only plain Python state until
their first call on a worker (the reference ships its plugins to Dask workers with ``client.scatter``,
``gtsfm/frontend/correspondence_generator/det_desc_correspondence_generator.py:65-68``).

There is NO fallback path: if the shared library is missing or a call fails, a ``RuntimeError`` is raised.
"""

from __future__ import annotations

import ctypes as C
import threading
from pathlib import Path
from typing import Optional

import os

_PKG = Path(__file__).resolve().parent.parent
# GTSFM_LIB: developer switch -- load a differently built copy of the library (kernel variants for A/B measurements)
LIB_PATH = Path(os.environ["GTSFM_LIB"]) if os.environ.get("GTSFM_LIB") else _PKG / "libgtsfm_amd.so"

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); mirrors include/gtsfm_amd.h one to one.
SIGNATURES = {
    "gtsfm_abi_version": (C.c_int, []),
    "gtsfm_last_error": (C.c_char_p, []),
    "gtsfm_packed_conv3x3_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "gtsfm_packed_linear_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "gtsfm_pack_conv3x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gtsfm_pack_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gtsfm_conv3x3_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "gtsfm_conv1_fused_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_linear_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
         C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p],
    ),
    "gtsfm_linear_rowmajor_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
         C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p],
    ),
    "gtsfm_pack_rows_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_sp_packed_weight_floats": (C.c_size_t, []),
    "gtsfm_sp_pack_weights": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "gtsfm_sp_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "gtsfm_sp_forward": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
         C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_sp_forward_masked": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
         C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_sp_softmax_d2s": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_sp_nms_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "gtsfm_sp_simple_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gtsfm_sp_extract_keypoints": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_void_p],
    ),
    "gtsfm_blob_floats": (C.c_size_t, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gtsfm_pack_blob": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "gtsfm_match_desc_ints": (C.c_size_t, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_match_build_desc": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gtsfm_move_blocks_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "gtsfm_attention_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
         C.c_int, C.c_int, C.c_float, C.c_void_p],
    ),
    "gtsfm_attention_split_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t]),
    "gtsfm_attention_split_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
         C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "gtsfm_attention_math_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int]),
    "gtsfm_attention_math_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
         C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "gtsfm_sg_workspace_bytes": (C.c_size_t, [C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_sg_forward": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_sg_forward_phase": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_prep_rgb_to_gray_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_prep_cubic_taps": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_prep_resize_cubic_u8": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    ),
    "gtsfm_sinkhorn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_sinkhorn_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_score_matrices_workspace_bytes": (C.c_size_t, [C.c_int]),
    "gtsfm_score_matrices_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "gtsfm_verify_workspace_bytes": (C.c_size_t, [C.c_longlong]),
    "gtsfm_verify_essential_f64": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
         C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_verify_fundamental_f64": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
         C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_verify_compact_matches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gtsfm_lg_assignment_workspace_bytes": (C.c_size_t, [C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_lg_assignment_f32": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_layernorm_gelu_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gtsfm_lg_workspace_bytes": (C.c_size_t, [C.c_int, C.c_void_p, C.c_void_p]),
    "gtsfm_lg_forward": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_lg_forward_phase": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
         C.c_void_p],
    ),
    "gtsfm_lg_forward_streams": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
         C.c_void_p, C.c_void_p],
    ),
    "gtsfm_sp_select_topk": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "gtsfm_sp_sample_descriptors": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    ),
}


def load() -> C.CDLL:
    """Load (once) and return the shared library; raises RuntimeError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension has not been built. Run `python -m gtsfm_amd.csrc.build` "
                "(or __graft_entry__.build()). gtsfm_amd has no CPU / PyTorch fallback."
            )
        # PyTorch-ROCm bundles its own libamdhip64.so.7; import it first so that this library binds to the SAME HIP
        # runtime instance (one device context, shared streams) instead of pulling in /opt/rocm's copy.
        import torch  # noqa: F401

        lib = C.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gtsfm_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed with code {rc}: {msg}")


def ptr(t) -> Optional[int]:
    """Device / host address of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream_handle() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
