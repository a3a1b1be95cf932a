"""Process base class: ``gtsfm.ui.gtsfm_process.GTSFMProcess`` / ``UiMetadata`` when GTSfM is importable; otherwise a
stand-in restating the registry contract (``gtsfm/ui/registry.py:15-45``, ``gtsfm/ui/gtsfm_process.py:36-65``): every
subclass is registered by ``__name__`` at class-definition time and exposes a static ``get_ui_metadata()``."""

from __future__ import annotations

import abc
import threading
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

try:  # pragma: no cover
    from gtsfm.ui.gtsfm_process import GTSFMProcess, UiMetadata  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001

    @dataclass(frozen=True, order=True)
    class UiMetadata:  # type: ignore[no-redef]
        display_name: str
        input_products: Tuple[str, ...]
        output_products: Tuple[str, ...]
        parent_plate: Optional[str] = None

    class _Registry(abc.ABCMeta):
        REGISTRY: Dict[str, type] = {}

        def __new__(mcs, name, bases, attrs):
            cls = super().__new__(mcs, name, bases, attrs)
            mcs.REGISTRY[cls.__name__] = cls
            return cls

        @classmethod
        def get_registry(mcs) -> Dict[str, type]:
            return dict(mcs.REGISTRY)

    class GTSFMProcess(metaclass=_Registry):  # type: ignore[no-redef]
        @staticmethod
        @abc.abstractmethod
        def get_ui_metadata() -> UiMetadata:
            ...


# The plugins build their device engine on first use, in the worker process. A worker with several threads (--threads_per_worker)
# may make the first calls at the same time: one of them builds, the others wait (module-level: plugin objects must stay picklable).
MODEL_LOAD_LOCK = threading.Lock()


def warn_if_cpu_requested(use_cuda: bool, plugin: str) -> None:
    """``use_cuda=False`` (the reference's own contract tests construct the plugins that way, and the reference then runs its torch
    model on the CPU): this package has no CPU path. Called once per plugin INSTANCE, when it builds its device engine.

    * ``GTSFM_AMD_STRICT_USE_CUDA=1``: ``use_cuda=False`` raises ``RuntimeError`` -- for deployments that use the flag to keep a
      worker off the device or expect CPU-bit-exact results.
    * default: with a GPU present the plugin runs on it, emits a ``RuntimeWarning`` and a WARNING log record naming the instance's
      plugin -- outputs are the reference's within the parity tolerances on either device, so GTSfM's tests and configs work
      unchanged on a GPU box; without a GPU the first call raises ``RuntimeError`` (``require_gpu``), never a silent fallback.
    INTEGRATION.md section 1 documents this as a contract deviation."""
    if use_cuda:
        return
    import logging
    import os
    import warnings

    import torch

    if os.environ.get("GTSFM_AMD_STRICT_USE_CUDA", "0") == "1":
        raise RuntimeError(f"gtsfm_amd.{plugin}: use_cuda=False was requested and GTSFM_AMD_STRICT_USE_CUDA=1: this implementation has no CPU path")
    if torch.cuda.is_available():
        msg = f"gtsfm_amd.{plugin}: use_cuda=False was requested, but this implementation has no CPU path; running on the GPU (GTSFM_AMD_STRICT_USE_CUDA=1 makes this an error)."
        logging.getLogger("gtsfm_amd").warning(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
