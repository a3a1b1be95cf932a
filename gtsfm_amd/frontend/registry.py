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
    model on the CPU): this package has no CPU path. With a GPU present the plugin runs on it and says so once -- outputs are the
    reference's within the parity tolerances on either device, so GTSfM's tests and configs work unchanged on a GPU box; without a
    GPU the first call raises ``RuntimeError`` (``require_gpu``), never a silent fallback."""
    if use_cuda:
        return
    import warnings

    import torch

    if torch.cuda.is_available():
        warnings.warn(f"gtsfm_amd.{plugin}: use_cuda=False was requested, but this implementation has no CPU path; running on the GPU.",
                      RuntimeWarning, stacklevel=3)
