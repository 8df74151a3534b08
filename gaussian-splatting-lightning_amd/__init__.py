"""
gaussian-splatting-lightning_amd — MI355X-native (gfx950) differentiable Gaussian-splatting rasterizer
behind the `Renderer` plugin interface of yzslab/gaussian-splatting-lightning.

Import name: ``gspl_amd`` (the directory name carries a hyphen; ``gspl_amd.py`` at the repo root
registers this directory as that package).  Layout:

    csrc/        hand-written HIP kernels + the C-ABI (include/gspl_hip.h)  -> libgspl_hip.so
    _lib.py      ctypes binding (no fallback: raises when the library is missing)
    ops/         autograd.Function wrappers with the reference's operator signatures (one module per concern; ops._state.STATE = the run-time state)
    renderers/   Renderer plugins (vanilla / gsplat-v0 / gsplat-v1 / distributed)
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "renderers"]
