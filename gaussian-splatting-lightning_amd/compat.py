"""Import shims for the reference's un-vendored native helpers that this package replaces.

The reference imports them by their own module names inside functions, e.g.
`from simple_knn._C import distCUDA2` (internal/models/vanilla_gaussian.py:122) or `from fused_ssim import fused_ssim`
(internal/metrics/vanilla_metrics.py:36).  `install()` registers stand-in
modules under those names — only for packages that are NOT importable — so that the reference runs unedited once
`gspl_amd.renderers` has been imported (which the `--model.renderer gspl_amd.renderers.<Name>` option does while the
configuration is parsed, long before the model is initialised from a point cloud).
"""
from __future__ import annotations

import importlib.util
import sys
import types


def _missing(name: str) -> bool:
    if name in sys.modules:
        return False
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError):
        return True


def install() -> list:
    """Returns the names of the shim modules that were installed."""
    installed = []
    if _missing("simple_knn"):
        from . import ops
        pkg = types.ModuleType("simple_knn")
        pkg.__doc__ = "gspl_amd stand-in for simple_knn (HIP; see gspl_amd.ops.distCUDA2)"
        sub = types.ModuleType("simple_knn._C")
        sub.distCUDA2 = ops.distCUDA2
        pkg._C = sub
        pkg.__path__ = []          # a package, so that `from simple_knn._C import ...` resolves through sys.modules
        sys.modules["simple_knn"] = pkg
        sys.modules["simple_knn._C"] = sub
        installed.append("simple_knn._C")
    if _missing("fused_ssim"):
        from . import ops
        mod = types.ModuleType("fused_ssim")
        mod.__doc__ = "gspl_amd stand-in for fused_ssim (HIP; see gspl_amd.ops.fused_ssim)"
        mod.fused_ssim = ops.fused_ssim
        sys.modules["fused_ssim"] = mod
        installed.append("fused_ssim")
    return installed
