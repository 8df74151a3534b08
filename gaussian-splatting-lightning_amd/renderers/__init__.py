"""Renderer plugins: `--model.renderer gspl_amd.renderers.<Name>`."""
# Stand-ins for the reference's native packages that are not installed (diff_gaussian_rasterization, the gsplat fork, simple_knn,
# fused_ssim): registered under their own module names FIRST — `internal/renderers/__init__.py` imports
# `diff_gaussian_rasterization` at import time (vanilla_renderer.py:14), so on a machine without the CUDA packages the reference's
# own `Renderer` base class (which the plugins must subclass inside the reference, gaussian_splatting.py:75-77) is only importable
# once the stand-in exists.  (A whole run on such a machine starts through `python -m gspl_amd.launch main.py fit ...`, which does
# the same before the reference's entry point is imported.)
from .. import compat as _compat

_compat.install()

from .renderer import Renderer, RendererConfig, RendererOutputInfo, RendererOutputTypes  # noqa: F401,E402
from .hip_vanilla_renderer import HipVanillaRenderer  # noqa: F401,E402
from .hip_gsplat_renderer import HipGSplatRenderer  # noqa: F401,E402
from .hip_pypreprocess_gsplat_renderer import HipPythonPreprocessGSplatRenderer  # noqa: F401,E402
from .hip_gsplat_v1_renderer import HipGSplatV1Renderer, HipGSplatV1RendererModule, GSplatV1  # noqa: F401,E402
from .hip_gsplat_hit_pixel_count_renderer import HipGSplatHitPixelCountRenderer  # noqa: F401,E402
from .hip_gsplat_distributed_renderer import HipGSplatDistributedRenderer, HipGSplatDistributedRendererImpl  # noqa: F401,E402
