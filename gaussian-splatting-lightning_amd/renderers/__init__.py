"""Renderer plugins: `--model.renderer gspl_amd.renderers.<Name>`."""
from .renderer import Renderer, RendererConfig, RendererOutputInfo, RendererOutputTypes  # noqa: F401
from .hip_vanilla_renderer import HipVanillaRenderer  # noqa: F401
from .hip_gsplat_renderer import HipGSplatRenderer  # noqa: F401
from .hip_pypreprocess_gsplat_renderer import HipPythonPreprocessGSplatRenderer  # noqa: F401
from .hip_gsplat_v1_renderer import HipGSplatV1Renderer, HipGSplatV1RendererModule, GSplatV1  # noqa: F401
from .hip_gsplat_hit_pixel_count_renderer import HipGSplatHitPixelCountRenderer  # noqa: F401
from .hip_gsplat_distributed_renderer import HipGSplatDistributedRenderer, HipGSplatDistributedRendererImpl  # noqa: F401

# Stand-ins for the reference's native helpers that are not installed (simple_knn): registered under their own module
# names, so the reference's `from simple_knn._C import distCUDA2` resolves to the HIP implementation without edits.
from .. import compat as _compat  # noqa: E402

_compat.install()
