"""Renderer plugins: `--model.renderer gspl_amd.renderers.<Name>`."""
from .renderer import Renderer, RendererConfig, RendererOutputInfo, RendererOutputTypes  # noqa: F401
from .hip_vanilla_renderer import HipVanillaRenderer  # noqa: F401
from .hip_gsplat_renderer import HipGSplatRenderer  # noqa: F401
from .hip_gsplat_v1_renderer import HipGSplatV1Renderer, HipGSplatV1RendererModule, GSplatV1  # noqa: F401
from .hip_gsplat_distributed_renderer import HipGSplatDistributedRenderer, HipGSplatDistributedRendererImpl  # noqa: F401
