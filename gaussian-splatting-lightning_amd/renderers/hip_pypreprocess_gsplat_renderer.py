"""HipPythonPreprocessGSplatRenderer — drop-in for the reference's `PythonPreprocessGSplatRenderer`
(internal/renderers/pypreprocess_gsplat_renderer.py:8-66; configs/pypreprocess_gsplat.yaml = BASELINE.json configs[0]).

The reference's class is the CPU-runnable configuration: projection in PyTorch (`internal/utils/gaussian_projection.py`), SH and
rasterization in gsplat's native ops.  Here all three stages are the HIP ops; what is kept is the CONTRACT of that class — its two
options (`block_size`, `anti_aliased`), no render types, and its output dictionary, which differs from `GSPlatRenderer`'s:
`viewspace_points_grad_scale` is the scalar 0.5 * max(H, W) (line 63) and `visibility_filter` is the projection's mask
(depth >= near AND at least one tile, line 64), not `radii > 0` recomputed by the caller.
"""
from __future__ import annotations

import torch

from .. import ops
from .hip_gsplat_renderer import DEFAULT_ANTI_ALIASED_STATUS, DEFAULT_BLOCK_SIZE, _project
from .renderer import Renderer, camera_hw, implementation_tile_size, model_sh_pair


class HipPythonPreprocessGSplatRenderer(Renderer):
    block_size: int = DEFAULT_BLOCK_SIZE

    anti_aliased: bool = DEFAULT_ANTI_ALIASED_STATUS

    def __init__(self) -> None:
        super().__init__()

    def forward(self, viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, **kwargs):
        img_width, img_height = camera_hw(viewpoint_camera)
        xys, depths, radii, conics, comp, num_tiles_hit, _ = _project(
            pc.get_xyz, pc.get_scaling, pc.get_rotation, viewpoint_camera, scaling_modifier, self.block_size, img_width, img_height)
        # the culled rows of the projection carry radius 0 and zero tiles: the mask the reference returns next to them
        mask = radii > 0
        # view directions, SH, +0.5 and the clamp in one kernel (pypreprocess_gsplat_renderer.py:37-40)
        rgbs = ops.sh_view_colors(pc.active_sh_degree, pc.get_xyz, viewpoint_camera.camera_center, *model_sh_pair(pc), mask)
        opacities = pc.get_opacity
        if self.anti_aliased is True:
            opacities = opacities * comp[:, None]
        rgb = ops.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, rgbs, opacities, img_height=img_height,
                                      img_width=img_width, block_width=implementation_tile_size(self.block_size), background=bg_color, return_alpha=False,
                                      channels_first=True)
        return {
            "render": rgb,
            "viewspace_points": xys,
            "viewspace_points_grad_scale": 0.5 * max(img_height, img_width),
            "visibility_filter": mask,
            "radii": radii,
        }
