"""Stand-in for `GSplatHitPixelCountRenderer` (reference: internal/renderers/gsplat_hit_pixel_count_renderer.py), the helper
LightGaussian-style pruning calls once per training camera (internal/utils/light_gaussian.py:37-50)."""
from typing import Optional

import torch

from .. import ops
from .renderer import Renderer, implementation_tile_size
from .hip_gsplat_renderer import HipGSplatRenderer
from .renderer import camera_hw


class HipGSplatHitPixelCountRenderer(Renderer):
    @staticmethod
    def hit_pixel_count(
            means3D: torch.Tensor,
            opacities: torch.Tensor,
            scales: Optional[torch.Tensor],
            rotations: Optional[torch.Tensor],      # normalised by the caller, as in the reference
            viewpoint_camera,
            scaling_modifier=1.0,
            anti_aliased: bool = True,
            block_size: int = 16,
            extra_projection_kwargs: dict = None,
    ):
        """-> (count [N] i32, opacity_score, alpha_score, visibility_score [N] f32) of one view."""
        with torch.no_grad():
            xys, depths, radii, conics, comp, num_tiles_hit, cov3d = HipGSplatRenderer.project(
                means3D=means3D, scales=scales, rotations=rotations, viewpoint_camera=viewpoint_camera,
                scaling_modifier=scaling_modifier, block_size=block_size, extra_projection_kwargs=extra_projection_kwargs)
            if anti_aliased is True:
                opacities = opacities * comp[:, None]
            W, H = camera_hw(viewpoint_camera)
            return ops.hit_pixel_count(xys, depths, radii, conics, num_tiles_hit, opacities, img_height=H, img_width=W,
                                       block_width=implementation_tile_size(block_size))
