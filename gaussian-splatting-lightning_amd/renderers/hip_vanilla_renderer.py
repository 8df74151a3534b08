"""HipVanillaRenderer — drop-in for the reference's default `VanillaRenderer`
(internal/renderers/vanilla_renderer.py:17-213), backed by the HIP `GaussianRasterizer` of ops/inria.py
instead of `diff_gaussian_rasterization`.

Select with   --model.renderer gspl_amd.renderers.HipVanillaRenderer   (INTEGRATION.md).
Output contract (vanilla_renderer.py:122-129): `render` [3,H,W], `viewspace_points` (zeros [N,3] whose
`.grad[:, :2]` receives the NDC-scaled screen-space gradient), `visibility_filter`, `radii`.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .. import ops
from .renderer import Renderer, RendererOutputInfo, RendererOutputTypes, camera_hw, camera_scalars, model_sh_pair, model_raw_parameters


def _tan_half(fov):
    return math.tan(float(fov) * 0.5)


class HipVanillaRenderer(Renderer):
    def __init__(self, compute_cov3D_python: bool = False, convert_SHs_python: bool = False, fuse_activations: bool = True):
        """fuse_activations (extension): a model whose getters are exp / normalize / sigmoid of stored parameters
        (`renderer.model_raw_parameters`) hands the rasterizer its RAW parameters and the activations run inside the preprocess
        kernels; False: the getters are always called."""
        super().__init__()
        self.compute_cov3D_python = compute_cov3D_python
        self.convert_SHs_python = convert_SHs_python
        self.fuse_activations = fuse_activations

    @staticmethod
    def _settings(viewpoint_camera, bg_color, scaling_modifier, sh_degree):
        W, H = camera_hw(viewpoint_camera)
        fov_x, fov_y = camera_scalars(viewpoint_camera, ("fov_x", "fov_y"))
        return ops.GaussianRasterizationSettings(
            image_height=H, image_width=W,
            tanfovx=_tan_half(fov_x), tanfovy=_tan_half(fov_y),
            bg=bg_color, scale_modifier=scaling_modifier,
            viewmatrix=viewpoint_camera.world_to_camera, projmatrix=viewpoint_camera.full_projection,
            sh_degree=int(sh_degree), campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)

    def forward(self, viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
                render_types: list = None, **kwargs):
        if render_types is None:
            render_types = ["rgb"]
        assert len(render_types) == 1, "Only single type is allowed currently"

        rendered_image_key = "render"
        if "depth" in render_types:
            rendered_image_key = "depth"
            w2c = viewpoint_camera.world_to_camera
            depth = (torch.matmul(pc.get_xyz, w2c[:3, :3]) + w2c[3, :3])[:, 2:]
            bg_color = torch.zeros_like(bg_color)
            override_color = depth.repeat(1, 3)

        means3D = pc.get_xyz
        # The screen-space tensor only CARRIES the 2D-mean gradient back to the density controller; its values are never read
        # (reference: `zeros_like(...) + 0`, vanilla_renderer.py:55-56 — a fill and an add per frame).  A leaf accepts
        # `retain_grad()` and receives `.grad` just the same.
        screenspace_points = torch.empty_like(means3D, dtype=means3D.dtype, device=bg_color.device).requires_grad_(True)
        settings = self._settings(viewpoint_camera, bg_color, scaling_modifier, pc.active_sh_degree)
        rasterizer = ops.GaussianRasterizer(raster_settings=settings)

        scales = rotations = cov3D_precomp = opacities = None
        raw = model_raw_parameters(pc) if (self.fuse_activations and not self.compute_cov3D_python) else None
        if self.compute_cov3D_python:
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        elif raw is not None:
            scales, rotations, opacities = raw
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation
        if opacities is None:
            opacities = pc.get_opacity

        shs = shs_rest = colors_precomp = None
        if override_color is None:
            # shs_dc / shs_rest where the model stores them (the reference passes `pc.get_features`, a torch.cat per step,
            # vanilla_renderer.py:99-109): same coefficients, same gradients, no copy in either direction
            dc, rest = model_sh_pair(pc)
            if self.convert_SHs_python:
                # the "python" colour path of the reference, served by the fused HIP SH kernel
                colors_precomp = ops.sh_view_colors(pc.active_sh_degree, pc.get_xyz, viewpoint_camera.camera_center,
                                                    dc, rest, detach_means=False)
            else:
                shs, shs_rest = dc, rest
        else:
            colors_precomp = override_color

        rendered_image, radii = rasterizer(
            means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
            opacities=opacities, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, shs_rest=shs_rest,
            raw_parameters=raw is not None)
        visibility_filter = radii > 0
        visibility_filter._gspl_radii_positive = True      # density.HipDensityStatsMixin: the mask the fused backward applies itself
        return {
            rendered_image_key: rendered_image,
            "viewspace_points": screenspace_points,
            "visibility_filter": visibility_filter,
            "radii": radii,
        }

    @staticmethod
    def render(means3D, opacity, scales, rotations, features, active_sh_degree: int, viewpoint_camera, bg_color,
               scaling_modifier=1.0, colors_precomp: Optional[torch.Tensor] = None, cov3D_precomp: Optional[torch.Tensor] = None):
        """Static helper with the reference's signature (vanilla_renderer.py:131-207)."""
        if colors_precomp is not None:
            assert features is None
        if cov3D_precomp is not None:
            assert scales is None and rotations is None
        screenspace_points = torch.empty_like(means3D, dtype=means3D.dtype, device=means3D.device).requires_grad_(True)
        settings = HipVanillaRenderer._settings(viewpoint_camera, bg_color, scaling_modifier, active_sh_degree)
        rendered_image, radii = ops.GaussianRasterizer(raster_settings=settings)(
            means3D=means3D, means2D=screenspace_points, shs=features, colors_precomp=colors_precomp,
            opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
        visibility_filter = radii > 0
        visibility_filter._gspl_radii_positive = True
        return {"render": rendered_image, "depth": None, "viewspace_points": screenspace_points,
                "visibility_filter": visibility_filter, "radii": radii}

    def get_available_outputs(self) -> Dict:
        return {"rgb": RendererOutputInfo("render"), "depth": RendererOutputInfo("depth", RendererOutputTypes.GRAY)}
