"""HipGSplatRenderer — drop-in for the reference's `GSPlatRenderer`
(internal/renderers/gsplat_renderer.py:11-391; configs/gsplat.yaml, gsplat-absgrad.yaml,
matrixcity/gsplat-aerial.yaml) on the HIP ops: same render types, same output dict, same static
helpers (`render`, `project`, `rasterize`, `rasterize_simplified`) that ~8 other reference renderers reuse.

Differences that are deliberate:
  * view directions, SH evaluation, `+0.5` and the clamp run in ONE kernel (`ops.sh_view_colors`);
  * width/height are read back once per camera object instead of with six `.item()`s per call (`renderer.camera_scalars`);
  * `absgrad=True` asks the compositing backward for `viewspace_points.absgrad`
    (what `configs/gsplat-absgrad.yaml` needs from the density controller's point of view).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .. import ops
from .renderer import Renderer, RendererOutputInfo, RendererOutputTypes, camera_hw, viewspace_grad_scale, implementation_tile_size, model_sh_pair

DEFAULT_BLOCK_SIZE: int = 16
DEFAULT_ANTI_ALIASED_STATUS: bool = True


def _project(means3D, scales, rotations, viewpoint_camera, scaling_modifier, block_size, W, H, kernel_size=0.3, **extra):
    block_size = implementation_tile_size(block_size)
    kernel_size = extra.pop("filter_2d_kernel_size", kernel_size)
    # the full, contiguous 4x4 world->camera matrix, built once per camera object (the reference hands the kernel a [3,4]
    # transposed view per frame; completing and copying it is a launch per frame)
    # (the cache is keyed on the source tensor's identity AND version counter: a pose updated in place — pose refinement,
    # viewer edits — or replaced is seen; a pose that takes part in autograd is never cached)
    w2c = viewpoint_camera.world_to_camera
    cached = getattr(viewpoint_camera, "_gspl_viewmat", None)
    if (cached is not None and cached[0] is w2c and cached[1] == w2c._version and cached[2].device == means3D.device
            and not w2c.requires_grad):
        vm = cached[2]
    else:
        vm = w2c.T.to(device=means3D.device, dtype=torch.float32).contiguous()
        if not w2c.requires_grad:
            try:
                viewpoint_camera._gspl_viewmat = (w2c, w2c._version, vm.detach())
            except Exception:      # a frozen camera type: no cache
                pass
    return ops.project_gaussians(
        means3d=means3D, scales=scales, glob_scale=scaling_modifier, quats=rotations,
        viewmat=vm,
        fx=viewpoint_camera.fx, fy=viewpoint_camera.fy, cx=viewpoint_camera.cx, cy=viewpoint_camera.cy,
        img_height=H, img_width=W, block_width=block_size, filter_2d_kernel_size=kernel_size, return_cov3d=False, **extra)


class HipGSplatRenderer(Renderer):
    _RGB_REQUIRED = 1
    _ALPHA_REQUIRED = 1 << 1
    _ACC_DEPTH_REQUIRED = 1 << 2
    _ACC_DEPTH_INVERTED_REQUIRED = 1 << 3
    _EXP_DEPTH_REQUIRED = 1 << 4
    _EXP_DEPTH_INVERTED_REQUIRED = 1 << 5
    _INVERSE_DEPTH_REQUIRED = 1 << 6
    _HARD_DEPTH_REQUIRED = 1 << 7
    _HARD_INVERSE_DEPTH_REQUIRED = 1 << 8

    RENDER_TYPE_BITS = {
        "rgb": _RGB_REQUIRED,
        "alpha": _ALPHA_REQUIRED | _ACC_DEPTH_REQUIRED,
        "acc_depth": _ACC_DEPTH_REQUIRED,
        "acc_depth_inverted": _ACC_DEPTH_REQUIRED | _ACC_DEPTH_INVERTED_REQUIRED,
        "exp_depth": _ACC_DEPTH_REQUIRED | _EXP_DEPTH_REQUIRED,
        "exp_depth_inverted": _ACC_DEPTH_REQUIRED | _EXP_DEPTH_REQUIRED | _EXP_DEPTH_INVERTED_REQUIRED,
        "inverse_depth": _INVERSE_DEPTH_REQUIRED,
        "hard_depth": _HARD_DEPTH_REQUIRED,
        "hard_inverse_depth": _HARD_INVERSE_DEPTH_REQUIRED,
    }

    def __init__(self, block_size: int = DEFAULT_BLOCK_SIZE, anti_aliased: bool = DEFAULT_ANTI_ALIASED_STATUS,
                 kernel_size: float = 0.3, absgrad: bool = False) -> None:
        super().__init__()
        self.block_size = block_size
        self.anti_aliased = anti_aliased
        self.filter_2d_kernel_size = kernel_size
        self.absgrad = absgrad

    def parse_render_types(self, render_types: list) -> int:
        if render_types is None:
            return self._RGB_REQUIRED
        bits = 0
        for i in render_types:
            bits |= self.RENDER_TYPE_BITS[i]
        return bits

    @staticmethod
    def is_type_required(bits: int, type: int) -> bool:
        return bits & type != 0

    def forward(self, viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, render_types: list = None, **kwargs):
        bits = self.parse_render_types(render_types)
        W, H = camera_hw(viewpoint_camera)
        rot = pc.get_rotation
        xys, depths, radii, conics, comp, num_tiles_hit, cov3d = _project(
            pc.get_xyz, pc.get_scaling, rot / rot.norm(dim=-1, keepdim=True), viewpoint_camera, scaling_modifier,
            self.block_size, W, H, getattr(self, "filter_2d_kernel_size", 0.3))

        opacities = pc.get_opacity
        if self.anti_aliased is True:
            opacities = opacities * comp[:, None]
        zero1 = torch.zeros((1,), dtype=torch.float, device=bg_color.device)

        # sorted once, shared by every pass that composites with `opacities`; tile hits that cannot reach alpha >= 1/255
        # are not listed.  Passes with other opacities ("hard" depth) bin for themselves inside rasterize_gaussians.
        # The count half is launched here; the emit half after the SH kernel, which keeps the device busy while the host
        # waits for the number of intersections.
        pending = ops.bin_gaussians_begin(xys, depths, radii, H, W, implementation_tile_size(self.block_size), conics=conics, opacities=opacities)
        isects = None

        def rasterize(feats, background, return_alpha=False, opac=opacities, absgrad=False, channels_first=False):
            nonlocal isects
            if isects is None:
                isects = ops.bin_gaussians_end(pending, lazy=True)      # the list length stays on the device until compositing is enqueued
            return ops.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, feats, opac, img_height=H, img_width=W,
                                           block_width=implementation_tile_size(self.block_size), background=background, return_alpha=return_alpha,
                                           absgrad=absgrad, isects=isects if opac is opacities else None,
                                           channels_first=channels_first)

        visible = radii > 0
        rgb = None
        if self.is_type_required(bits, self._RGB_REQUIRED):
            rgbs = ops.sh_view_colors(pc.active_sh_degree, pc.get_xyz, viewpoint_camera.camera_center, *model_sh_pair(pc),
                                      visible, detach_means=True)
            # [3,H,W] straight from the kernel: the reference permutes an [H,W,3] image, which costs the loss a 25 MB copy in
            # the forward and another one for the incoming gradient in the backward
            rgb = rasterize(rgbs, bg_color, absgrad=getattr(self, "absgrad", False), channels_first=True)

        alpha = acc_depth_im = acc_depth_inverted_im = exp_depth_im = exp_depth_inverted_im = None
        if self.is_type_required(bits, self._ACC_DEPTH_REQUIRED):
            acc_depth_im, alpha = rasterize(depths.unsqueeze(-1), zero1, True)
            alpha = alpha[..., None]
            if self.is_type_required(bits, self._ACC_DEPTH_INVERTED_REQUIRED):
                acc_depth_inverted_im = torch.where(acc_depth_im > 0, 1. / acc_depth_im, acc_depth_im.detach().max()).permute(2, 0, 1)
            if self.is_type_required(bits, self._EXP_DEPTH_REQUIRED):
                exp_depth_im = torch.where(alpha > 0, acc_depth_im / alpha, acc_depth_im.detach().max()).permute(2, 0, 1)
            alpha = alpha.permute(2, 0, 1) if self.is_type_required(bits, self._ALPHA_REQUIRED) else None
            acc_depth_im = acc_depth_im.permute(2, 0, 1)
            if self.is_type_required(bits, self._EXP_DEPTH_INVERTED_REQUIRED):
                exp_depth_inverted_im = torch.where(exp_depth_im > 0, 1. / exp_depth_im, exp_depth_im.detach().max())

        inverse_depth_im = None
        if self.is_type_required(bits, self._INVERSE_DEPTH_REQUIRED):
            inverse_depth = 1. / (depths.clamp_min(0.) + 1e-8).unsqueeze(-1)
            inverse_depth_im = rasterize(inverse_depth, zero1).permute(2, 0, 1)

        hard_opac = None
        hard_depth_im = hard_inverse_depth_im = None
        if self.is_type_required(bits, self._HARD_DEPTH_REQUIRED | self._HARD_INVERSE_DEPTH_REQUIRED):
            hard_opac = opacities + (1 - opacities.detach())
        if self.is_type_required(bits, self._HARD_DEPTH_REQUIRED):
            hard_depth_im = rasterize(depths.unsqueeze(-1), zero1, opac=hard_opac).permute(2, 0, 1)
        if self.is_type_required(bits, self._HARD_INVERSE_DEPTH_REQUIRED):
            inverse_depth = 1. / (depths.clamp_min(0.) + 1e-8).unsqueeze(-1)
            hard_inverse_depth_im = rasterize(inverse_depth, zero1, opac=hard_opac).permute(2, 0, 1)

        return {
            "render": rgb, "alpha": alpha, "acc_depth": acc_depth_im, "acc_depth_inverted": acc_depth_inverted_im,
            "exp_depth": exp_depth_im, "exp_depth_inverted": exp_depth_inverted_im, "inverse_depth": inverse_depth_im,
            "hard_depth": hard_depth_im, "hard_inverse_depth": hard_inverse_depth_im,
            "viewspace_points": xys,
            "viewspace_points_grad_scale": viewspace_grad_scale(W, H, xys),
            "visibility_filter": visible,
            "radii": radii,
        }

    # ---- static helpers reused by other reference renderers (gsplat_renderer.py:203-378) ----------
    @staticmethod
    def render(means3D, opacities, scales, rotations, features, active_sh_degree: int, viewpoint_camera, bg_color,
               scaling_modifier=1.0, anti_aliased: bool = DEFAULT_ANTI_ALIASED_STATUS, colors_precomp=None, color_computer=None,
               block_size: int = DEFAULT_BLOCK_SIZE, extra_projection_kwargs: dict = None):
        W, H = camera_hw(viewpoint_camera)
        xys, depths, radii, conics, comp, num_tiles_hit, cov3d = _project(
            means3D, scales, rotations, viewpoint_camera, scaling_modifier, block_size, W, H, **(extra_projection_kwargs or {}))
        if colors_precomp is not None:
            rgbs = colors_precomp
        elif color_computer is not None:
            rgbs = color_computer(locals())
        else:
            rgbs = ops.sh_view_colors(active_sh_degree, means3D, viewpoint_camera.camera_center, features, None, radii > 0)
        if anti_aliased is True:
            opacities = opacities * comp[:, None]
        rgb = ops.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, rgbs, opacities, img_height=H, img_width=W,
                                      block_width=implementation_tile_size(block_size), background=bg_color, return_alpha=False)
        return {"render": rgb.permute(2, 0, 1), "viewspace_points": xys,
                "viewspace_points_grad_scale": viewspace_grad_scale(W, H, xys),
                "visibility_filter": radii > 0, "radii": radii}

    @staticmethod
    def project(means3D, scales, rotations, viewpoint_camera, scaling_modifier=1.0, block_size: int = DEFAULT_BLOCK_SIZE,
                extra_projection_kwargs: dict = None):
        W, H = camera_hw(viewpoint_camera)
        return _project(means3D, scales, rotations, viewpoint_camera, scaling_modifier, block_size, W, H, **(extra_projection_kwargs or {}))

    @staticmethod
    def rasterize_simplified(project_results, viewpoint_camera, colors, bg_color, opacities, anti_aliased: bool = True):
        xys, depths, radii, conics, comp, num_tiles_hit, cov3d = project_results
        W, H = camera_hw(viewpoint_camera)
        if anti_aliased is True:
            opacities = opacities * comp[:, None]
        return ops.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacities, img_height=H, img_width=W,
                                       block_width=DEFAULT_BLOCK_SIZE, background=bg_color, return_alpha=False).permute(2, 0, 1)

    @staticmethod
    def rasterize(opacities, rgbs, bg_color, project_results: Tuple, viewpoint_camera, xys_retain_grad: bool = True,
                  block_size: int = DEFAULT_BLOCK_SIZE, anti_aliased: bool = DEFAULT_ANTI_ALIASED_STATUS):
        W, H = camera_hw(viewpoint_camera)
        xys, depths, radii, conics, comp, num_tiles_hit, cov3d = project_results
        if xys_retain_grad is True:
            try:
                xys.retain_grad()
            except Exception:
                pass
        if anti_aliased is True:
            opacities = opacities * comp[:, None]
        rgb = ops.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, rgbs, opacities, img_height=H, img_width=W,
                                      block_width=implementation_tile_size(block_size), background=bg_color, return_alpha=False)
        return {"render": rgb.permute(2, 0, 1), "viewspace_points": xys,
                "viewspace_points_grad_scale": viewspace_grad_scale(W, H, xys),
                "visibility_filter": radii > 0, "radii": radii}

    def get_available_outputs(self) -> Dict:
        g = RendererOutputTypes.GRAY
        return {
            "rgb": RendererOutputInfo("render"),
            "alpha": RendererOutputInfo("alpha", type=g),
            "acc_depth": RendererOutputInfo("acc_depth", type=g),
            "acc_depth_inverted": RendererOutputInfo("acc_depth_inverted", type=g),
            "exp_depth": RendererOutputInfo("exp_depth", type=g),
            "exp_depth_inverted": RendererOutputInfo("exp_depth_inverted", type=g),
            "inverse_depth": RendererOutputInfo("inverse_depth", type=g),
            "hard_depth": RendererOutputInfo("hard_depth", type=g),
            "hard_inverse_depth": RendererOutputInfo("hard_inverse_depth", type=g),
        }
