"""HipGSplatV1Renderer — drop-in for the reference's staged gsplat-v1 renderer
(internal/renderers/gsplat_v1_renderer.py:23-603; configs/gsplat_v1*.yaml): `GSplatV1Renderer` config
dataclass, `GSplatV1RendererModule`, and the static `GSplatV1` helper class that the reference's
distributed renderer and several research renderers call (`preprocess_camera`, `project`, `isect_encode`,
`rasterize`).  All native calls go to the HIP ops.

`runtime_options.camera_model` ("pinhole" | "ortho" | "fisheye") is forwarded to the projection (csrc/projection.hip).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Tuple

import torch

from .. import ops
from .renderer import Renderer, RendererConfig, RendererOutputInfo, RendererOutputTypes, camera_hw, implementation_tile_size


@dataclass
class HipGSplatV1Renderer(RendererConfig):
    block_size: int = 16
    anti_aliased: bool = True
    filter_2d_kernel_size: float = 0.3
    separate_sh: bool = True
    """Read shs_dc / shs_rest in place (no torch.cat of the features each step)."""
    tile_based_culling: bool = False
    max_viewspace_grad_scale: float = 65535.

    def instantiate(self, *args, **kwargs) -> "HipGSplatV1RendererModule":
        return HipGSplatV1RendererModule(self)


@dataclass
class RuntimeOptions:
    radius_clip: float = 0.
    camera_model: str = "pinhole"


def build_rotation_col2(q: torch.Tensor) -> torch.Tensor:
    """Third column of the rotation matrix of (w,x,y,z) quaternions (the splat normal the reference takes from
    `build_rotation(...)[:, :3, -1]`, gsplat_v1_renderer.py:247)."""
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], dim=-1)


class HipGSplatV1RendererModule(Renderer):
    _RGB_REQUIRED = 1
    _ALPHA_REQUIRED = 1 << 1
    _ACC_DEPTH_REQUIRED = 1 << 2
    _ACC_DEPTH_INVERTED_REQUIRED = 1 << 3
    _EXP_DEPTH_REQUIRED = 1 << 4
    _EXP_DEPTH_INVERTED_REQUIRED = 1 << 5
    _INVERSE_DEPTH_REQUIRED = 1 << 6
    _HARD_DEPTH_REQUIRED = 1 << 7
    _HARD_INVERSE_DEPTH_REQUIRED = 1 << 8
    _DEPTH_ALTERNATIVE = 1 << 9
    _NORMAL_REQUIRED = 1 << 10

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
        "inv_depth_alt": _DEPTH_ALTERNATIVE,
        "normal": _NORMAL_REQUIRED,
    }

    def __init__(self, config: HipGSplatV1Renderer):
        super().__init__()
        self.config = config
        self.runtime_options = RuntimeOptions()
        # the module consumes only (flatten_ids, isect_offsets): the list-only binning yields exactly the lists of
        # isect_tiles + isect_offset_encode without materialising and sorting 64-bit keys (2 + 4 radix passes instead of 6
        # over 12-byte pairs); the static GSplatV1.isect_encode keeps the keyed version for callers that want isect_ids
        self.isect_encode = GSplatV1.isect_encode_lists_only
        if self.config.tile_based_culling:                       # gsplat_v1_renderer.py:85-87
            self.isect_encode = GSplatV1.isect_encode_tile_based_culling
        self._inv_depth_alt_state = 0
        self._inv_depth_alt = [self.RENDER_TYPE_BITS["inverse_depth"], self.RENDER_TYPE_BITS["hard_inverse_depth"]]

    def parse_render_types(self, render_types: list) -> int:
        if render_types is None:
            return self._RGB_REQUIRED
        bits = 0
        for i in render_types:
            bits |= self.RENDER_TYPE_BITS[i]
        if self.is_type_required(bits, self._DEPTH_ALTERNATIVE):
            bits |= self._inv_depth_alt[self._inv_depth_alt_state]
            self._inv_depth_alt_state = int(not self._inv_depth_alt_state)
        return bits

    @staticmethod
    def is_type_required(bits: int, type: int) -> bool:
        return bits & type != 0

    def get_scales(self, camera, gaussian_model, **kwargs) -> Tuple[torch.Tensor, Any]:
        return gaussian_model.get_scales(), None

    def get_opacities(self, camera, gaussian_model, projections: Tuple, visibility_filter, status: Any, **kwargs):
        return gaussian_model.get_opacities().squeeze(-1), status

    def get_rgbs(self, camera, gaussian_model, projections: Tuple, visibility_filter, status: Any, **kwargs):
        pre_activated = getattr(gaussian_model, "is_pre_activated", False)
        if pre_activated or not self.config.separate_sh:
            ops.join_pending_updates(gaussian_model.get_xyz.device)      # `get_features` is a torch read of the SH parameters
            return ops.sh_view_colors(gaussian_model.active_sh_degree, gaussian_model.get_xyz, camera.camera_center,
                                      gaussian_model.get_features, None, visibility_filter)
        return ops.sh_view_colors(gaussian_model.active_sh_degree, gaussian_model.get_xyz, camera.camera_center,
                                  gaussian_model.get_shs_dc(), gaussian_model.get_shs_rest(), visibility_filter)

    def forward(self, viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, render_types: list = None, **kwargs):
        bits = self.parse_render_types(render_types)
        preprocessed_camera = GSplatV1.preprocess_camera(viewpoint_camera)

        scales, status = self.get_scales(viewpoint_camera, pc, **kwargs)
        if scaling_modifier != 1.:
            scales = scales * scaling_modifier
        projections = GSplatV1.project(
            preprocessed_camera, pc.get_means(), scales, pc.get_rotations(), eps2d=self.config.filter_2d_kernel_size,
            anti_aliased=self.config.anti_aliased, radius_clip=self.runtime_options.radius_clip,
            camera_model=self.runtime_options.camera_model)
        radii, means2d, depths, conics, compensations = projections
        radii_squeezed = radii.squeeze(0)
        visibility_filter = radii_squeezed > 0

        opacities, status = self.get_opacities(viewpoint_camera, pc, projections, visibility_filter, status, **kwargs)
        opacities = opacities.unsqueeze(0)
        if self.config.anti_aliased:
            opacities = opacities * compensations
        # lists-only binning of this package: the list length stays on the device until the first compositing launch is enqueued
        lazy = {"lazy": True} if getattr(self.isect_encode, "__func__", None) in (GSplatV1.isect_encode_lists_only.__func__,
                                                                                   GSplatV1.isect_encode_tile_based_culling.__func__) else {}
        isects = self.isect_encode(preprocessed_camera, projections, opacities, tile_size=implementation_tile_size(self.config.block_size), **lazy)

        means2d = means2d.squeeze(0)
        projection_for_rasterization = radii, means2d, depths, conics, compensations
        zero1 = torch.zeros((1,), dtype=torch.float, device=bg_color.device)

        def rasterize(input_features, background, return_alpha=False, opac=opacities, absgrad=True, channels_first=False, track_hits=False):
            c, a = GSplatV1.rasterize(preprocessed_camera, projection_for_rasterization, isects, opacities=opac,
                                      colors=input_features, background=background, tile_size=implementation_tile_size(self.config.block_size),
                                      absgrad=absgrad, channels_first=channels_first, track_hits=track_hits)
            return (c, a.squeeze(0).squeeze(-1)) if return_alpha else c

        outputs = {
            "render": None, "alpha": None, "acc_depth": None, "acc_depth_inverted": None, "exp_depth": None,
            "exp_depth_inverted": None, "inverse_depth": None, "hard_depth": None, "hard_inverse_depth": None,
            "normal": None, "inv_depth_alt": None,
            "viewspace_points": means2d,
            "viewspace_points_grad_scale": 0.5 * torch.tensor([preprocessed_camera[-1]]).to(means2d).clamp_(max=self.config.max_viewspace_grad_scale),
            "visibility_filter": visibility_filter, "acc_vis": None, "radii": radii_squeezed, "scales": scales,
            "opacities": opacities[0], "projections": projections, "isects": isects, "camera": viewpoint_camera,
            "preprocessed_camera": preprocessed_camera,
        }

        feats, bgs, index, n = [], [], {}, 0
        if self.is_type_required(bits, self._RGB_REQUIRED):
            feats.append(self.get_rgbs(viewpoint_camera, pc, projections, visibility_filter, status, **kwargs))
            bgs.append(bg_color)
            index["render"] = (n, n + 3)
            n += 3
        if self.is_type_required(bits, self._ACC_DEPTH_REQUIRED):
            feats.append(depths[0].unsqueeze(-1))
            bgs.append(zero1)
            index["acc_depth"] = (n, n + 1)
            n += 1
        if self.is_type_required(bits, self._NORMAL_REQUIRED):
            normals = build_rotation_col2(pc.get_rotations())
            dirs = pc.get_means() - viewpoint_camera.camera_center
            flip = torch.where(torch.einsum("ij,ij->i", normals, dirs) > 0, -1., 1.)
            feats.append(normals * flip.unsqueeze(-1))
            bgs.append(torch.zeros((3,), device=bg_color.device))
            index["normal"] = (n, n + 3)
            n += 3

        if n > 0:
            f = feats[0] if len(feats) == 1 else torch.concat(feats, dim=-1)
            b = bgs[0] if len(bgs) == 1 else torch.concat(bgs, dim=-1)
            # [D,H,W] straight from the kernel (the reference permutes an [H,W,D] image: every consumer of "render" would then
            # copy it to make it contiguous, forward and backward); the per-type outputs are contiguous channel slices
            # track_hits: the fork's rasterizer sets `means2d.has_hit_any_pixels` in its forward (read below as `acc_vis`)
            render_features, render_alpha = rasterize(f, background=b, return_alpha=True, channels_first=True, track_hits=True)
            render_alpha = render_alpha.unsqueeze(0)
            for k, (s, e) in index.items():
                outputs[k] = render_features[s:e]
            outputs["alpha"] = render_alpha
            outputs["acc_vis"] = means2d.has_hit_any_pixels          # avoid overriding by hard depth (gsplat_v1_renderer.py:286-287)
            if self.is_type_required(bits, self._ACC_DEPTH_INVERTED_REQUIRED):
                d = outputs["acc_depth"]
                outputs["acc_depth_inverted"] = torch.where(d > 0, 1. / d, d.detach().max())
            if self.is_type_required(bits, self._EXP_DEPTH_REQUIRED):
                d = outputs["acc_depth"]
                exp_depth_im = torch.where(render_alpha > 0, d / render_alpha, d.detach().max())
                outputs["exp_depth"] = exp_depth_im
                if self.is_type_required(bits, self._EXP_DEPTH_INVERTED_REQUIRED):
                    outputs["exp_depth_inverted"] = torch.where(exp_depth_im > 0, 1. / exp_depth_im, exp_depth_im.detach().max())

        if self.is_type_required(bits, self._INVERSE_DEPTH_REQUIRED):
            inverse_depth = 1. / (depths[0].clamp_min(0.) + 1e-8).unsqueeze(-1)
            im = rasterize(inverse_depth, zero1).permute(2, 0, 1)
            outputs["inverse_depth"] = outputs["inv_depth_alt"] = im
        if self.is_type_required(bits, self._HARD_DEPTH_REQUIRED):
            outputs["hard_depth"] = rasterize(depths[0].unsqueeze(-1), zero1, opac=opacities + (1 - opacities.detach()),
                                              absgrad=False).permute(2, 0, 1)
        if self.is_type_required(bits, self._HARD_INVERSE_DEPTH_REQUIRED):
            inverse_depth = 1. / (depths[0].clamp_min(0.) + 1e-8).unsqueeze(-1)
            im = rasterize(inverse_depth, zero1, opac=opacities + (1 - opacities.detach()), absgrad=False).permute(2, 0, 1)
            outputs["hard_inverse_depth"] = outputs["inv_depth_alt"] = im
        outputs["isects"] = GSplatV1.settled_isects(isects)
        return outputs

    def setup_web_viewer_tabs(self, viewer, server, tabs):
        """The two run-time options of the reference's viewer tab (gsplat_v1_renderer.py:350-352, 615-661): a "Radius Clip" number
        and a "Camera Model" drop-down that write `runtime_options` and ask the viewer to render again.  `server` is the viewer's
        viser server (only its `gui.add_number` / `gui.add_dropdown` are used; viser itself is not imported here)."""
        options = self.runtime_options
        with tabs.add_tab("gsplat"):
            clip = server.gui.add_number(label="Radius Clip", initial_value=options.radius_clip, step=0.1, min=0., max=65535.)
            model = server.gui.add_dropdown(label="Camera Model", options=["pinhole", "ortho", "fisheye"], initial_value=options.camera_model)

        @clip.on_update
        def _(_):
            options.radius_clip = clip.value
            viewer.rerender_for_all_client()

        @model.on_update
        def _(_):
            options.camera_model = model.value
            viewer.rerender_for_all_client()

        self._viewer_options = (clip, model)

    def get_available_outputs(self):
        g = RendererOutputTypes.GRAY
        return {
            "rgb": RendererOutputInfo("render"), "alpha": RendererOutputInfo("alpha", type=g),
            "acc_depth": RendererOutputInfo("acc_depth", type=g),
            "acc_depth_inverted": RendererOutputInfo("acc_depth_inverted", type=g),
            "exp_depth": RendererOutputInfo("exp_depth", type=g),
            "exp_depth_inverted": RendererOutputInfo("exp_depth_inverted", type=g),
            "inverse_depth": RendererOutputInfo("inverse_depth", type=g),
            "hard_depth": RendererOutputInfo("hard_depth", type=g),
            "hard_inverse_depth": RendererOutputInfo("hard_inverse_depth", type=g),
            "normal": RendererOutputInfo("normal", type=RendererOutputTypes.NORMAL_MAP),
        }


class GSplatV1:
    """Static staged API with the reference's signatures (gsplat_v1_renderer.py:370-603)."""

    @classmethod
    def preprocess_camera(cls, viewpoint_camera):
        """(viewmats [1,4,4], Ks [1,3,3], (W, H)) — built on the device (no host sync) and kept on the camera object: dataset
        cameras persist across steps and the six small launches that assemble K cost more host time than the projection
        call.  The cache is keyed on identity and version counter of every source tensor (pose refinement, viewer edits)."""
        src = (viewpoint_camera.world_to_camera, viewpoint_camera.fx, viewpoint_camera.fy, viewpoint_camera.cx, viewpoint_camera.cy)
        tensors = all(isinstance(v, torch.Tensor) for v in src)
        if tensors and not any(v.requires_grad for v in src):
            key = tuple((id(v), v._version) for v in src)
            hit = getattr(viewpoint_camera, "_gspl_v1_camera", None)
            if (hit is not None and hit[0] == key and all(a is b for a, b in zip(hit[1], src))
                    and hit[2]._version == hit[4] and hit[3]._version == hit[5]):      # (a caller may have edited them in place)
                return hit[2], hit[3], camera_hw(viewpoint_camera)
        viewmats = viewpoint_camera.world_to_camera.T.unsqueeze(0)
        dev = viewmats.device
        Ks = torch.zeros((1, 3, 3), dtype=torch.float, device=dev)
        Ks[0, 0, 0], Ks[0, 1, 1], Ks[0, 0, 2], Ks[0, 1, 2], Ks[0, 2, 2] = \
            viewpoint_camera.fx, viewpoint_camera.fy, viewpoint_camera.cx, viewpoint_camera.cy, 1.0
        if tensors and not any(v.requires_grad for v in src):
            viewmats = viewmats.contiguous()
            try:
                viewpoint_camera._gspl_v1_camera = (key, src, viewmats, Ks, viewmats._version, Ks._version)
            except AttributeError:
                pass
        return viewmats, Ks, camera_hw(viewpoint_camera)

    @classmethod
    def project(cls, preprocessed_camera: Tuple, means3d, scales, quats, eps2d: float = 0.3, anti_aliased: bool = True, **kwargs):
        """-> (radii [1,N], means2d [1,N,2], depths [1,N], conics [1,N,3], compensations [1,N])"""
        return ops.fully_fused_projection(
            means3d, None, quats, scales, viewmats=preprocessed_camera[0], Ks=preprocessed_camera[1],
            width=preprocessed_camera[2][0], height=preprocessed_camera[2][1], eps2d=eps2d,
            calc_compensations=anti_aliased, packed=False, **kwargs)

    @classmethod
    def isect_encode(cls, preprocessed_camera: Tuple, projection_results, tile_size: int = 16):
        """-> (tiles_per_gauss [1,N], isect_ids [I], flatten_ids [I], isect_offsets [1,th,tw])"""
        img_width, img_height = preprocessed_camera[-1]
        radii, means2d, depths, _, _ = projection_results
        tile_width = math.ceil(int(img_width) / float(tile_size))
        tile_height = math.ceil(int(img_height) / float(tile_size))
        tiles_per_gauss, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height,
                                                                  packed=False, n_cameras=1)
        return tiles_per_gauss, isect_ids, flatten_ids, ops.isect_offset_encode(isect_ids, 1, tile_width, tile_height)

    @classmethod
    def isect_encode_with_unused_opacities(cls, preprocessed_camera: Tuple, projection_results, opacities, tile_size: int = 16):
        return cls.isect_encode(preprocessed_camera, projection_results, tile_size)

    @classmethod
    def isect_encode_lists_only(cls, preprocessed_camera: Tuple, projection_results, opacities, tile_size: int = 16, lazy: bool = False):
        """Same per-tile lists as `isect_encode` (bit-identical flatten_ids / isect_offsets, tests/test_hip_parity.py), without
        the 64-bit keys: -> (None, None, flatten_ids [I], isect_offsets [1,th,tw])."""
        img_width, img_height = preprocessed_camera[-1]
        radii, means2d, depths, _, _ = projection_results
        tile_width = math.ceil(int(img_width) / float(tile_size))
        tile_height = math.ceil(int(img_height) / float(tile_size))
        flatten_ids, offsets = ops.bin_gaussians(means2d.reshape(-1, 2), depths.reshape(-1), radii.reshape(-1), int(img_height), int(img_width), tile_size,
                                                 lazy=lazy)
        return None, None, flatten_ids, offsets.reshape(1, tile_height, tile_width)

    @classmethod
    def isect_encode_tile_based_culling(cls, preprocessed_camera: Tuple, projection_results, opacities, tile_size: int = 16, lazy: bool = False):
        """Tile-based culling (StopThePop; reference: gsplat_v1_renderer.py:477-522): a (tile, Gaussian) pair is listed only
        if the Gaussian can reach alpha >= 1/255 somewhere in the tile.  Here that is the list-only two-level binning with
        its exact ellipse-vs-tile test (`ops.bin_gaussians`), so the 64-bit keys are never materialised:
        -> (tiles_per_gauss = None, isect_ids = None, flatten_ids [I'], isect_offsets [1,th,tw]); the reference consumes only
        the last two (`rasterize`, gsplat_v1_renderer.py:588-601).  `opacities` [1,N]: the ones compositing will use.
        lazy=True (the plugins' own forward): `flatten_ids` may come back as an `ops.LazyLists` — lists whose length the host has not
        read yet, which `rasterize` accepts; `settled_isects` turns the tuple into tensors afterwards."""
        img_width, img_height = preprocessed_camera[-1]
        radii, means2d, depths, conics, _ = projection_results
        tile_width = math.ceil(int(img_width) / float(tile_size))
        tile_height = math.ceil(int(img_height) / float(tile_size))
        flatten_ids, offsets = ops.bin_gaussians(means2d.reshape(-1, 2), depths.reshape(-1), radii.reshape(-1), int(img_height), int(img_width),
                                                 tile_size, conics=conics.reshape(-1, 3), opacities=opacities.reshape(-1), lazy=lazy)
        return None, None, flatten_ids, offsets.reshape(1, tile_height, tile_width)

    @staticmethod
    def settled_isects(isects):
        """The `isects` tuple with tensors only: an `ops.LazyLists` in the flatten_ids slot is replaced by the exact-length tensor
        (by then the compositing launches it fed are enqueued, so the wait for the count costs nothing)."""
        if isinstance(isects[2], ops.LazyLists):
            flat, _ = isects[2].resolve()
            return isects[0], isects[1], flat, isects[3]
        return isects

    @classmethod
    def preprocess(cls, preprocessed_camera: Tuple, means3d, scales, quats, eps2d: float = 0.3, anti_aliased: bool = True,
                   tile_size: int = 16, tile_based_culling: bool = False, opacities: torch.Tensor = None):
        projections = cls.project(preprocessed_camera, means3d=means3d, scales=scales, quats=quats, eps2d=eps2d, anti_aliased=anti_aliased)
        opacities = opacities.unsqueeze(0).squeeze(-1)
        if anti_aliased:
            opacities = opacities * projections[-1]
        if tile_based_culling:
            isects = cls.isect_encode_tile_based_culling(preprocessed_camera, projections, opacities, tile_size=tile_size)
        else:
            isects = cls.isect_encode(preprocessed_camera, projections, tile_size=tile_size)
        radii, means2d, depths, conics, compensations = projections
        return (radii, means2d.squeeze(0), depths, conics, compensations), isects, opacities

    @classmethod
    def rasterize(cls, preprocessed_camera: Tuple, projections, isects, opacities, colors, background, tile_size: int = 16,
                  absgrad: bool = True, **kwargs):
        """projections' means2d must be [N,2]; opacities [1,N]; colors [N,D]; background [D]
        -> (colors [H,W,D], alphas [H,W,1])"""
        img_width, img_height = preprocessed_camera[-1]
        _, means2d, _, conics, _ = projections
        _, _, flatten_ids, isect_offsets = isects
        rendered_colors, rendered_alphas = ops.rasterize_to_pixels(
            means2d=means2d, conics=conics, colors=colors.unsqueeze(0), opacities=opacities,
            image_width=int(img_width), image_height=int(img_height), tile_size=tile_size,
            isect_offsets=isect_offsets, flatten_ids=flatten_ids, backgrounds=background.unsqueeze(0), absgrad=absgrad, **kwargs)
        return rendered_colors.squeeze(0), rendered_alphas.squeeze(0)

    @staticmethod
    def get_intrinsics_matrix(fx, fy, cx, cy, device):
        K = torch.eye(3, device=device)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
        return K
