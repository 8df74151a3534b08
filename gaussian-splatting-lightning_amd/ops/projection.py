"""Projection (`fully_fused_projection`, `project_gaussians`) and SH colours (`spherical_harmonics*`, `sh_view_colors*`)."""
from __future__ import annotations

import os
from typing import NamedTuple, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .. import _lib as L
from ._state import STATE as S
from ._common import (_SUPPORTED_D, _packed_row_stride, _guarded, _f32c, _rows, _raw_ptr, _grad_or_zeros, _side_stream, colour_stream,
                      join_pending_updates, _await_updates, _take_event)

# =============================================================================================
# projection
# =============================================================================================
class _ProjectFn(torch.autograd.Function):
    @staticmethod
    @_guarded(1)
    def forward(ctx, means, scales, quats, viewmats, Ks, width, height, tile_size, scale_modifier,
                eps2d, near_plane, far_plane, radius_clip, calc_compensations, want_tiles, camera_model=0, want_cov3d=False):
        lib = L.lib()
        means, scales, quats, viewmats, Ks = map(_f32c, (means, scales, quats, viewmats, Ks))
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        radii = torch.empty((C, N), dtype=torch.int32, device=dev)
        means2d = torch.empty((C, N, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((C, N), dtype=torch.float32, device=dev)
        conics = torch.empty((C, N, 3), dtype=torch.float32, device=dev)
        comps = torch.empty((C, N), dtype=torch.float32, device=dev) if calc_compensations else None
        tiles = torch.empty((C, N), dtype=torch.int32, device=dev) if want_tiles else None
        cov3d = torch.empty((C, N, 6), dtype=torch.float32, device=dev) if want_cov3d else None
        L.call("gspl_project_fwd", 
            C, N, L.ptr(means), L.ptr(scales), L.ptr(quats), L.ptr(viewmats), L.ptr(Ks),
            int(width), int(height), int(tile_size), float(scale_modifier), float(eps2d), float(near_plane),
            float(far_plane), float(radius_clip), int(camera_model),
            L.ptr(radii), L.ptr(means2d), L.ptr(depths), L.ptr(conics), L.ptr(comps), L.ptr(tiles), L.ptr(cov3d), L.stream())
        ctx.save_for_backward(means, scales, quats, viewmats, Ks, radii)
        ctx.cfg = (int(width), int(height), float(scale_modifier), float(eps2d), bool(calc_compensations), int(camera_model))
        ctx.set_materialize_grads(False)      # unused outputs (radii, tiles, often depths) arrive as None, not as zero tensors
        ctx.mark_non_differentiable(radii)
        outs = [radii, means2d, depths, conics]
        outs.append(comps if comps is not None else torch.empty(0, device=dev))
        if tiles is not None:
            ctx.mark_non_differentiable(tiles)
        outs.append(tiles if tiles is not None else torch.empty(0, dtype=torch.int32, device=dev))
        if cov3d is not None:
            ctx.mark_non_differentiable(cov3d)
        outs.append(cov3d if cov3d is not None else torch.empty(0, device=dev))
        return tuple(outs)

    @staticmethod
    @_guarded(0)
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps, _v_tiles, _v_cov3d=None):
        lib = L.lib()
        means, scales, quats, viewmats, Ks, radii = ctx.saved_tensors
        width, height, scale_modifier, eps2d, calc_comp, camera_model = ctx.cfg
        C, N = radii.shape
        dev = means.device
        s2 = s3 = 0
        if C == 1 and v_means2d is not None and v_conics is not None:
            # columns of the compositing backward's packed rows are read in place
            v_means2d, s2 = _rows(v_means2d, 2)
            v_conics, s3 = _rows(v_conics, 3)
        else:
            v_means2d = _grad_or_zeros(v_means2d, (C, N, 2), dev)
            v_conics = _grad_or_zeros(v_conics, (C, N, 3), dev)
        v_depths = _f32c(v_depths) if v_depths is not None else None
        v_comps = (_f32c(v_comps) if v_comps is not None and v_comps.numel() else None) if calc_comp else None
        alloc = torch.empty if C == 1 else torch.zeros
        v_means = alloc((N, 3), dtype=torch.float32, device=dev)
        v_scales = alloc((N, 3), dtype=torch.float32, device=dev)
        v_quats = alloc((N, 4), dtype=torch.float32, device=dev)
        L.call("gspl_project_bwd", 
            C, N, L.ptr(means), L.ptr(scales), L.ptr(quats), L.ptr(viewmats), L.ptr(Ks),
            width, height, scale_modifier, eps2d, camera_model, L.ptr(radii),
            _raw_ptr(v_means2d), s2, L.ptr(v_depths), _raw_ptr(v_conics), s3, L.ptr(v_comps),
            L.ptr(v_means), L.ptr(v_scales), L.ptr(v_quats), L.stream())
        return (v_means, v_scales, v_quats) + (None,) * 14


def fully_fused_projection(
        means: Tensor, covars: Optional[Tensor], quats: Tensor, scales: Tensor, viewmats: Tensor, Ks: Tensor,
        width: int, height: int, eps2d: float = 0.3, near_plane: float = 0.01, far_plane: float = 1e10,
        radius_clip: float = 0.0, packed: bool = False, sparse_grad: bool = False,
        calc_compensations: bool = False, camera_model: str = "pinhole", tile_size: int = 16,
        scale_modifier: float = 1.0):
    """gsplat-v1 signature (reference call: gsplat_v1_renderer.py:408-421).
    means [N,3], quats [N,4] (wxyz), scales [N,3], viewmats [C,4,4] (world->camera, NOT transposed),
    Ks [C,3,3].  Returns (radii [C,N] i32, means2d [C,N,2], depths [C,N], conics [C,N,3],
    compensations [C,N] | None)."""
    if covars is not None:
        raise NotImplementedError("covars input is not part of the reference's call sites")
    if packed:
        raise NotImplementedError("packed=True is not used by the reference (always packed=False)")
    if camera_model not in L.CAMERA_MODELS:
        raise ValueError(f"camera_model={camera_model!r}: one of {sorted(L.CAMERA_MODELS)}")
    assert means.dim() == 2 and means.shape[1] == 3, means.shape
    assert quats.shape == (means.shape[0], 4) and scales.shape == (means.shape[0], 3)
    assert viewmats.dim() == 3 and viewmats.shape[1:] == (4, 4) and Ks.shape == (viewmats.shape[0], 3, 3)
    radii, means2d, depths, conics, comps, _, _ = _ProjectFn.apply(
        means, scales, quats, viewmats, Ks, width, height, tile_size, scale_modifier, eps2d, near_plane, far_plane,
        radius_clip, calc_compensations, False, L.CAMERA_MODELS[camera_model])
    return radii, means2d, depths, conics, (comps if calc_compensations else None)




def _const_row(dev):
    """[0, 0, 0, 1] on `dev` (last row of a world->camera matrix), built once per device."""
    k = ("row", dev)
    if k not in S.consts:
        S.consts[k] = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
    return S.consts[k]


def _const_scalars(dev):
    k = ("01", dev)
    if k not in S.consts:
        S.consts[k] = (torch.zeros((), dtype=torch.float32, device=dev), torch.ones((), dtype=torch.float32, device=dev))
    return S.consts[k]


def _cached_intrinsics(fx: float, fy: float, cx: float, cy: float, dev):
    """K [3,3] on `dev` for python-float intrinsics; a training run cycles through a fixed camera set, so the
    host->device copy happens once per camera instead of once per step."""
    k = ("K", fx, fy, cx, cy, dev)
    K = S.consts.get(k)
    if K is None:
        if len(S.consts) > 4096:
            S.consts.clear()
        K = S.consts[k] = torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
    return K


_COV3D_FULL = (0, 1, 2, 1, 3, 4, 2, 4, 5)       # upper triangle (xx xy xz yy yz zz) -> row-major 3x3


def project_gaussians(
        means3d: Tensor, scales: Tensor, glob_scale: float, quats: Tensor, viewmat: Tensor,
        fx, fy, cx, cy, img_height: int, img_width: int, block_width: int,
        clip_thresh: float = 0.01, filter_2d_kernel_size: float = 0.3, return_cov3d: bool = True):
    """gsplat-v0 signature (reference call: gsplat_renderer.py:64-79).  viewmat [3|4, 4] world->camera.
    Returns (xys [N,2], depths [N], radii [N] i32, conics [N,3], compensation [N], num_tiles_hit [N] i32, cov3d [N,3,3]).
    cov3d = (R S)(R S)^T with zeros for culled Gaussians, as the in-tree Python returns it (gaussian_projection.py:47,137); written
    by the projection kernel (24 B per Gaussian), detached — no caller in the reference differentiates it.  `return_cov3d=False`
    (what the renderers pass: they drop it, gsplat_renderer.py:64) skips the output and returns None in its place."""
    dev = means3d.device
    N = means3d.shape[0]
    viewmat = viewmat.to(torch.float32)
    vm = viewmat if viewmat.shape[0] == 4 else torch.cat([viewmat, _const_row(dev)], dim=0)
    if isinstance(fx, Tensor):
        z, one = _const_scalars(dev)
        K = torch.stack([fx.reshape(()).float(), z, cx.reshape(()).float(), z, fy.reshape(()).float(), cy.reshape(()).float(),
                         z, z, one]).view(3, 3)
    else:
        K = _cached_intrinsics(float(fx), float(fy), float(cx), float(cy), dev)
    radii, xys, depths, conics, comps, tiles, cov6 = _ProjectFn.apply(
        means3d, scales, quats, vm[None], K[None], img_width, img_height, block_width, glob_scale,
        filter_2d_kernel_size, clip_thresh, 1e10, 0.0, True, True, L.GSPL_CAMERA_PINHOLE, bool(return_cov3d))
    cov3d = cov6.view(N, 6)[:, _COV3D_FULL].view(N, 3, 3) if return_cov3d else None
    # views, not selects: their backward is a view of the incoming gradient (select_backward allocates zeros + copies)
    return xys.view(N, 2), depths.view(N), radii.view(N), conics.view(N, 3), comps.view(N), tiles.view(N), cov3d


# =============================================================================================
# spherical harmonics
# =============================================================================================
class _SHFn(torch.autograd.Function):
    @staticmethod
    @_guarded(2)
    def forward(ctx, degree, dirs, origin, dc, rest, masks, flags):
        """dc: [N,K,3] merged (rest is None) or [N,1,3]; rest: [N,K-1,3] or None."""
        lib = L.lib()
        dirs, dc, rest = _f32c(dirs), _f32c(dc), _f32c(rest)
        origin = _f32c(origin)
        N = dirs.shape[0]
        dev = dirs.device
        merged = rest is None
        if merged:
            K = dc.shape[1]
            dc_stride = rest_stride = 3 * K
            rest_ptr = L.ptr(dc, offset_bytes=12) if K > 1 else None
            n_coeffs = K
        else:
            assert dc.shape[1] == 1
            dc_stride, rest_stride = 3, 3 * rest.shape[1]
            rest_ptr = L.ptr(rest) if rest.shape[1] > 0 else None
            n_coeffs = 1 + rest.shape[1]
        if (degree + 1) ** 2 > n_coeffs:
            raise ValueError(f"degree {degree} needs {(degree + 1) ** 2} coefficients, got {n_coeffs}")
        mask8 = None
        if masks is not None:
            # a bool mask is reinterpreted, not converted (a conversion is one more launch per frame)
            mask8 = masks.contiguous().view(torch.uint8) if masks.dtype == torch.bool else masks.to(torch.uint8).contiguous()
        colors = torch.empty((N, 3), dtype=torch.float32, device=dev)
        clamped = torch.empty((N, 3), dtype=torch.uint8, device=dev) if (flags & L.GSPL_SH_ADD_HALF_CLAMP) else None
        _await_updates(dc, rest)
        L.call("gspl_sh_fwd", N, int(degree), L.ptr(dirs), L.ptr(origin), L.ptr(dc), dc_stride, rest_ptr, rest_stride,
                                L.ptr(mask8), int(flags), L.ptr(colors), L.ptr(clamped), L.stream())
        ctx.save_for_backward(dirs, origin, dc, rest, mask8, clamped)
        ctx.cfg = (int(degree), int(flags), merged, n_coeffs, dc_stride, rest_stride)
        return colors

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_colors):
        lib = L.lib()
        dirs, origin, dc, rest, mask8, clamped = ctx.saved_tensors
        degree, flags, merged, n_coeffs, dc_stride, rest_stride = ctx.cfg
        N = dirs.shape[0]
        dev = dirs.device
        v_colors, vcs = _rows(v_colors, 3)
        need_dirs = ctx.needs_input_grad[1]
        v_dirs = torch.empty((N, 3), dtype=torch.float32, device=dev) if need_dirs else None
        v_dc = torch.empty_like(dc)
        if merged:
            v_rest = None
            v_rest_ptr = L.ptr(v_dc, offset_bytes=12) if n_coeffs > 1 else None
            rest_ptr = L.ptr(dc, offset_bytes=12) if n_coeffs > 1 else None
        else:
            v_rest = torch.empty_like(rest)
            v_rest_ptr = L.ptr(v_rest) if n_coeffs > 1 else None
            rest_ptr = L.ptr(rest) if n_coeffs > 1 else None
        L.call("gspl_sh_bwd", N, degree, n_coeffs, L.ptr(dirs), L.ptr(origin), L.ptr(dc), dc_stride, rest_ptr, rest_stride,
                                L.ptr(mask8), flags, L.ptr(clamped), _raw_ptr(v_colors), vcs,
                                L.ptr(v_dc), v_rest_ptr, L.ptr(v_dirs), L.stream())
        return None, v_dirs, None, v_dc, v_rest, None, None


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """gsplat signature (reference call: gsplat_renderer.py:105, gsplat_v1_renderer.py:124).
    dirs [N,3] (need not be unit), coeffs [N,K,3], masks [N] bool -> colours [N,3] (no +0.5)."""
    assert dirs.shape[-1] == 3 and coeffs.dim() == 3 and coeffs.shape[-1] == 3 and coeffs.shape[0] == dirs.shape[0]
    return _SHFn.apply(degrees_to_use, dirs, None, coeffs, None, masks, 0)


def spherical_harmonics_decomposed(degrees_to_use: int, dirs: Tensor, dc: Tensor, coeffs: Tensor,
                                   masks: Optional[Tensor] = None) -> Tensor:
    """yzslab-fork signature (reference call: gsplat_v1_renderer.py:124-130): dc [N,1,3], coeffs [N,K-1,3]."""
    return _SHFn.apply(degrees_to_use, dirs, None, dc, coeffs, masks, 0)


def sh_view_colors(degree: int, means: Tensor, camera_center: Tensor, dc: Tensor, rest: Optional[Tensor],
                   masks: Optional[Tensor] = None, detach_means: bool = True) -> Tensor:
    """Fused `clamp(SH(means - camera_center) + 0.5, min=0)` (gsplat_renderer.py:104-106) in one kernel:
    no viewdirs tensor, no separate clamp pass.  dc [N,1,3] + rest [N,K-1,3], or dc = merged [N,K,3] with rest None."""
    m = means.detach() if detach_means else means
    return _SHFn.apply(degree, m, camera_center, dc, rest, masks, L.GSPL_SH_ADD_HALF_CLAMP)


class _SHBatchedFn(torch.autograd.Function):
    """clamp(SH(means - origins[c]) + 0.5, 0) for C cameras in one launch (`gspl_sh_fwd_batched`): the coefficient rows are
    read once for all cameras; the backward sums the coefficient gradients over the cameras and writes them once."""

    @staticmethod
    def forward(ctx, degree, means, origins, dc, rest, radii):
        means, origins, dc, rest = _f32c(means), _f32c(origins), _f32c(dc), _f32c(rest)
        C, N = origins.shape[0], means.shape[0]
        dev = means.device
        merged = rest is None
        if merged:
            K = dc.shape[1]
            dc_stride = rest_stride = 3 * K
            rest_ptr = L.ptr(dc, offset_bytes=12) if K > 1 else None
            n_coeffs = K
        else:
            assert dc.shape[1] == 1
            dc_stride, rest_stride = 3, 3 * rest.shape[1]
            rest_ptr = L.ptr(rest) if rest.shape[1] > 0 else None
            n_coeffs = 1 + rest.shape[1]
        if (degree + 1) ** 2 > n_coeffs:
            raise ValueError(f"degree {degree} needs {(degree + 1) ** 2} coefficients, got {n_coeffs}")
        radii = None if radii is None else radii.to(torch.int32).contiguous()
        colors = torch.empty((C, N, 3), dtype=torch.float32, device=dev)
        clamped = torch.empty((C, N, 3), dtype=torch.uint8, device=dev)
        if N > 0:
            with torch.cuda.device(dev):
                _await_updates(dc, rest)
                L.call("gspl_sh_fwd_batched", C, N, int(degree), L.ptr(means), L.ptr(origins), L.ptr(dc), dc_stride, rest_ptr, rest_stride,
                       L.ptr(radii), L.GSPL_SH_ADD_HALF_CLAMP, L.ptr(colors), L.ptr(clamped), L.stream())
        ctx.save_for_backward(means, origins, dc, rest, radii, clamped)
        ctx.cfg = (int(degree), merged, n_coeffs, dc_stride, rest_stride)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        means, origins, dc, rest, radii, clamped = ctx.saved_tensors
        degree, merged, n_coeffs, dc_stride, rest_stride = ctx.cfg
        C, N = origins.shape[0], means.shape[0]
        v_colors = _f32c(v_colors)
        v_dc = torch.empty_like(dc)
        if merged:
            v_rest = None
            v_rest_ptr = L.ptr(v_dc, offset_bytes=12) if n_coeffs > 1 else None
        else:
            v_rest = torch.empty_like(rest)
            v_rest_ptr = L.ptr(v_rest) if n_coeffs > 1 else None
        if N > 0:
            with torch.cuda.device(means.device):
                L.call("gspl_sh_bwd_batched", C, N, degree, n_coeffs, L.ptr(means), L.ptr(origins), dc_stride, rest_stride,
                       L.ptr(radii), L.GSPL_SH_ADD_HALF_CLAMP, L.ptr(clamped), L.ptr(v_colors), L.ptr(v_dc), v_rest_ptr, L.stream())
        return None, None, None, v_dc, v_rest, None


def sh_view_colors_batched(degree: int, means: Tensor, camera_centers: Tensor, dc: Tensor, rest: Optional[Tensor],
                           radii: Optional[Tensor] = None) -> Tensor:
    """`sh_view_colors` for C cameras at once: camera_centers [C,3], radii [C,N] (rows with radius <= 0 are skipped)
    -> colours [C,N,3].  Means are detached (as gsplat_distributed_renderer.py:417 does)."""
    return _SHBatchedFn.apply(degree, means.detach(), camera_centers, dc, rest, radii)
