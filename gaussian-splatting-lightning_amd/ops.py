"""
ops.py — torch.autograd.Function wrappers over the C-ABI, mirroring the operator interfaces the
reference's renderers call (same names, argument meaning and error behaviour):

  gsplat v1 (internal/renderers/gsplat_v1_renderer.py:8-20)
      fully_fused_projection, isect_tiles, isect_offset_encode, rasterize_to_pixels,
      spherical_harmonics, spherical_harmonics_decomposed
  gsplat v0 (internal/renderers/gsplat_renderer.py:2-4, pypreprocess_gsplat_renderer.py:1-2)
      project_gaussians, rasterize_gaussians
  Inria (internal/renderers/vanilla_renderer.py:14)
      GaussianRasterizationSettings, GaussianRasterizer

Host side only: shape checks, buffer allocation through torch's caching allocator, stream hand-off.
All arithmetic happens in libgspl_hip.so; nothing here falls back to PyTorch math.
"""
from __future__ import annotations

import os
from typing import NamedTuple, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib as L

_SUPPORTED_D = (1, 2, 3, 4, 8)
_PACKED_ROW_STRIDE = int(__import__("os").environ.get("GSPL_PACKED_STRIDE", "0"))


def _packed_row_stride(nv: int) -> int:
    """Floats per packed gradient row of the compositing backward."""
    return max(nv, _PACKED_ROW_STRIDE)


def _guarded(pos: int):
    """Decorator: run the function with the device of its `pos`-th positional argument (a tensor, or for a backward the
    autograd context whose first saved tensor decides) made current, so that kernels are enqueued on THAT device's current
    stream even when the caller's current device is another one (single-process multi-GPU, viewer / eval helpers)."""
    def deco(fn):
        import functools

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            a = args[pos]
            if not isinstance(a, Tensor):
                saved = getattr(a, "saved_tensors", None)
                a = next((t for t in (saved or ()) if isinstance(t, Tensor)), None)
            if a is None or not a.is_cuda:
                return fn(*args, **kwargs)
            with L.device_guard(a):
                return fn(*args, **kwargs)
        return wrapper
    return deco


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if PENDING_UPDATES and (t.dtype != torch.float32 or not t.is_contiguous()):
        join_pending_updates(t.device)      # the copy below is a torch read of what may be a parameter with an update in flight
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _rows(g: Optional[Tensor], width: int):
    """(tensor, row stride in floats) for a gradient whose rows may be columns of a wider packed buffer
    (what the compositing backward hands out): consumed in place when the layout allows, copied otherwise.
    Stride 0 means dense."""
    if g is None:
        return None, 0
    if g.dtype != torch.float32:
        g = g.float()
    if g.is_contiguous():
        return g, 0
    if g.shape[-1] == width and g.stride(-1) == 1:
        lead = [d for d in range(g.dim() - 1) if g.shape[d] != 1]
        if len(lead) == 1 and g.stride(lead[0]) >= width:
            return g, g.stride(lead[0])
    return g.contiguous(), 0


def _raw_ptr(t: Optional[Tensor]):
    """Device pointer of a possibly non-contiguous tensor's first element."""
    import ctypes
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _grad_or_zeros(g: Optional[Tensor], like_shape, device) -> Tensor:
    if g is None:
        return torch.zeros(like_shape, dtype=torch.float32, device=device)
    return _f32c(g)


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


_CONSTS: dict = {}


def _const_row(dev):
    """[0, 0, 0, 1] on `dev` (last row of a world->camera matrix), built once per device."""
    k = ("row", dev)
    if k not in _CONSTS:
        _CONSTS[k] = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
    return _CONSTS[k]


def _const_scalars(dev):
    k = ("01", dev)
    if k not in _CONSTS:
        _CONSTS[k] = (torch.zeros((), dtype=torch.float32, device=dev), torch.ones((), dtype=torch.float32, device=dev))
    return _CONSTS[k]


def _cached_intrinsics(fx: float, fy: float, cx: float, cy: float, dev):
    """K [3,3] on `dev` for python-float intrinsics; a training run cycles through a fixed camera set, so the
    host->device copy happens once per camera instead of once per step."""
    k = ("K", fx, fy, cx, cy, dev)
    K = _CONSTS.get(k)
    if K is None:
        if len(_CONSTS) > 4096:
            _CONSTS.clear()
        K = _CONSTS[k] = torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
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


# =============================================================================================
# tile binning
# =============================================================================================
@_guarded(1)
def _isect(mode: int, means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_w: int, tile_h: int):
    lib = L.lib()
    means2d, depths = _f32c(means2d.detach()), _f32c(depths.detach())
    radii = radii.to(torch.int32).contiguous()
    N = means2d.shape[0]
    dev = means2d.device
    tiles = torch.empty((N,), dtype=torch.int32, device=dev)
    cum = torch.empty((N,), dtype=torch.int64, device=dev)
    if N == 0:
        z64 = torch.empty((0,), dtype=torch.int64, device=dev)
        return tiles, z64, torch.empty((0,), dtype=torch.int32, device=dev)
    ws_bytes = lib.gspl_isect_workspace_bytes(N, 0)
    if ws_bytes == 0:
        raise RuntimeError("gspl_isect_workspace_bytes failed: " + lib.gspl_last_error().decode())
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    L.call("gspl_isect_count", N, mode, L.ptr(means2d), L.ptr(radii), tile_size, tile_w, tile_h,
                                 L.ptr(tiles), L.ptr(cum), L.ptr(ws), ws_bytes, L.stream())
    n_isects = int(cum[-1].item())        # the one host read-back of the pipeline (sizes the sort buffers)
    isect_ids = torch.empty((n_isects,), dtype=torch.int64, device=dev)
    flatten_ids = torch.empty((n_isects,), dtype=torch.int32, device=dev)
    if n_isects > 0:
        ws_bytes = lib.gspl_isect_workspace_bytes(N, n_isects)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        L.call("gspl_isect_emit_sort", N, mode, L.ptr(means2d), L.ptr(radii), L.ptr(depths), L.ptr(cum),
                                         tile_size, tile_w, tile_h, n_isects, L.ptr(isect_ids), L.ptr(flatten_ids),
                                         L.ptr(ws), ws_bytes, L.stream())
    return tiles, isect_ids, flatten_ids


def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int, tile_height: int,
                sort: bool = True, packed: bool = False, n_cameras: Optional[int] = None,
                camera_ids: Optional[Tensor] = None, gaussian_ids: Optional[Tensor] = None,
                mode: int = L.GSPL_MODE_GSPLAT):
    """gsplat-v1 signature (reference call: gsplat_v1_renderer.py:446-457).  Single camera:
    means2d [1,N,2] or [N,2], radii [1,N] or [N], depths likewise.
    Returns (tiles_per_gauss [1,N] i32, isect_ids [I] i64, flatten_ids [I] i32)."""
    if packed or camera_ids is not None or gaussian_ids is not None:
        raise NotImplementedError("packed mode is not used by the reference")
    if n_cameras not in (None, 1) or (means2d.dim() == 3 and means2d.shape[0] != 1):
        raise NotImplementedError("one camera per call (the reference always renders one camera per rank)")
    if not sort:
        raise NotImplementedError("sort=False is not used by the reference")
    tiles, ids, flat = _isect(mode, means2d.reshape(-1, 2), radii.reshape(-1), depths.reshape(-1), tile_size, tile_width, tile_height)
    return tiles[None], ids, flat


@_guarded(0)
def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """gsplat-v1 signature (gsplat_v1_renderer.py:458) -> offsets [n_cameras, tile_height, tile_width] i32."""
    if n_cameras != 1:
        raise NotImplementedError("one camera per call")
    lib = L.lib()
    offsets = torch.empty((1, tile_height, tile_width), dtype=torch.int32, device=isect_ids.device)
    isect_ids = isect_ids.contiguous()
    L.call("gspl_isect_offsets", isect_ids.shape[0], L.ptr(isect_ids) if isect_ids.numel() else None,
                                   tile_width, tile_height, L.ptr(offsets), L.stream())
    return offsets


# =============================================================================================
# compositing
# =============================================================================================
# When True, the compositing backward also reports which splats some pixel actually composited and attaches the mask as
# `has_hit_any_pixels` to the caller's screen-space tensor (the fork-only side channel gsplat's SelectiveAdam adapter
# reads, internal/optimizers.py:39).  Off by default: it is one more byte store per (tile, splat) in the hot kernel.
TRACK_HIT_PIXELS = False
# Staged binning (`bin_gaussians`): with a speculative emission in flight the tile sort is enqueued before the host has read the
# list length (False: wait for the count first, then sort — the round-1 order; kept for A/B runs and the tests of both orders).
DEVICE_SIDE_LIST_LENGTH = True
# Introspection for bench.py / tools: with KEEP_LAST_RASTER set, the last compositing forward leaves its per-splat inputs and
# tile lists in LAST_RASTER (a dict of tensors; nothing is copied).
KEEP_LAST_RASTER = False
LAST_RASTER: Optional[dict] = None


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    @_guarded(1)
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, offsets, flatten_ids,
                absgrad, mode, layout, track_hits=False):
        lib = L.lib()
        means2d_in = means2d
        means2d, conics, colors, opacities = map(_f32c, (means2d, conics, colors, opacities))
        backgrounds = _f32c(backgrounds)
        N, D = colors.shape
        assert D in _SUPPORTED_D
        dev = means2d.device
        tile_w, tile_h = (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size
        offsets = offsets.to(torch.int32).contiguous()
        assert offsets.numel() == tile_w * tile_h
        lazy = flatten_ids if isinstance(flatten_ids, LazyLists) else None
        if lazy is not None and lazy.settled:
            flatten_ids, lazy = lazy.flat, None
        if lazy is None:
            flatten_ids = flatten_ids.to(torch.int32).contiguous()
            n_isects = flatten_ids.shape[0]
        shape = (height, width, D) if layout == L.GSPL_LAYOUT_HWC else (D, height, width)
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        alphas = torch.empty((height, width), dtype=torch.float32, device=dev)
        final_Ts = torch.empty((height, width), dtype=torch.float32, device=dev)
        last_ids = torch.empty((height, width), dtype=torch.int32, device=dev)
        # `has_hit_any_pixels` of the fork's rasterizer (set in ITS forward; read as `acc_vis`, gsplat_v1_renderer.py:287, and by
        # SelectiveAdam, internal/optimizers.py:39): which splats some pixel actually composited
        hit = torch.zeros((N,), dtype=torch.uint8, device=dev) if track_hits else None
        with torch.cuda.device(dev):
            if lazy is not None:
                # lists whose length is still on its way to the host (LazyLists): composite on the capacity-sized buffer, the end of
                # the last list is read on the device; THEN look at the count, and repeat the launch if the guess had been too low
                L.call("gspl_composite_fwd",
                       N, -1, D, mode, layout, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opacities), L.ptr(backgrounds),
                       width, height, tile_size, tile_w, tile_h, L.ptr(lazy.offsets_ext), L.ptr(lazy.flat_cap),
                       L.ptr(out), L.ptr(alphas), L.ptr(final_Ts), L.ptr(last_ids), L.ptr(hit), L.stream())
                held = lazy.settle()
                flatten_ids, offsets = lazy.flat, lazy.offsets.to(torch.int32).contiguous()
                n_isects = flatten_ids.shape[0]
                if not held and hit is not None:
                    hit.zero_()
            if lazy is None or not held:
                L.call("gspl_composite_fwd",
                       N, n_isects, D, mode, layout, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opacities), L.ptr(backgrounds),
                       width, height, tile_size, tile_w, tile_h, L.ptr(offsets), L.ptr(flatten_ids) if n_isects else None,
                       L.ptr(out), L.ptr(alphas), L.ptr(final_Ts), L.ptr(last_ids), L.ptr(hit), L.stream())
        if hit is not None:
            means2d_in.has_hit_any_pixels = hit.view(torch.bool)
        ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, final_Ts, last_ids)
        if KEEP_LAST_RASTER:
            global LAST_RASTER
            LAST_RASTER = dict(mode=mode, width=width, height=height, means2d=means2d, conics=conics, opacities=opacities,
                               colors=colors, flatten_ids=flatten_ids, offsets=offsets)
        ctx.cfg = (width, height, tile_size, tile_w, tile_h, bool(absgrad), mode, layout)
        ctx.means2d_ref = means2d_in      # the caller's tensor object: `.absgrad` is attached to it in backward
        return out, alphas

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_out, v_alphas):
        lib = L.lib()
        means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, final_Ts, last_ids = ctx.saved_tensors
        width, height, tile_size, tile_w, tile_h, absgrad, mode, layout = ctx.cfg
        N, D = colors.shape
        dev = means2d.device
        n_isects = flatten_ids.shape[0]
        NV = 6 + D + (2 if absgrad else 0)
        RS = _packed_row_stride(NV)
        packed = torch.zeros((N, RS), dtype=torch.float32, device=dev)      # one memset, one row per splat
        if n_isects > 0 and N > 0:
            v_out = _grad_or_zeros(v_out, final_Ts.shape + (D,) if layout == L.GSPL_LAYOUT_HWC else (D,) + final_Ts.shape, dev)
            v_alphas = _f32c(v_alphas) if v_alphas is not None else None
            hit = torch.zeros((N,), dtype=torch.uint8, device=dev) if TRACK_HIT_PIXELS else None
            L.call("gspl_composite_bwd_packed",
                N, n_isects, D, mode, layout, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opacities), L.ptr(backgrounds),
                width, height, tile_size, tile_w, tile_h, L.ptr(offsets), L.ptr(flatten_ids), L.ptr(final_Ts), L.ptr(last_ids),
                L.ptr(v_out), L.ptr(v_alphas), L.ptr(packed), RS, 1 if absgrad else 0, L.ptr(hit), L.stream())
            if hit is not None:
                ctx.means2d_ref.has_hit_any_pixels = hit.bool()
        v_means2d, v_conics, v_opac, v_colors = packed[:, 0:2], packed[:, 2:5], packed[:, 5], packed[:, 6:6 + D]
        v_abs = packed[:, 6 + D:8 + D] if absgrad else None
        if absgrad:
            # same side channel as gsplat: the density controller reads `viewspace_points.absgrad`
            # (internal/density_controllers/vanilla_density_controller.py:112-113)
            ctx.means2d_ref.absgrad = v_abs
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[4]:
            T_final = final_Ts
            vo = v_out if layout == L.GSPL_LAYOUT_HWC else v_out.permute(1, 2, 0)
            v_bg = (vo * T_final[..., None]).sum(dim=(0, 1))
        return v_means2d, v_conics, v_colors, v_opac.reshape(opacities.shape), v_bg, None, None, None, None, None, None, None, None, None


def _composite(means2d, conics, colors, opacities, backgrounds, width, height, tile_size, offsets, flatten_ids,
               absgrad, mode, layout, track_hits=False):
    """Channel-count adapter: kernels are built for D in {1,2,3,4,8}; other widths are zero-padded /
    split into groups of 8 (extra channels composite to zero and carry zero gradient)."""
    D = colors.shape[1]
    if D in _SUPPORTED_D:
        return _CompositeFn.apply(means2d, conics, colors, opacities, backgrounds, width, height, tile_size, offsets,
                                  flatten_ids, absgrad, mode, layout, track_hits)
    outs, alphas = [], None
    for s in range(0, D, 8):
        e = min(D, s + 8)
        c = colors[:, s:e]
        bg = None if backgrounds is None else backgrounds[s:e]
        w = e - s
        pad = next(d for d in _SUPPORTED_D if d >= w) - w
        if pad:
            c = torch.nn.functional.pad(c, (0, pad))
            bg = None if bg is None else torch.nn.functional.pad(bg, (0, pad))
        o, alphas = _CompositeFn.apply(means2d, conics, c, opacities, bg, width, height, tile_size, offsets, flatten_ids,
                                       absgrad and s == 0, mode, layout, track_hits and s == 0)
        outs.append(o[..., :w] if layout == L.GSPL_LAYOUT_HWC else o[:w])
    return torch.cat(outs, dim=-1 if layout == L.GSPL_LAYOUT_HWC else 0), alphas


def rasterize_to_pixels(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor,
                        image_width: int, image_height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
                        backgrounds: Optional[Tensor] = None, masks: Optional[Tensor] = None, packed: bool = False,
                        absgrad: bool = False, channels_first: bool = False, track_hits: bool = False) -> Tuple[Tensor, Tensor]:
    """gsplat signature as the reference calls it (gsplat_v1_renderer.py:588-601): means2d [N,2] (or [1,N,2]),
    conics [1,N,3], colors [1,N,D], opacities [1,N], isect_offsets [1,th,tw], backgrounds [1,D].
    Returns (colors [1,H,W,D], alphas [1,H,W,1]).  With absgrad=True, backward sets `means2d.absgrad`.
    channels_first (extension): colors come out as [1,D,H,W] straight from the kernel (see `rasterize_gaussians`).
    track_hits: set `means2d.has_hit_any_pixels` ([N] bool: composited by some pixel) in the forward, as the fork's rasterizer does."""
    if packed or masks is not None:
        raise NotImplementedError("packed / masks are not used by the reference")
    m2 = means2d if means2d.dim() == 2 else means2d.squeeze(0)
    out, alphas = _composite(m2, conics.reshape(-1, 3), colors.reshape(-1, colors.shape[-1]), opacities.reshape(-1),
                             None if backgrounds is None else backgrounds.reshape(-1), image_width, image_height, tile_size,
                             isect_offsets.reshape(-1), flatten_ids, absgrad, L.GSPL_MODE_GSPLAT,
                             L.GSPL_LAYOUT_CHW if channels_first else L.GSPL_LAYOUT_HWC, track_hits)
    if absgrad and m2 is not means2d:
        raise ValueError("absgrad needs means2d given as [N,2] so that .absgrad lands on the caller's tensor")
    return out[None], alphas[None, ..., None]


@torch.no_grad()
def composite_scores(means2d: Tensor, conics: Tensor, opacities: Tensor, image_width: int, image_height: int, tile_size: int,
                     isect_offsets: Tensor, flatten_ids: Tensor, pixel_weights: Optional[Tensor] = None,
                     mode: int = L.GSPL_MODE_GSPLAT, with_dist: bool = False):
    """Per-splat sums over the pixels each splat contributes to (`gspl_composite_scores`): returns
    (count [N] i32, opacity_sum, alpha_sum, visibility_sum (= sum of blending weights alpha*T), weighted_sum
    (= sum of pixel_weights * alpha * T, None without pixel_weights), dist_sum (None unless with_dist)), all [N] f32."""
    if tile_size != 16:
        raise NotImplementedError("tile_size 16 only")
    m2 = _f32c(means2d.detach()).reshape(-1, 2)
    con = _f32c(conics.detach()).reshape(-1, 3)
    op = _f32c(opacities.detach()).reshape(-1)
    N, dev = m2.shape[0], m2.device
    offs = isect_offsets.reshape(-1).to(torch.int32).contiguous()
    flat = flatten_ids.to(torch.int32).contiguous()
    tile_w, tile_h = (image_width + 15) // 16, (image_height + 15) // 16
    assert offs.numel() == tile_w * tile_h
    count = torch.zeros((N,), dtype=torch.int32, device=dev)
    sums = torch.zeros((5, N), dtype=torch.float32, device=dev)
    pw = None
    if pixel_weights is not None:
        pw = _f32c(pixel_weights.detach()).reshape(image_height, image_width)
    n_isects = flat.shape[0]
    if N > 0 and n_isects > 0:
        with torch.cuda.device(dev):
            L.call("gspl_composite_scores", N, n_isects, mode, L.ptr(m2), L.ptr(con), L.ptr(op), image_width, image_height, 16, tile_w, tile_h,
                   L.ptr(offs), L.ptr(flat), L.ptr(pw), L.ptr(count), L.ptr(sums[0]), L.ptr(sums[1]), L.ptr(sums[2]),
                   L.ptr(sums[3]) if pw is not None else None, L.ptr(sums[4]) if with_dist else None, L.stream())
    return count, sums[0], sums[1], sums[2], (sums[3] if pw is not None else None), (sums[4] if with_dist else None)


def hit_pixel_count(xys: Tensor, depths: Tensor, radii: Tensor, conics: Tensor, num_tiles_hit: Tensor, opacities: Tensor,
                    img_height: int, img_width: int, block_width: int = 16):
    """Signature of the gsplat fork's `hit_pixel_count` as the reference calls it
    (internal/renderers/gsplat_hit_pixel_count_renderer.py:34-44): returns (count [N] i32, opacity_score, alpha_score,
    visibility_score [N] f32) of one view — the number of pixels a splat is composited into and the sums of its opacity,
    alpha and blending weight alpha*T over them (LightGaussian's importance terms; restated, parity unpinned)."""
    flat, offsets = bin_gaussians(xys, depths, radii, img_height, img_width, block_width, conics=conics, opacities=opacities)
    count, o_sum, a_sum, v_sum, _, _ = composite_scores(xys, conics, opacities, img_width, img_height, block_width, offsets, flat)
    return count, o_sum, a_sum, v_sum


def rasterize_to_weights(means2d: Tensor, conics: Tensor, opacities: Tensor, image_width: int, image_height: int, tile_size: int,
                         isect_offsets: Tensor, flatten_ids: Tensor, pixel_weights: Tensor):
    """Signature of the gsplat fork's `rasterize_to_weights` as the reference calls it
    (internal/density_controllers/taming_3dgs_density_controller.py:429-439): batched inputs ([1,N,..], pixel_weights
    [1,H,W]); returns (accum_weights, reverse_counts, blend_weights, dist_accum), each [1,N] f32: per splat, over the pixels
    it contributes to, the sum of pixel_weight * alpha * T, the number of pixels, the sum of alpha * T and the sum of the
    pixel-to-centre distances (Taming-3DGS score terms; restated from the paper's description, parity unpinned)."""
    count, _, _, v_sum, w_sum, d_sum = composite_scores(means2d, conics, opacities, image_width, image_height, tile_size, isect_offsets,
                                                        flatten_ids, pixel_weights=pixel_weights, with_dist=True)
    return w_sum[None], count.float()[None], v_sum[None], d_sum[None]


class _side_stream:
    """`with _side_stream(dev) as s:` runs the enclosed launches on a per-device side stream that first waits for everything
    already enqueued on the current stream; `s.join()` makes the current stream wait for them.  Set GSPL_SIDE_STREAM=0 to
    keep everything on the caller's stream."""
    _streams: dict = {}
    _handles: dict = {}
    _low: dict = {}
    _torch: dict = {}

    def __init__(self, dev):
        import os
        self.enabled = os.environ.get("GSPL_SIDE_STREAM", "1") != "0"
        self.dev = dev
        self.ctx = None
        if self.enabled:
            key = (dev.type, dev.index)
            s = _side_stream._streams.get(key)
            if s is None:
                s = _side_stream._streams[key] = torch.cuda.Stream(device=dev)
            self.stream = s

    def __enter__(self):
        if self.enabled:
            self.stream.wait_stream(torch.cuda.current_stream(self.dev))
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.enabled:
            torch.cuda.current_stream(self.dev).wait_stream(self.stream)


def colour_stream(dev):
    """(raw handle, torch stream) of the stream the fused rasterizer launches its colour (SH) kernel on, next to the binning on the
    caller's stream: the package's torch side stream (default priority), or with GSPL_SIDE_LOW_PRIORITY=1 the library's
    lowest-priority stream (torch cannot create one below the default); (None, None) with GSPL_SIDE_STREAM=0."""
    side = _side_stream(dev)
    if not side.enabled:
        return None, None
    hk = (dev.type, dev.index)
    raw = _side_stream._handles.get(hk)
    if raw is None:
        with torch.cuda.device(dev):
            raw = (L.lib().gspl_low_priority_stream() or 0) if SIDE_LOW_PRIORITY else 0
        low = bool(raw)
        if not raw:
            raw = side.stream.cuda_stream      # (~10 us of Python per look-up: cached)
        _side_stream._handles[hk] = raw
        _side_stream._low[hk] = low
        _side_stream._torch[hk] = torch.cuda.ExternalStream(raw, device=dev) if low else side.stream
    return raw, _side_stream._torch[hk]


# Parameter updates in flight on another stream (optimizers.FusedAdam(deferred=...): the Adam update of the SH coefficients runs on the
# colour stream, under the next frame's geometry / binning kernels): data_ptr -> (event recorded after the update, raw handle of the
# stream it was launched on, device).  Every kernel of this package that reads a parameter which may be deferred — the SH colour kernels —
# calls `_await_updates` on the stream it launches on; the optimizer retires its entries at its next step.
PENDING_UPDATES: dict = {}


def join_pending_updates(device=None):
    """Make the current stream of `device` (default: every device with an entry) wait for ALL parameter updates in flight
    (`FusedAdam(deferred=...)`).  `_await_updates` recognises a parameter by its data pointer, which covers the kernels of this
    package reading it in place; a TORCH read — `torch.cat` inside `get_features`, a dtype / layout copy, user code in
    `on_train_batch_end` — produces a new tensor that no pointer table can tie to the update, so every place of this package that
    reads a possibly-deferred parameter through torch calls this first (renderers/renderer.py: `model_sh_pair`, `_f32c` above)."""
    if not PENDING_UPDATES:
        return
    seen = set()
    for done, _raw, dev in list(PENDING_UPDATES.values()):
        if id(done) in seen:
            continue
        seen.add(id(done))
        if device is not None and torch.device(dev) != torch.device(device):
            continue
        torch.cuda.current_stream(dev).wait_event(done)


def _await_updates(*tensors, on_raw_stream=None):
    """Make the current stream wait for the in-flight updates of `tensors` (no-op for updates launched on `on_raw_stream`, which
    stream order already covers)."""
    if not PENDING_UPDATES:
        return
    for t in tensors:
        if t is None:
            continue
        ent = PENDING_UPDATES.get(t.data_ptr())
        if ent is not None and ent[1] != on_raw_stream:
            torch.cuda.current_stream(t.device).wait_event(ent[0])


class _PendingBins:
    """Binning in flight: the count/depth-sort half has been launched and the number of intersections is on its
    way to a pinned host word; `bin_gaussians_end` waits for it and launches the emit/sort half."""
    __slots__ = ("N", "mode", "means2d", "radii", "cull_c", "cull_o", "order", "cum", "spans", "offsets", "tile_w", "tile_h",
                 "block_width", "host_count", "event", "dev", "capacity", "ws2", "ws2_bytes", "big_list", "depths", "count", "offsets_buf")


# =============================================================================================
# visible-splat records of the Gaussian-sharded renderer (csrc/records.hip)
# =============================================================================================
def _batched(ts: Sequence[Tensor]) -> Tensor:
    """[C, ...] tensor of C per-camera tensors: the base buffer itself when they are its consecutive slices (what
    `batch_project` hands out), a stacked copy otherwise."""
    t0 = ts[0]
    base = t0._base
    if base is not None and base.is_contiguous() and base.dim() == t0.dim() + 1 and base.shape[0] == len(ts) and base.shape[1:] == t0.shape:
        step = t0.numel() * t0.element_size()
        if all(t._base is base and t.is_contiguous() and t.data_ptr() == base.data_ptr() + i * step for i, t in enumerate(ts)):
            return base
    return torch.stack([t.contiguous() for t in ts])


class _UnbindFn(torch.autograd.Function):
    """`t.unbind(0)` whose backward hands the batched gradient through when the per-slice gradients already ARE consecutive
    slices of one buffer (what `_PackRecordsFn.backward` returns) — `t[i]` costs a zero fill plus a copy per slice there."""

    @staticmethod
    def forward(ctx, t):
        ctx.shape, ctx.like = t.shape, (t.dtype, t.device)
        ctx.set_materialize_grads(False)
        return t.unbind(0)

    @staticmethod
    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return None
        if any(g is None for g in grads):
            dt, dev = ctx.like
            grads = [g if g is not None else torch.zeros(ctx.shape[1:], dtype=dt, device=dev) for g in grads]
        return _batched(grads)


def unbind_cameras(t: Tensor):
    """Per-camera views of a [C, ...] tensor (see `_UnbindFn`)."""
    return _UnbindFn.apply(t) if t.requires_grad else t.unbind(0)


_PINNED_ENDS: dict = {}


class _PackRecordsFn(torch.autograd.Function):
    """(opacities [N], C x (radii, means2d, depths, conics, compensations, rgbs)) -> records [M, 12] grouped by camera,
    ends [C] (CPU int64: one past each camera's last row)."""

    @staticmethod
    @_guarded(3)
    def forward(ctx, C, has_comp, opacities, *flat):
        lib = L.lib()
        groups = [flat[k * C:(k + 1) * C] for k in range(6)]
        radii = _batched(groups[0])
        if radii.dtype != torch.int32:
            radii = radii.to(torch.int32)
        means2d, depths, conics = (_f32c(_batched(g)) for g in groups[1:4])
        comps = _f32c(_batched(groups[4])) if has_comp else None
        rgbs = _f32c(_batched(groups[5]))
        opac = _f32c(opacities.detach()).reshape(-1)
        N, dev = radii.shape[1], radii.device
        assert opac.shape[0] == N and means2d.shape == (C, N, 2) and conics.shape == (C, N, 3) and rgbs.shape == (C, N, 3)
        records = torch.empty((max(C * N, 1), L.GSPL_RECORD_FLOATS), dtype=torch.float32, device=dev)
        slots = torch.empty((C, N), dtype=torch.int32, device=dev)
        ends = torch.empty((C,), dtype=torch.int64, device=dev)
        pool = _PINNED_ENDS.setdefault(C, [])
        host_ends = pool.pop() if pool else torch.empty((C,), dtype=torch.int64).pin_memory()
        ws_bytes = lib.gspl_records_workspace_bytes(C, N)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        L.call("gspl_records_pack_fwd", C, N, L.ptr(radii), L.ptr(means2d), L.ptr(depths), L.ptr(conics), L.ptr(comps), L.ptr(opac), L.ptr(rgbs),
               L.ptr(records), L.ptr(slots), L.ptr(ends), host_ends.data_ptr(), L.ptr(ws), ws_bytes, L.stream())
        ev = _take_event(dev)
        ev.record()
        ev.synchronize()                 # the split sizes of the all-to-all are needed on the host (as in the reference)
        _EVENTS[dev.index].append(ev)
        ends_cpu = host_ends.clone() if C * N > 0 else torch.zeros((C,), dtype=torch.int64)
        pool.append(host_ends)
        total = int(ends_cpu[-1]) if C > 0 else 0
        ctx.save_for_backward(slots)
        ctx.cfg = (C, N, has_comp, tuple(opacities.shape))
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(ends_cpu)
        return records[:total], ends_cpu

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_records, _v_ends):
        (slots,) = ctx.saved_tensors
        C, N, has_comp, opac_shape = ctx.cfg
        dev = slots.device
        if v_records is None:
            return (None,) * (3 + 6 * C)
        v_records = _f32c(v_records)
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        v_means2d, v_depths, v_conics, v_rgbs, v_opac = e(C, N, 2), e(C, N), e(C, N, 3), e(C, N, 3), e(N)
        v_comps = e(C, N) if has_comp else None
        L.call("gspl_records_pack_bwd", C, N, L.ptr(slots), L.ptr(v_records), L.ptr(v_means2d), L.ptr(v_depths), L.ptr(v_conics), L.ptr(v_comps),
               L.ptr(v_opac), L.ptr(v_rgbs), L.stream())
        per_cam = lambda t: [t[c] for c in range(C)] if t is not None else [None] * C
        return (None, None, v_opac.reshape(opac_shape), *([None] * C), *per_cam(v_means2d), *per_cam(v_depths), *per_cam(v_conics),
                *per_cam(v_comps), *per_cam(v_rgbs))


def pack_visible_records(projection_results_list, rgb_list, opacities: Tensor):
    """Records of every (camera, local splat) with radius > 0, grouped by camera (csrc/records.hip): what the reference builds
    with a concat + boolean-mask selection per camera (gsplat_distributed_renderer.py:313-360).
    projection_results_list[c] = (radii, means2d, depths, conics, compensations | None, ...).  Returns (records [M,12],
    counts per camera as a python list)."""
    C = len(projection_results_list)
    has_comp = projection_results_list[0][4] is not None
    cols = [[r[k] for r in projection_results_list] for k in range(5)]
    if not has_comp:
        cols[4] = [r[2] for r in projection_results_list]      # placeholder tensors (ignored)
    records, ends = _PackRecordsFn.apply(C, has_comp, opacities, *cols[0], *cols[1], *cols[2], *cols[3], *cols[4], *rgb_list)
    e = [0] + [int(v) for v in ends.tolist()]
    return records, [e[i + 1] - e[i] for i in range(C)]


class _PackAllRecordsFn(torch.autograd.Function):
    """(opacities [N], C x (radii, means2d, depths, conics, compensations, rgbs)) -> records [C*N, 12]: one row per (camera, local
    splat), camera-major, rows of invisible splats zero.  The fixed-size exchange format: nothing about it depends on a number
    the host would have to wait for.  Backward: `gspl_records_pack_bwd` with identity slots for the visible rows."""

    @staticmethod
    @_guarded(3)
    def forward(ctx, C, has_comp, opacities, *flat):
        groups = [flat[k * C:(k + 1) * C] for k in range(6)]
        radii = _batched(groups[0])
        if radii.dtype != torch.int32:
            radii = radii.to(torch.int32)
        means2d, depths, conics = (_f32c(_batched(g)) for g in groups[1:4])
        comps = _f32c(_batched(groups[4])) if has_comp else torch.ones_like(depths)
        rgbs = _f32c(_batched(groups[5]))
        opac = _f32c(opacities.detach()).reshape(-1)
        N, dev = radii.shape[1], radii.device
        assert opac.shape[0] == N and means2d.shape == (C, N, 2) and conics.shape == (C, N, 3) and rgbs.shape == (C, N, 3)
        vis = radii > 0
        rec = torch.cat([means2d, depths.unsqueeze(-1), conics, comps.unsqueeze(-1), opac.reshape(1, N, 1).expand(C, N, 1), rgbs,
                         radii.view(torch.float32).unsqueeze(-1)], dim=-1)
        rec = torch.where(vis.unsqueeze(-1), rec, _zero_scalar(dev))
        ident = _IDENTITY_SLOTS.get((C, N, dev))
        if ident is None:
            if len(_IDENTITY_SLOTS) > 8:
                _IDENTITY_SLOTS.clear()
            ident = _IDENTITY_SLOTS[(C, N, dev)] = torch.arange(C * N, dtype=torch.int32, device=dev).reshape(C, N)
        slots = torch.where(vis, ident, -1)
        ctx.save_for_backward(slots)
        ctx.cfg = (C, N, has_comp, tuple(opacities.shape))
        ctx.set_materialize_grads(False)
        return rec.reshape(C * N, L.GSPL_RECORD_FLOATS)

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_records):
        return _PackRecordsFn.backward(ctx, v_records, None)


_IDENTITY_SLOTS: dict = {}
_ZERO_SCALARS: dict = {}


def _zero_scalar(dev):
    z = _ZERO_SCALARS.get(dev)
    if z is None:
        z = _ZERO_SCALARS[dev] = torch.zeros((), dtype=torch.float32, device=dev)
    return z


def pack_all_records(projection_results_list, rgb_list, opacities: Tensor) -> Tensor:
    """Records of EVERY (camera, local splat), camera-major, invisible rows zeroed (radius 0 keeps them out of the receiver's
    lists): [C*N, 12].  Same arguments as `pack_visible_records`; no device read-back."""
    C = len(projection_results_list)
    has_comp = projection_results_list[0][4] is not None
    cols = [[r[k] for r in projection_results_list] for k in range(5)]
    if not has_comp:
        cols[4] = [r[2] for r in projection_results_list]      # placeholder tensors (ignored)
    return _PackAllRecordsFn.apply(C, has_comp, opacities, *cols[0], *cols[1], *cols[2], *cols[3], *cols[4], *rgb_list)


class _UnpackRecordsFn(torch.autograd.Function):
    @staticmethod
    @_guarded(1)
    def forward(ctx, records, fold_compensation):
        records = _f32c(records)
        M, dev = records.shape[0], records.device
        radii = torch.empty((M,), dtype=torch.int32, device=dev)
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        means2d, depths, conics, opac, rgbs = e(M, 2), e(M), e(M, 3), e(M), e(M, 3)
        L.call("gspl_records_unpack_fwd", M, int(bool(fold_compensation)), L.ptr(records), L.ptr(radii), L.ptr(means2d), L.ptr(depths), L.ptr(conics),
               L.ptr(opac), L.ptr(rgbs), L.stream())
        ctx.save_for_backward(records)
        ctx.fold = int(bool(fold_compensation))
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        return radii, means2d, depths, conics, opac, rgbs

    @staticmethod
    @_guarded(0)
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_opac, v_rgbs):
        (records,) = ctx.saved_tensors
        M = records.shape[0]
        v_means2d, s2 = _rows(v_means2d, 2)
        v_conics, s3 = _rows(v_conics, 3)
        v_rgbs, sc = _rows(v_rgbs, 3)
        s1 = 0
        if v_opac is not None:
            v_opac = v_opac.float() if v_opac.dtype != torch.float32 else v_opac
            flat = v_opac.reshape(-1) if v_opac.is_contiguous() else v_opac
            if flat.dim() == 1:
                v_opac, s1 = flat, (0 if flat.is_contiguous() else flat.stride(0))
            else:
                v_opac, s1 = v_opac.contiguous().reshape(-1), 0
        v_depths = _f32c(v_depths)
        v_records = torch.empty_like(records)
        L.call("gspl_records_unpack_bwd", M, ctx.fold, L.ptr(records), _raw_ptr(v_means2d), s2, L.ptr(v_depths), _raw_ptr(v_conics), s3,
               _raw_ptr(v_opac), s1, _raw_ptr(v_rgbs), sc, L.ptr(v_records), L.stream())
        return v_records, None


def unpack_visible_records(records: Tensor, fold_compensation: bool):
    """records [M,12] -> radii [M] i32, means2d [M,2], depths [M], conics [M,3], opacities [M] (x compensation when
    `fold_compensation`), rgbs [M,3] — the `torch.split` (+ the anti-aliasing product) of the reference's receiving side."""
    return _UnpackRecordsFn.apply(records, fold_compensation)


_PINNED_WORDS: list = []      # free list of pinned int64 words for the count read-back
_EVENTS: dict = {}            # device index -> free list of events (constructing one costs ~15 us of host time per frame)


def _take_event(dev):
    pool = _EVENTS.setdefault(dev.index, [])
    return pool.pop() if pool else torch.cuda.Event()


MAX_ISECTS = 2 ** 30 - 1      # RADIX_MAX_ITEMS of csrc/gspl_sort.h: the list positions and the sort's workgroup spans are 32-bit
_LAST_ISECTS: dict = {}       # (device, tile grid) -> list length of the last frame: the guess of the speculative emission
# How the guesses fared (bench.py reports the miss rate): frames binned, frames without a guess (first of a size: the host waits),
# frames whose guess was too low (emission, sort and — in the fused call — compositing are repeated).
SPECULATION = {"frames": 0, "cold": 0, "misses": 0}
SPECULATIVE_EMIT = os.environ.get("GSPL_SPECULATIVE_EMIT", "1") != "0"


@_guarded(0)
def bin_gaussians_begin(xys: Tensor, depths: Tensor, radii: Tensor, img_height: int, img_width: int, block_width: int = 16,
                        mode: int = L.GSPL_MODE_GSPLAT, conics: Optional[Tensor] = None,
                        opacities: Optional[Tensor] = None) -> _PendingBins:
    """First half of `bin_gaussians`: per-Gaussian tile counts, depth order and their scan (`gspl_bin_count`), then an
    ASYNCHRONOUS copy of the total to the host.  Work that does not depend on the lists (the SH kernel) can be launched
    before `bin_gaussians_end`, so the device is busy while the host waits for the one number that sizes the sort."""
    if block_width not in (8, 16, 32):
        raise NotImplementedError("block_width must be 8, 16 or 32 (the reference default is 16, gsplat_renderer.py:6)")
    lib = L.lib()
    p = _PendingBins()
    p.block_width = block_width
    p.tile_w, p.tile_h = (img_width + block_width - 1) // block_width, (img_height + block_width - 1) // block_width
    p.mode = mode
    p.means2d, depths = _f32c(xys.detach()), _f32c(depths.detach())
    p.radii = radii.to(torch.int32).contiguous()
    p.N = N = p.means2d.shape[0]
    p.dev = dev = p.means2d.device
    p.cull_c = p.cull_o = None
    if conics is not None and opacities is not None:
        p.cull_c, p.cull_o = _f32c(conics.detach()).reshape(-1, 3), _f32c(opacities.detach()).reshape(-1)
    # tiles + 1 entries: the device-side-length sort stores the list length behind the per-tile starts; callers get the first tiles
    p.offsets_buf = torch.empty((p.tile_w * p.tile_h + 1,), dtype=torch.int32, device=dev)
    p.offsets = p.offsets_buf[:p.tile_w * p.tile_h]
    p.order = p.cum = p.spans = p.host_count = p.event = p.ws2 = p.big_list = p.depths = None
    p.capacity = p.ws2_bytes = 0
    if N > 0:
        p.order = torch.empty((N,), dtype=torch.int32, device=dev)
        p.cum = torch.empty((N + 1,), dtype=torch.int64, device=dev)      # scan [N] + the number of big splats
        p.big_list = torch.empty((N,), dtype=torch.int32, device=dev)     # depth-order indices of the splats taller than 16 tile rows
        p.spans = torch.empty((N, L.GSPL_BIN_SPAN_BYTES // 4), dtype=torch.int32, device=dev)
        ws_bytes = lib.gspl_bin_workspace_bytes(N, 0)
        if ws_bytes == 0:
            raise RuntimeError("gspl_bin_workspace_bytes failed: " + lib.gspl_last_error().decode())
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        p.depths = depths

        # the one host read-back of the pipeline: the list length (sizes the sort buffers) and the number of big splats, stored into
        # pinned host memory by the scan kernel itself (no copy launch)
        p.host_count = _PINNED_WORDS.pop() if _PINNED_WORDS else torch.empty((2,), dtype=torch.int64).pin_memory()
        L.call("gspl_bin_count", N, mode, L.ptr(p.means2d), L.ptr(p.radii), L.ptr(depths), L.ptr(p.cull_c), L.ptr(p.cull_o),
               block_width, p.tile_w, p.tile_h, L.ptr(p.order), L.ptr(p.cum), L.ptr(p.big_list), L.ptr(p.spans), p.host_count.data_ptr(),
               L.ptr(ws), ws_bytes, L.stream())
        p.event = _take_event(dev)
        p.event.record()
        # Speculative emission: the emit kernel's grid depends on N only, so it is launched NOW with room for a guess of
        # the list length (the last frame's, plus a margin) and runs while the host waits for the real number; a guess
        # that turns out too low costs one repeated emission in `bin_gaussians_end`.
        guess = _LAST_ISECTS.get((dev.index, p.tile_w, p.tile_h), 0)
        if SPECULATIVE_EMIT and guess > 0:
            p.capacity = min(int(guess * 1.25) + 65536, MAX_ISECTS)
            p.ws2_bytes = lib.gspl_bin_workspace_bytes(N, p.capacity)
            if p.ws2_bytes == 0:
                raise RuntimeError("gspl_bin_workspace_bytes failed: " + lib.gspl_last_error().decode())
            p.ws2 = torch.empty((p.ws2_bytes,), dtype=torch.uint8, device=dev)
            _emit(p)
    return p


def _emit(p: "_PendingBins"):
    L.call("gspl_bin_emit", p.N, p.mode, L.ptr(p.means2d), L.ptr(p.radii), L.ptr(p.cull_c), L.ptr(p.cull_o), L.ptr(p.order), L.ptr(p.cum),
           L.ptr(p.big_list), L.ptr(p.spans), p.block_width, p.tile_w, p.tile_h, p.capacity, L.ptr(p.ws2), p.ws2_bytes, L.stream())


class LazyLists:
    """The per-tile lists of a binning whose LENGTH the host does not know yet (`bin_gaussians_end(p, lazy=True)`): the records were
    emitted with room for a guess, the sort reads the real length on the device, and the compositing call that receives this object
    in place of `flatten_ids` is launched on the capacity-sized buffer with the device-side end of the last list (n_isects = -1)
    BEFORE the host looks at the count — by then the device is long past it, so the frame has no blocking wait.  A guess that was
    too low repeats emission, sort and that compositing launch.  After the first compositing call (or `resolve()`), `flat` is the
    exact-length tensor; `offsets` is valid (as device memory) from the start."""
    __slots__ = ("p", "flat_cap", "flat", "offsets", "offsets_ext", "settled", "held")

    def __init__(self, p: "_PendingBins", flat_cap: Tensor):
        self.p, self.flat_cap, self.flat = p, flat_cap, None
        self.offsets, self.offsets_ext = p.offsets, p.offsets_buf
        self.settled = self.held = False

    def settle(self) -> bool:
        """Wait for the count (a formality once later work has been enqueued) and fix the lists: True if the guess held."""
        if not self.settled:
            with L.device_guard(self.p.dev):
                n_isects = _bin_count_arrived(self.p)
                self.held = 0 < n_isects <= self.p.capacity
                if self.held:
                    self.flat = self.flat_cap[:n_isects]
                    self.p.ws2 = None
                else:
                    self.flat, self.offsets = _bin_finish(self.p, n_isects)
            self.settled, self.flat_cap = True, None
        return self.held

    def resolve(self):
        """(flatten_ids, offsets) as tensors (waits for the count if nobody has yet)."""
        self.settle()
        return self.flat, self.offsets


def bin_gaussians_end(p: _PendingBins, lazy: bool = False):
    """Second half: waits for the count, then (emits and) sorts the (tile, Gaussian) lists.
    Returns (flatten_ids [I] i32, offsets [tile_h*tile_w] i32).  lazy=True: (LazyLists, offsets) when the emission was speculative —
    for callers that hand the lists straight to a compositing call of this module (see `LazyLists`)."""
    with L.device_guard(p.dev):
        return _bin_gaussians_end(p, lazy)


def _bin_count_arrived(p: _PendingBins) -> int:
    """The list length of the frame (blocks until the scan kernel's store to pinned memory is visible) + the speculation book-keeping."""
    p.event.synchronize()
    _EVENTS[p.dev.index].append(p.event)
    n_isects = int(p.host_count[0])
    _PINNED_WORDS.append(p.host_count)
    if n_isects > MAX_ISECTS:
        raise RuntimeError(f"{n_isects} (tile, Gaussian) intersections in one frame: the per-tile lists of this library hold at most "
                           f"2^30-1 = {MAX_ISECTS} entries (fewer / smaller Gaussians, a larger tile size or a lower resolution)")
    _LAST_ISECTS[(p.dev.index, p.tile_w, p.tile_h)] = n_isects
    SPECULATION["frames"] += 1
    if p.capacity == 0:
        SPECULATION["cold"] += 1
    elif n_isects > p.capacity:
        SPECULATION["misses"] += 1
    return n_isects


def _bin_finish(p: _PendingBins, n_isects: int):
    """Emission (again, if the guess was too low or there was none) and sort with the list length known to the host."""
    lib = L.lib()
    N, dev = p.N, p.dev
    flat = torch.empty((n_isects,), dtype=torch.int32, device=dev)
    if n_isects > 0 and (p.ws2 is None or p.capacity < n_isects):
        p.capacity = n_isects
        p.ws2_bytes = lib.gspl_bin_workspace_bytes(N, n_isects)
        p.ws2 = torch.empty((p.ws2_bytes,), dtype=torch.uint8, device=dev)
        _emit(p)
    L.call("gspl_bin_sort", N, p.tile_w, p.tile_h, n_isects, max(p.capacity, n_isects), L.ptr(flat) if n_isects else None, L.ptr(p.offsets),
           L.ptr(p.ws2) if n_isects else None, p.ws2_bytes if n_isects else 0, L.stream())
    p.ws2 = None
    return flat, p.offsets


def _bin_gaussians_end(p: _PendingBins, lazy: bool = False):
    N, dev = p.N, p.dev
    if N > 0 and p.ws2 is not None and DEVICE_SIDE_LIST_LENGTH:
        # The records were emitted speculatively: sort them BEFORE the host knows how many there are (the sort reads the length on
        # the device, its grid is sized by the capacity), so that the device has the whole sort queued while the host waits for the
        # count — and check the guess afterwards.
        flat_cap = torch.empty((p.capacity,), dtype=torch.int32, device=dev)
        L.call("gspl_bin_sort_device_count", N, p.tile_w, p.tile_h, L.ptr(p.cum, offset_bytes=8 * (N - 1)), p.capacity, L.ptr(flat_cap),
               L.ptr(p.offsets_buf), L.ptr(p.ws2), p.ws2_bytes, L.stream())
        lz = LazyLists(p, flat_cap)
        return (lz, p.offsets) if lazy else lz.resolve()
    n_isects = _bin_count_arrived(p) if N > 0 else 0
    return _bin_finish(p, n_isects)


def bin_gaussians(xys: Tensor, depths: Tensor, radii: Tensor, img_height: int, img_width: int, block_width: int = 16,
                  mode: int = L.GSPL_MODE_GSPLAT, conics: Optional[Tensor] = None, opacities: Optional[Tensor] = None, lazy: bool = False):
    """Binning half of `rasterize_gaussians`, exposed so that several compositing passes over the same
    projection (rgb + depth variants, gsplat_renderer.py:101-185) share one sort.
    With `conics` and `opacities` (the ones the compositing call will use) tile hits that cannot reach
    alpha >= 1/255 anywhere in the tile are not listed — same images and gradients, ~40 % shorter lists.
    Returns (flatten_ids [I] i32, offsets [tile_h*tile_w] i32)."""
    return bin_gaussians_end(bin_gaussians_begin(xys, depths, radii, img_height, img_width, block_width, mode, conics, opacities), lazy)


def rasterize_gaussians(xys: Tensor, depths: Tensor, radii: Tensor, conics: Tensor, num_tiles_hit: Tensor,
                        colors: Tensor, opacity: Tensor, img_height: int, img_width: int, block_width: int,
                        background: Optional[Tensor] = None, return_alpha: bool = False, absgrad: bool = False,
                        isects=None, channels_first: bool = False):
    """gsplat-v0 signature (reference call: gsplat_renderer.py:86-99): bins + composites in one call.
    colors [N,D], opacity [N,1] -> [H,W,D] (and alpha [H,W] when return_alpha).
    channels_first (extension): the image comes out as [D,H,W] straight from the kernel — what the reference builds with
    `.permute(2, 0, 1)` and every consumer (loss, metrics) then has to make contiguous, forward and backward."""
    if block_width not in (8, 16, 32):
        raise NotImplementedError("block_width must be 8, 16 or 32 (the reference default is 16, gsplat_renderer.py:6)")
    flat, offsets = isects if isects is not None else bin_gaussians(xys, depths, radii, img_height, img_width, block_width,
                                                                    conics=conics, opacities=opacity, lazy=True)
    out, alphas = _composite(xys, conics, colors, opacity.reshape(-1), background, img_width, img_height, block_width,
                             offsets, flat, absgrad, L.GSPL_MODE_GSPLAT, L.GSPL_LAYOUT_CHW if channels_first else L.GSPL_LAYOUT_HWC)
    return (out, alphas) if return_alpha else out


# =============================================================================================
# The Gaussian-sharded renderer's step as THREE autograd nodes instead of eleven
# (internal/renderers/gsplat_distributed_renderer.py:252-311 project + colours, :127-211 exchange, :356-389 rasterize)
# =============================================================================================
# The staged formulation of that step (fully_fused_projection -> 5 x unbind_cameras -> sh_view_colors_batched ->
# pack_visible_records -> all_to_all_rows -> unpack_visible_records -> bin_gaussians -> rasterize_to_pixels) costs the host 1.3-1.5 ms
# per step at 1 M Gaussians for 1.25-1.38 ms of kernels (tools/micro/host_sharded_profile.py): eleven autograd nodes, each with its
# Python forward, its engine dispatch in the backward and its per-camera tuples.  The three nodes below run the SAME stage bodies —
# the forward / backward static methods of the stage wrappers above, called with a stand-in context, so there is one copy of every
# launch sequence — and hand batched [C, N, ...] buffers from stage to stage:
#
#   sharded_front     project (C cameras, one launch) -> SH colours (C cameras, one launch) -> pack the visible splats' records
#   sharded_exchange  the all-to-all of the records (a callable of the caller: this module knows no process groups) AND the tap that
#                     gives every camera's screen-space positions their gradient (`xys[c].grad` is what the reference's
#                     DistributedVanillaDensityControllerImpl reads): its backward turns the record gradients into per-(camera, splat)
#                     gradients once (gspl_records_pack_bwd), returns the means2d part as the gradient of `xys` and leaves the rest
#                     in the step's `stash` for the front node's backward
#   sharded_back      unpack -> bin (lists whose length stays on the device) -> composite
class _StageCtx:
    """Stand-in for the autograd context of one stage wrapper."""
    __slots__ = ("saved_tensors", "needs_input_grad", "cfg", "fold", "means2d_ref")

    def __init__(self, needs_input_grad=()):
        self.saved_tensors = ()
        self.needs_input_grad = needs_input_grad
        self.means2d_ref = None

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


def _save_stages(ctx, stages):
    """Keep the tensors the stage bodies saved through the REAL context (no attribute references to output tensors: those would
    be reference cycles through grad_fn), with the split points to rebuild the stand-in contexts in the backward."""
    flat, cuts = [], []
    for s in stages:
        flat.extend(s.saved_tensors)
        cuts.append(len(flat))
        s.saved_tensors = ()
    ctx.save_for_backward(*flat)
    ctx.cuts = cuts


def _load_stages(ctx, stages):
    saved, lo = ctx.saved_tensors, 0
    for s, hi in zip(stages, ctx.cuts):
        s.saved_tensors = saved[lo:hi]
        lo = hi


class _ShardFrontFn(torch.autograd.Function):
    @staticmethod
    @_guarded(1)
    def forward(ctx, means, scales, quats, opacities, dc, rest, viewmats, Ks, centers, width, height, eps2d, degree, padded, stash):
        C = viewmats.shape[0]
        proj, sh, pack = _StageCtx(), _StageCtx(), _StageCtx()
        radii, means2d, depths, conics, comps, _, _ = _ProjectFn.forward(
            proj, means, scales, quats, viewmats, Ks, width, height, 16, 1.0, eps2d, 0.01, 1e10, 0.0, True, False, L.GSPL_CAMERA_PINHOLE, False)
        lib = L.lib()
        N, dev = means.shape[0], radii.device
        opac = _f32c(opacities.detach()).reshape(-1)
        assert opac.shape[0] == N
        slots = torch.empty((C, N), dtype=torch.int32, device=dev)
        records = torch.empty((max(C * N, 1), L.GSPL_RECORD_FLOATS), dtype=torch.float32, device=dev)
        if padded:
            # the fixed-size format: one record per (camera, local splat), invisible rows zeroed — no count, no wait
            colors = _SHBatchedFn.forward(sh, degree, means, centers, dc, rest, radii)
            L.call("gspl_records_pad_fwd", C, N, L.ptr(radii), L.ptr(means2d), L.ptr(depths), L.ptr(conics), L.ptr(comps), L.ptr(opac),
                   L.ptr(colors), L.ptr(records), L.ptr(slots), L.stream())
            ends = torch.arange(1, C + 1, dtype=torch.int64) * N
            records = records[:C * N]
        else:
            # The pack in two phases (csrc/records.hip): the record COUNTS need the radii only, so they are on their way to the host
            # (pinned memory, an event behind them) before the colour kernel is even launched; the host waits for them with the
            # colour kernel and the scatter still queued on the device — the wait of the counted exchange (the reference's
            # gsplat_distributed_renderer.py:141-160 reads the counts back after everything) no longer drains the stream.
            ends_dev = torch.empty((C,), dtype=torch.int64, device=dev)
            pool = _PINNED_ENDS.setdefault(C, [])
            host_ends = pool.pop() if pool else torch.empty((C,), dtype=torch.int64).pin_memory()
            ws_bytes = lib.gspl_records_workspace_bytes(C, N)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            L.call("gspl_records_count_fwd", C, N, L.ptr(radii), L.ptr(slots), L.ptr(ends_dev), host_ends.data_ptr(), L.ptr(ws), ws_bytes, L.stream())
            ev = _take_event(dev)
            ev.record()
            colors = _SHBatchedFn.forward(sh, degree, means, centers, dc, rest, radii)
            L.call("gspl_records_scatter_fwd", C, N, L.ptr(radii), L.ptr(slots), L.ptr(means2d), L.ptr(depths), L.ptr(conics), L.ptr(comps),
                   L.ptr(opac), L.ptr(colors), L.ptr(records), L.stream())
            ev.synchronize()                 # the split sizes of the all-to-all are needed on the host (as in the reference)
            _EVENTS[dev.index].append(ev)
            ends = host_ends.clone() if C * N > 0 else torch.zeros((C,), dtype=torch.int64)
            pool.append(host_ends)
            records = records[:int(ends[-1]) if C > 0 else 0]
        pack.save_for_backward(slots)
        pack.cfg = (C, N, True, tuple(opacities.shape))
        # the pack stage's state travels in the stash: the exchange node's backward runs that stage's backward
        stash["pack"] = pack
        ctx.stash = stash
        ctx.stages = (proj, sh)
        _save_stages(ctx, ctx.stages)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(ends, radii, depths, conics, comps)
        return records, ends, radii, means2d, depths, conics, comps

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_records, _v_ends, _v_radii, v_means2d, _v_depths, _v_conics, _v_comps):
        if v_records is not None:
            raise RuntimeError("the records of ops.sharded_front must reach their consumer through ops.sharded_exchange")
        proj, sh = ctx.stages
        _load_stages(ctx, ctx.stages)
        rest_of = ctx.stash.pop("grads", None)      # left by _ShardExchangeFn.backward, which the engine runs before this node
        v_depths = v_conics = v_comps = v_opac = v_colors = None
        if rest_of is not None:
            v_depths, v_conics, v_comps, v_opac, v_colors = rest_of
        v_dc = v_rest = None
        if v_colors is not None:
            _, _, _, v_dc, v_rest, _ = _SHBatchedFn.backward(sh, v_colors)
        v_means = v_scales = v_quats = None
        if v_means2d is not None or v_conics is not None:
            v_means, v_scales, v_quats = _ProjectFn.backward(proj, None, v_means2d, v_depths, v_conics, v_comps, None, None)[:3]
        return (v_means, v_scales, v_quats, v_opac, v_dc, v_rest) + (None,) * 9


class _ShardExchangeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, records, stash, route, *xys):
        ctx.stash, ctx.route, ctx.C = stash, route, len(xys)
        ctx.set_materialize_grads(False)
        if route is None:
            return records.view_as(records)
        return route[0](records)

    @staticmethod
    def backward(ctx, v_records):
        C = ctx.C
        if v_records is None:
            return (None,) * (3 + C)
        if ctx.route is not None:
            v_records = ctx.route[1](v_records)
        pack = ctx.stash.pop("pack", None)      # (released here: the pack stage's buffers are dead after this backward)
        if pack is None:
            raise RuntimeError("sharded_exchange: the backward of this step has already run and released its pack state; the three-node "
                               "step is single-use (fused_step=False gives the stage-by-stage step, which supports retain_graph)")
        grads = _PackRecordsFn.backward(pack, v_records, None)
        # (None, None, v_opac, C x None (radii), C x v_means2d, C x v_depths, C x v_conics, C x v_comps, C x v_colors): per-camera
        # slices of one buffer each
        v_opac = grads[2]
        per = lambda k: grads[3 + k * C:3 + (k + 1) * C]
        ctx.stash["grads"] = (_batched(per(2)), _batched(per(3)), _batched(per(4)), v_opac, _batched(per(5)))
        return (None, None, None) + tuple(per(1))


class _ShardBackFn(torch.autograd.Function):
    @staticmethod
    @_guarded(1)
    def forward(ctx, records, backgrounds, width, height, tile_size, fold_compensation, cull):
        unpack, comp = _StageCtx(), _StageCtx()
        radii, means2d, depths, conics, opac, colors = _UnpackRecordsFn.forward(unpack, records, fold_compensation)
        flat, offsets = bin_gaussians(means2d, depths, radii, height, width, tile_size, conics=conics if cull else None,
                                      opacities=opac if cull else None, lazy=True)
        out, alphas = _CompositeFn.forward(comp, means2d, conics, colors, opac, backgrounds, width, height, tile_size, offsets, flat,
                                           False, L.GSPL_MODE_GSPLAT, L.GSPL_LAYOUT_CHW, False)
        ctx.stages = (unpack, comp)
        _save_stages(ctx, ctx.stages)
        ctx.set_materialize_grads(False)
        return out, alphas

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_out, v_alphas):
        unpack, comp = ctx.stages
        _load_stages(ctx, ctx.stages)
        comp.needs_input_grad = (False, False, False, False, ctx.needs_input_grad[1])
        v_means2d, v_conics, v_colors, v_opac, v_bg = _CompositeFn.backward(comp, v_out, v_alphas)[:5]
        v_records, _ = _UnpackRecordsFn.backward(unpack, None, v_means2d, None, v_conics, v_opac, v_colors)
        return v_records, v_bg, None, None, None, None, None


def sharded_front(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, shs_dc: Tensor, shs_rest: Optional[Tensor],
                  viewmats: Tensor, Ks: Tensor, camera_centers: Tensor, width: int, height: int, eps2d: float, sh_degree: int,
                  stash: dict, padded: bool = False):
    """This rank's shard seen from C cameras, packed for the exchange — `fully_fused_projection(calc_compensations=True)` +
    `sh_view_colors_batched` + `pack_visible_records` as ONE autograd node (means are detached for the colours, as
    gsplat_distributed_renderer.py:417 does).  viewmats [C,4,4], Ks [C,3,3], camera_centers [C,3]; `stash`: a dict private to
    this step, handed to `sharded_exchange` as well.
    padded=False: the records of the VISIBLE splats, compacted (counted exchange: the call waits for the per-camera counts — which
    leave the device before the colour kernel runs); padded=True: one record per (camera, local splat), invisible rows zeroed
    (fixed-size exchange: no count, no wait).
    Returns (records [M,12] grouped by camera, counts per camera (python list), radii [C,N] i32, means2d [C,N,2], depths [C,N],
    conics [C,N,3], compensations [C,N]).  Only `records` and `means2d` carry gradients — means2d through `sharded_exchange`'s
    `xys` argument; radii / depths / conics / compensations are handed out for inspection (detached)."""
    records, ends, radii, means2d, depths, conics, comps = _ShardFrontFn.apply(
        means, scales, quats, opacities, shs_dc, shs_rest, viewmats, Ks, camera_centers, int(width), int(height), float(eps2d),
        int(sh_degree), bool(padded), stash)
    e = [0] + [int(v) for v in ends.tolist()]
    return records, [e[i + 1] - e[i] for i in range(len(e) - 1)], radii, means2d, depths, conics, comps


def sharded_exchange(records: Tensor, stash: dict, xys: Sequence[Tensor], route=None) -> Tensor:
    """The records on their way to the ranks that render them.  `route`: None (one rank: nothing travels) or a pair of callables
    (forward, backward) mapping the sent rows to the received rows and the received rows' gradients back to the sent rows' (the
    all-to-all with split sizes and its reverse: `distributed.all_to_all_route`).  `xys`: the per-camera views of `sharded_front`'s
    means2d (`unbind_cameras`); after a backward pass `xys[c].grad` (with `retain_grad()`) is d loss / d means2d of camera c."""
    return _ShardExchangeFn.apply(records, stash, route, *xys)


def sharded_back(records: Tensor, backgrounds: Optional[Tensor], width: int, height: int, tile_size: int = 16,
                 fold_compensation: bool = True, tile_based_culling: bool = False):
    """Received records -> image: `unpack_visible_records` + `bin_gaussians` (list-only, optional tile-based culling, list length on
    the device) + `rasterize_to_pixels(channels_first=True)` as ONE autograd node.  Returns (image [D,H,W], alphas [H,W])."""
    if tile_size not in (8, 16, 32):
        raise NotImplementedError("tile_size must be 8, 16 or 32")
    return _ShardBackFn.apply(records, backgrounds, int(width), int(height), int(tile_size), bool(fold_compensation), bool(tile_based_culling))


# =============================================================================================
# Inria API  (diff_gaussian_rasterization.GaussianRasterizer)
# =============================================================================================
class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool = False
    debug: bool = False


def _split_sh(sh, sh_rest):
    """(sh, sh_rest, n_coeffs) of the rasterizer's colour input: `sh` [N, K, 3] alone, or the reference model's two parameters
    `shs_dc` [N, 1, 3] and `shs_rest` [N, K - 1, 3] (internal/models/vanilla_gaussian.py:266-300), which the kernels then read — and
    whose gradients they write — in place: `get_shs()`'s per-step `torch.cat` (and its backward's two slice copies) never run."""
    if sh is None:
        if sh_rest is not None:
            raise ValueError("shs_rest without shs")
        return None, None, 0
    if sh_rest is None:
        return sh, None, sh.shape[1]
    sh_rest = _f32c(sh_rest)
    if sh.dim() != 3 or sh.shape[1] != 1 or sh_rest.dim() != 3 or sh_rest.shape[0] != sh.shape[0] or sh_rest.shape[2] != 3:
        raise ValueError(f"shs / shs_rest must be [N,1,3] and [N,K-1,3], got {tuple(sh.shape)} and {tuple(sh_rest.shape)}")
    if sh_rest.shape[1] == 0:
        return sh, None, 1
    return sh, sh_rest, 1 + sh_rest.shape[1]


class _InriaRasterizeFn(torch.autograd.Function):
    @staticmethod
    @_guarded(1)
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings, sh_rest=None):
        lib = L.lib()
        s: GaussianRasterizationSettings = settings
        dev = means3D.device
        means3D = _f32c(means3D)
        N = means3D.shape[0]
        H, W = int(s.image_height), int(s.image_width)
        sh, colors_precomp, scales, rotations, cov3D_precomp = map(_f32c, (sh, colors_precomp, scales, rotations, cov3D_precomp))
        opac = _f32c(opacities).reshape(-1)
        viewm, projm, campos = _f32c(s.viewmatrix), _f32c(s.projmatrix), _f32c(s.campos)
        bg = _f32c(s.bg)
        sh, sh_rest, n_coeffs = _split_sh(sh, sh_rest)
        radii = torch.empty((N,), dtype=torch.int32, device=dev)
        means2d = torch.empty((N, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((N,), dtype=torch.float32, device=dev)
        conics = torch.empty((N, 3), dtype=torch.float32, device=dev)
        colors = torch.empty((N, 3), dtype=torch.float32, device=dev)
        clamped = torch.empty((N, 3), dtype=torch.uint8, device=dev)
        cov3d = torch.empty((N, 6), dtype=torch.float32, device=dev)
        # d colour / d view direction, left by the colour kernel for the backward (which then reads no coefficients for v_means)
        sh_jac = torch.empty((N, 9), dtype=torch.float32, device=dev) if (sh is not None and ctx.needs_input_grad[0]) else None
        tile = 16
        tile_w, tile_h = (W + tile - 1) // tile, (H + tile - 1) // tile
        def preprocess(phases):
            if N > 0:
                if phases & L.GSPL_INRIA_COLOURS:
                    _await_updates(sh, sh_rest)
                L.call("gspl_inria_preprocess_fwd",
                       N, int(s.sh_degree), n_coeffs, L.ptr(means3D), L.ptr(scales), L.ptr(rotations), L.ptr(cov3D_precomp),
                       L.ptr(sh), L.ptr(sh_rest), L.ptr(colors_precomp), L.ptr(viewm), L.ptr(projm), L.ptr(campos), W, H, tile,
                       float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier),
                       L.ptr(radii), L.ptr(means2d), L.ptr(depths), L.ptr(conics), L.ptr(colors), L.ptr(clamped), L.ptr(cov3d),
                       L.ptr(sh_jac) if (phases & L.GSPL_INRIA_COLOURS) else None, phases, L.stream())
        # geometry, then two independent chains: the SH kernel (HBM-bound, one launch) on a side stream, and the count /
        # depth-sort half of the binning (a dozen small latency-bound launches) on the caller's stream; the host meanwhile
        # waits for the one number that sizes the tile sort.
        preprocess(L.GSPL_INRIA_GEOMETRY)
        with _side_stream(dev) as side:
            preprocess(L.GSPL_INRIA_COLOURS)
        pending = bin_gaussians_begin(means2d, depths, radii, H, W, tile, mode=L.GSPL_MODE_INRIA, conics=conics, opacities=opac)
        flat, offsets = bin_gaussians_end(pending)
        side.join()
        n_isects = flat.shape[0]
        out = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        alphas = torch.empty((H, W), dtype=torch.float32, device=dev)
        final_Ts = torch.empty((H, W), dtype=torch.float32, device=dev)
        last_ids = torch.empty((H, W), dtype=torch.int32, device=dev)
        L.call("gspl_composite_fwd", 
            N, n_isects, 3, L.GSPL_MODE_INRIA, L.GSPL_LAYOUT_CHW, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opac),
            L.ptr(bg), W, H, tile, tile_w, tile_h, L.ptr(offsets), L.ptr(flat) if n_isects else None,
            L.ptr(out), L.ptr(alphas), L.ptr(final_Ts), L.ptr(last_ids), None, L.stream())
        ctx.save_for_backward(means3D, scales, rotations, cov3D_precomp, sh, opac, viewm, projm, campos, bg,
                              radii, means2d, conics, colors, clamped, cov3d, offsets, flat, final_Ts, last_ids, sh_jac, sh_rest)
        if KEEP_LAST_RASTER:
            global LAST_RASTER
            LAST_RASTER = dict(mode=L.GSPL_MODE_INRIA, width=W, height=H, means2d=means2d, conics=conics, opacities=opac,
                               colors=colors, flatten_ids=flat, offsets=offsets, radii=radii, depths=depths)
        ctx.cfg = (H, W, tile, tile_w, tile_h, int(s.sh_degree), n_coeffs, float(s.tanfovx), float(s.tanfovy),
                   float(s.scale_modifier), colors_precomp is not None, opacities.shape)
        ctx.set_materialize_grads(False)      # the integer `radii` output would otherwise get a zero "gradient" tensor per step
        ctx.mark_non_differentiable(radii)
        ctx.means2D_ref = means2D       # the caller's screen-space tensor: `.has_hit_any_pixels` is attached to it in backward
        return out, radii

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_out, _v_radii):
        lib = L.lib()
        (means3D, scales, rotations, cov3D_precomp, sh, opac, viewm, projm, campos, bg,
         radii, means2d, conics, colors, clamped, cov3d, offsets, flat, final_Ts, last_ids, sh_jac, sh_rest) = ctx.saved_tensors
        H, W, tile, tile_w, tile_h, degree, n_coeffs, tanfovx, tanfovy, scale_modifier, has_precomp_colors, opac_shape = ctx.cfg
        N = means3D.shape[0]
        dev = means3D.device
        n_isects = flat.shape[0]
        v_out = _grad_or_zeros(v_out, (3, H, W), dev)
        RS = _packed_row_stride(9)
        packed = torch.zeros((N, RS), dtype=torch.float32, device=dev)       # x y | a b c | opacity | r g b | pad
        if n_isects > 0:
            hit = torch.zeros((N,), dtype=torch.uint8, device=dev) if TRACK_HIT_PIXELS else None
            L.call("gspl_composite_bwd_packed",
                N, n_isects, 3, L.GSPL_MODE_INRIA, L.GSPL_LAYOUT_CHW, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opac),
                L.ptr(bg), W, H, tile, tile_w, tile_h, L.ptr(offsets), L.ptr(flat), L.ptr(final_Ts), L.ptr(last_ids),
                L.ptr(v_out), None, L.ptr(packed), RS, 0, L.ptr(hit), L.stream())
            if hit is not None and ctx.means2D_ref is not None:
                ctx.means2D_ref.has_hit_any_pixels = hit.bool()
        v_opac = torch.empty((N,), dtype=torch.float32, device=dev)      # dense copy of the packed column (written below)
        v_means = torch.empty((N, 3), dtype=torch.float32, device=dev)
        v_ndc = torch.empty((N, 3), dtype=torch.float32, device=dev)
        use_cov = cov3D_precomp is not None
        v_scales = None if use_cov else torch.empty((N, 3), dtype=torch.float32, device=dev)
        v_quats = None if use_cov else torch.empty((N, 4), dtype=torch.float32, device=dev)
        v_cov = torch.empty((N, 6), dtype=torch.float32, device=dev) if use_cov else None
        v_sh = None if has_precomp_colors else torch.empty_like(sh)
        v_sh_rest = None if sh_rest is None else torch.empty_like(sh_rest)
        v_cp = torch.empty((N, 3), dtype=torch.float32, device=dev) if has_precomp_colors else None
        if N > 0:
            L.call("gspl_inria_preprocess_bwd", 
                N, degree, n_coeffs, L.ptr(means3D), L.ptr(scales), L.ptr(rotations), L.ptr(cov3d), L.ptr(sh), L.ptr(sh_rest),
                L.ptr(viewm), L.ptr(projm), L.ptr(campos), W, H, tanfovx, tanfovy, scale_modifier,
                L.ptr(radii), L.ptr(clamped), L.ptr(packed), L.ptr(packed, offset_bytes=8), L.ptr(packed, offset_bytes=24), RS,
                L.ptr(v_means), L.ptr(v_scales), L.ptr(v_quats), L.ptr(v_cov), L.ptr(v_sh), L.ptr(v_sh_rest), L.ptr(v_cp), L.ptr(v_ndc),
                L.ptr(packed, offset_bytes=20), L.ptr(v_opac), L.ptr(sh_jac), L.stream())
        # order: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings, sh_rest
        return v_means, v_ndc, v_sh, v_cp, v_opac.reshape(opac_shape), v_scales, v_quats, v_cov, None, v_sh_rest


# ---- the same rasterizer through ONE C-ABI call per direction (gspl_rasterize_inria_fwd/bwd, csrc/fused.hip) -----------------
FUSED_INRIA = os.environ.get("GSPL_FUSED_INRIA", "1") != "0"
# The colour stream at the device's lowest priority (a stream from the library; torch cannot create one below the default) or at the
# default priority (the package's torch side stream).  Round 3, 16 rotating cameras, two runs each on one box: no difference for the
# colour kernel alone (1.280 / 1.280 vs 1.292 / 1.273 ms per step), and with the deferred shs_rest update on the same stream the
# default priority is the faster one (1.251 / 1.257 vs 1.266 / 1.264) — and the only two runs with 6-9 ms stalls of single steps had
# the low-priority stream carrying the update.  Default: the default priority.
SIDE_LOW_PRIORITY = os.environ.get("GSPL_SIDE_LOW_PRIORITY", "0") != "0"
_ALLOC_TLS = __import__("threading").local()


def _alloc_trampoline(_ctx, tag, nbytes):
    """`gspl_alloc_fn`: hand the library a block of torch-owned device memory; the tensors stay with the caller's holder."""
    holder = _ALLOC_TLS.holder
    try:
        t = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=holder["device"])
        holder.setdefault(tag, []).append(t)
        return t.data_ptr()
    except Exception as e:      # an exception must not cross the C boundary: NULL = failure, re-raised by the caller
        holder["error"] = e
        return 0


_ALLOC_CB = L.ALLOC_FN(_alloc_trampoline)


def _view(buf: Tensor, ptr: int, shape, dtype) -> Tensor:
    """Typed view of a region of a byte buffer the library carved up (ptr = device address inside `buf`)."""
    import math
    off = ptr - buf.data_ptr()
    nbytes = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
    return buf[off:off + nbytes].view(dtype).view(shape)


class _InriaFusedFn(torch.autograd.Function):
    @staticmethod
    @_guarded(1)
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings, sh_rest=None):
        import ctypes
        s: GaussianRasterizationSettings = settings
        dev = means3D.device
        means3D = _f32c(means3D)
        N = means3D.shape[0]
        H, W = int(s.image_height), int(s.image_width)
        sh, colors_precomp, scales, rotations, cov3D_precomp = map(_f32c, (sh, colors_precomp, scales, rotations, cov3D_precomp))
        opac = _f32c(opacities).reshape(-1)
        viewm, projm, campos, bg = _f32c(s.viewmatrix), _f32c(s.projmatrix), _f32c(s.campos), _f32c(s.bg)
        sh, sh_rest, n_coeffs = _split_sh(sh, sh_rest)
        out = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((N,), dtype=torch.int32, device=dev)
        tile_w, tile_h = (W + 15) // 16, (H + 15) // 16
        key = (dev.index, tile_w, tile_h)
        guess = _LAST_ISECTS.get(key, 0)
        hint = min(int(guess * 1.25) + 65536, MAX_ISECTS) if (SPECULATIVE_EMIT and guess > 0) else 0
        state = L.InriaState()
        holder = {"device": dev}
        _ALLOC_TLS.holder = holder
        side = _side_stream(dev)
        with torch.cuda.device(dev):
            side_handle = None
            low = False
            raw = None
            if side.enabled:
                # the colour stream (default priority; GSPL_SIDE_LOW_PRIORITY=1: the library's lowest-priority stream, on which the
                # colour kernel yields to the key pass and the depth sort it runs next to)
                raw, _ = colour_stream(dev)
                low = _side_stream._low.get((dev.type, dev.index), False)
                side_handle = ctypes.c_void_p(raw)
            # coefficient updates still in flight (FusedAdam(deferred=...)): on the colour stream itself stream order covers them
            _await_updates(sh, sh_rest, on_raw_stream=raw)
            try:
                L.call("gspl_rasterize_inria_fwd", N, int(s.sh_degree), n_coeffs, L.ptr(means3D), L.ptr(scales), L.ptr(rotations),
                       L.ptr(cov3D_precomp), L.ptr(sh), L.ptr(sh_rest), L.ptr(colors_precomp), L.ptr(opac), L.ptr(viewm), L.ptr(projm), L.ptr(campos), L.ptr(bg),
                       W, H, float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier), _ALLOC_CB, None, hint,
                       L.ptr(out), L.ptr(radii), ctypes.byref(state), L.stream(), side_handle)
            except RuntimeError:
                if "error" in holder:
                    raise holder["error"]
                raise
            finally:
                _ALLOC_TLS.holder = None
        if side.enabled and not low:
            # blocks the colour kernel used on the side stream are freed by the caller's stream: tell the allocator
            # (the library's own low-priority stream is joined inside the call: stream order on the caller's stream covers it)
            for t in holder.get(L.GSPL_BUF_GEOMETRY, []):
                t.record_stream(side.stream)
        _LAST_ISECTS[key] = int(state.n_isects)
        SPECULATION["frames"] += 1
        if hint == 0:
            SPECULATION["cold"] += 1
        elif int(state.n_isects) > hint:
            SPECULATION["misses"] += 1
        holder.pop(L.GSPL_BUF_BINNING, None)           # scratch of the count half and of the tile sort: not needed again
        holder.pop(L.GSPL_BUF_LISTS_WORK, None)
        ctx.save_for_backward(means3D, scales, rotations, sh, opac, viewm, projm, campos, bg, radii, sh_rest)
        ctx.holder, ctx.state = holder, state
        ctx.cfg = (H, W, int(s.sh_degree), n_coeffs, float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier), colors_precomp is not None,
                   cov3D_precomp is not None, opacities.shape)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        ctx.means2D_ref = means2D
        if KEEP_LAST_RASTER:
            global LAST_RASTER
            geom, lists = holder[L.GSPL_BUF_GEOMETRY][0], holder.get(L.GSPL_BUF_LISTS, [None])[-1]
            img = holder[L.GSPL_BUF_IMAGE][0]
            nI = int(state.n_isects)
            LAST_RASTER = dict(mode=L.GSPL_MODE_INRIA, width=W, height=H, means2d=_view(geom, state.means2d, (N, 2), torch.float32),
                               conics=_view(geom, state.conics, (N, 3), torch.float32), opacities=opac,
                               colors=_view(geom, state.colors, (N, 3), torch.float32),
                               flatten_ids=(lists[:4 * nI].view(torch.int32) if lists is not None else torch.empty(0, dtype=torch.int32, device=dev)),
                               offsets=_view(img, state.offsets, (tile_w * tile_h,), torch.int32), radii=radii,
                               depths=_view(geom, state.depths, (N,), torch.float32))
        return out, radii

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_out, _v_radii):
        import ctypes
        means3D, scales, rotations, sh, opac, viewm, projm, campos, bg, radii, sh_rest = ctx.saved_tensors
        H, W, degree, n_coeffs, tanfovx, tanfovy, scale_modifier, has_precomp_colors, use_cov, opac_shape = ctx.cfg
        N = means3D.shape[0]
        dev = means3D.device
        v_out = _grad_or_zeros(v_out, (3, H, W), dev)
        E = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        packed = E(N, 9)
        hit = torch.empty((N,), dtype=torch.uint8, device=dev) if TRACK_HIT_PIXELS else None
        v_means, v_ndc, v_opac = E(N, 3), E(N, 3), E(N)
        v_scales = None if use_cov else E(N, 3)
        v_quats = None if use_cov else E(N, 4)
        v_cov = E(N, 6) if use_cov else None
        v_sh = None if has_precomp_colors else torch.empty_like(sh)
        v_sh_rest = None if sh_rest is None else torch.empty_like(sh_rest)
        v_cp = E(N, 3) if has_precomp_colors else None
        if N > 0:
            with torch.cuda.device(dev):
                L.call("gspl_rasterize_inria_bwd", degree, n_coeffs, L.ptr(means3D), L.ptr(scales), L.ptr(rotations), L.ptr(sh), L.ptr(sh_rest),
                       L.ptr(opac), L.ptr(viewm), L.ptr(projm), L.ptr(campos), L.ptr(bg), tanfovx, tanfovy, scale_modifier, L.ptr(radii),
                       ctypes.byref(ctx.state), L.ptr(v_out), L.ptr(packed), L.ptr(hit), L.ptr(v_means), L.ptr(v_ndc), L.ptr(v_sh), L.ptr(v_sh_rest),
                       L.ptr(v_cp), L.ptr(v_opac), L.ptr(v_scales), L.ptr(v_quats), L.ptr(v_cov), L.stream())
            if hit is not None and ctx.means2D_ref is not None:
                ctx.means2D_ref.has_hit_any_pixels = hit.view(torch.bool)
        else:
            for t in (v_means, v_ndc, v_opac):
                t.zero_()
        ctx.holder = None
        return v_means, v_ndc, v_sh, v_cp, v_opac.reshape(opac_shape), v_scales, v_quats, v_cov, None, v_sh_rest


class GaussianRasterizer(torch.nn.Module):
    """Drop-in for `diff_gaussian_rasterization.GaussianRasterizer` as the reference uses it
    (internal/renderers/vanilla_renderer.py:79,111-120): returns (color [3,H,W], radii [N] i32);
    `means2D.grad` receives the screen-space gradient in the Inria (NDC-scaled) units."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, shs_rest=None):
        """`shs_rest` (extension; also accepted as `shs=(shs_dc, shs_rest)`): the model's two SH parameters as they are stored."""
        if isinstance(shs, (tuple, list)):
            shs, shs_rest = shs
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        # one C-ABI call per direction (csrc/fused.hip) unless GSPL_FUSED_INRIA=0 selects the stage-by-stage orchestration
        fn = _InriaFusedFn if FUSED_INRIA else _InriaRasterizeFn
        if shs_rest is not None and shs_rest.shape[1] == 0:
            shs_rest = None
        return fn.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, self.raster_settings, shs_rest)


# =============================================================================================
# radix sort of the binning stage (exported for the parity tests)
# =============================================================================================
def _radix_sort(fn: str, keys: Tensor, vals, begin_bit: int, end_bit: int):
    import ctypes
    lib = L.lib()
    n = keys.numel()
    k0, k1 = keys.clone(), torch.empty_like(keys)
    v0 = v1 = None
    if vals is not None:
        v0, v1 = vals.clone(), torch.empty_like(vals)
    ws_bytes = lib.gspl_radix_sort_workspace_bytes(n, keys.element_size(), begin_bit, end_bit)
    if ws_bytes == 0:
        raise RuntimeError("gspl_radix_sort_workspace_bytes: unsupported size or bit range")
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=keys.device)
    res = ctypes.c_int(-1)
    with torch.cuda.device(keys.device):
        if vals is not None:
            L.call(fn, n, L.ptr(k0), L.ptr(k1), L.ptr(v0), L.ptr(v1), begin_bit, end_bit, ctypes.byref(res), L.ptr(ws), ws_bytes, L.stream())
        else:
            L.call(fn, n, L.ptr(k0), L.ptr(k1), begin_bit, end_bit, ctypes.byref(res), L.ptr(ws), ws_bytes, L.stream())
    return ((k0, v0), (k1, v1))[res.value]


def radix_sort_pairs(keys: Tensor, vals: Tensor, begin_bit: int = 0, end_bit: int = 32):
    """Stable ascending sort of (u32 key, u32 value) pairs on key bits [begin_bit, end_bit): the depth sort of
    `bin_gaussians`, exposed for the parity tests.  keys/vals: int32 or uint32 tensors holding the bit patterns."""
    assert keys.is_cuda and keys.dtype in (torch.int32, torch.uint32) and vals.dtype in (torch.int32, torch.uint32)
    return _radix_sort("gspl_radix_sort_pairs_u32", keys.contiguous(), vals.contiguous(), begin_bit, end_bit)


def radix_sort_keys64(keys: Tensor, begin_bit: int, end_bit: int) -> Tensor:
    """Stable ascending sort of u64 records on key bits [begin_bit, end_bit) (at most 32 bits): the tile sort of
    `bin_gaussians`.  keys: int64 tensor holding the bit patterns."""
    assert keys.is_cuda and keys.dtype == torch.int64
    return _radix_sort("gspl_radix_sort_keys_u64", keys.contiguous(), None, begin_bit, end_bit)[0]


# =============================================================================================
# simple_knn  
# =============================================================================================
def distCUDA2(points: Tensor) -> Tensor:
    """Drop-in for `simple_knn._C.distCUDA2` (reference call site: internal/models/vanilla_gaussian.py:122-124):
    points [N,3] on the GPU -> [N] mean squared distance to the three nearest other points (fp32)."""
    lib = L.lib()
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be on the GPU (the reference calls it with `.cuda()`)")
    pts = _f32c(points.detach()).reshape(-1, 3)
    N = pts.shape[0]
    out = torch.empty((N,), dtype=torch.float32, device=pts.device)
    if N == 0:
        return out
    ws_bytes = lib.gspl_knn_workspace_bytes(N)
    if ws_bytes == 0:
        raise RuntimeError("gspl_knn_workspace_bytes failed: " + lib.gspl_last_error().decode())
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        L.call("gspl_knn3_mean_dist2", N, L.ptr(pts), L.ptr(out), L.ptr(ws), ws_bytes, L.stream())
    return out


# =============================================================================================
# fused photometric loss terms  
# =============================================================================================
class _L1SSIMFn(torch.autograd.Function):
    """(mean |x - y|, mean SSIM(x, y)); gradients flow to the first image only (the second is the ground truth),
    as in the `fused_ssim` package the reference can opt into (vanilla_metrics.py:35-39)."""

    @staticmethod
    def forward(ctx, img1, img2, train):
        lib = L.lib()
        if not img1.is_cuda or not img2.is_cuda:
            raise RuntimeError("l1_ssim: images must be on the GPU")
        if img1.shape != img2.shape or img1.dim() < 2:
            raise ValueError(f"l1_ssim: shapes {tuple(img1.shape)} vs {tuple(img2.shape)}")
        x, y = _f32c(img1), _f32c(img2)
        H, W = int(x.shape[-2]), int(x.shape[-1])
        planes = x.numel() // (H * W) if H * W > 0 else 0
        if planes == 0:
            raise ValueError("l1_ssim: empty image")
        dev = x.device
        means = torch.empty((2,), dtype=torch.float32, device=dev)
        keep = bool(train) and img1.requires_grad
        maps = torch.empty((3, planes, H, W), dtype=torch.float32, device=dev) if keep else None
        ws_bytes = lib.gspl_loss_workspace_bytes(planes, H, W)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.call("gspl_loss_l1_ssim_fwd", planes, H, W, L.ptr(x), L.ptr(y), L.ptr(means),
                   L.ptr(maps[0]) if keep else None, L.ptr(maps[1]) if keep else None, L.ptr(maps[2]) if keep else None,
                   L.ptr(ws), ws_bytes, L.stream())
        ctx.save_for_backward(x, y, maps)
        ctx.dims = (planes, H, W, img1.shape)
        return means[0], means[1]

    @staticmethod
    def backward(ctx, v_l1, v_ssim):
        x, y, maps = ctx.saved_tensors
        planes, H, W, shape = ctx.dims
        if maps is None and v_ssim is not None:
            raise RuntimeError("l1_ssim: backward through SSIM needs train=True in the forward")
        v_img = torch.empty_like(x)
        v_l1 = _f32c(v_l1) if v_l1 is not None else None
        v_ssim = _f32c(v_ssim) if v_ssim is not None else None
        use_ssim = maps is not None and v_ssim is not None
        with torch.cuda.device(x.device):
            L.call("gspl_loss_l1_ssim_bwd", planes, H, W, L.ptr(x), L.ptr(y),
                   L.ptr(maps[0]) if use_ssim else None, L.ptr(maps[1]) if use_ssim else None, L.ptr(maps[2]) if use_ssim else None,
                   L.ptr(v_l1), L.ptr(v_ssim), 1.0 if v_l1 is not None else 0.0, 1.0 if use_ssim else 0.0, L.ptr(v_img), L.stream())
        return v_img.reshape(shape), None, None


def l1_ssim(img1: Tensor, img2: Tensor, train: bool = True):
    """(mean |img1 - img2|, mean SSIM) of [..., H, W] images in one pass over the pixels; differentiable w.r.t. img1."""
    return _L1SSIMFn.apply(img1, img2, train)


def fused_ssim(img1: Tensor, img2: Tensor, padding: str = "same", train: bool = True) -> Tensor:
    """Drop-in for `fused_ssim.fused_ssim` as the reference calls it (internal/metrics/vanilla_metrics.py:36-38,
    taming_3dgs_density_controller.py:405): img [B,C,H,W] -> mean SSIM, gradient to img1."""
    if padding != "same":
        raise NotImplementedError("fused_ssim: only padding='same' (the reference's call sites use the default)")
    return _L1SSIMFn.apply(img1, img2, train)[1]


class _PhotometricLossFn(torch.autograd.Function):
    """loss = w_l1 * mean|x - y| + w_ssim * (1 - mean SSIM) as ONE forward (tile kernel + reduction that also forms the
    weighted sum) and ONE backward kernel: no element-wise torch kernels between the two."""

    @staticmethod
    def forward(ctx, img1, img2, w_l1, w_ssim):
        lib = L.lib()
        if not img1.is_cuda or not img2.is_cuda:
            raise RuntimeError("photometric_loss: images must be on the GPU")
        if img1.shape != img2.shape or img1.dim() < 2:
            raise ValueError(f"photometric_loss: shapes {tuple(img1.shape)} vs {tuple(img2.shape)}")
        x, y = _f32c(img1), _f32c(img2)
        H, W = int(x.shape[-2]), int(x.shape[-1])
        planes = x.numel() // (H * W) if H * W > 0 else 0
        if planes == 0:
            raise ValueError("photometric_loss: empty image")
        dev = x.device
        means = torch.empty((3,), dtype=torch.float32, device=dev)          # (L1, SSIM, weighted loss)
        keep = img1.requires_grad
        maps = torch.empty((3, planes, H, W), dtype=torch.float32, device=dev) if keep else None
        ws_bytes = lib.gspl_loss_workspace_bytes(planes, H, W)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.call("gspl_loss_photometric_fwd", planes, H, W, L.ptr(x), L.ptr(y), float(w_l1), float(w_ssim), L.ptr(means),
                   L.ptr(maps[0]) if keep else None, L.ptr(maps[1]) if keep else None, L.ptr(maps[2]) if keep else None,
                   L.ptr(ws), ws_bytes, L.stream())
        ctx.save_for_backward(x, y, maps)
        ctx.cfg = (planes, H, W, img1.shape, float(w_l1), float(w_ssim))
        ctx.terms = means            # (L1, SSIM) of the last call, for logging without another pass
        return means[2]

    @staticmethod
    def backward(ctx, v_loss):
        x, y, maps = ctx.saved_tensors
        planes, H, W, shape, w_l1, w_ssim = ctx.cfg
        v_img = torch.empty_like(x)
        v = _f32c(v_loss)
        with torch.cuda.device(x.device):
            # d loss = w_l1 * d L1 - w_ssim * d SSIM, both scaled by the same upstream scalar
            L.call("gspl_loss_l1_ssim_bwd", planes, H, W, L.ptr(x), L.ptr(y),
                   L.ptr(maps[0]), L.ptr(maps[1]), L.ptr(maps[2]), L.ptr(v), L.ptr(v), w_l1, -w_ssim, L.ptr(v_img), L.stream())
        return v_img.reshape(shape), None, None, None


def photometric_loss(image: Tensor, gt_image: Tensor, lambda_dssim: float = 0.2) -> Tensor:
    """(1 - lambda) * L1 + lambda * (1 - SSIM): the reference's training loss (vanilla_metrics.py:66-68), one forward and one
    backward kernel."""
    return _PhotometricLossFn.apply(image, gt_image, 1.0 - lambda_dssim, lambda_dssim)
