"""Tile compositing (`rasterize_to_pixels`, `rasterize_gaussians`) and the per-splat score passes."""
from __future__ import annotations

import os
from typing import NamedTuple, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .. import _lib as L
from ._state import STATE as S
from ._common import (_SUPPORTED_D, _packed_row_stride, _guarded, _f32c, _rows, _raw_ptr, _grad_or_zeros, _side_stream, colour_stream,
                      join_pending_updates, _await_updates, _take_event)
from .binning import LazyLists, bin_gaussians

# =============================================================================================
# compositing
# =============================================================================================
# When True, the compositing backward also reports which splats some pixel actually composited and attaches the mask as
# `has_hit_any_pixels` to the caller's screen-space tensor (the fork-only side channel gsplat's SelectiveAdam adapter
# reads, internal/optimizers.py:39).  Off by default: it is one more byte store per (tile, splat) in the hot kernel.
# Staged binning (`bin_gaussians`): with a speculative emission in flight the tile sort is enqueued before the host has read the
# list length (False: wait for the count first, then sort — the round-1 order; kept for A/B runs and the tests of both orders).
# Introspection for bench.py / tools: with S.keep_last_raster set, the last compositing forward leaves its per-splat inputs and
# tile lists in S.last_raster (a dict of tensors; nothing is copied).


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
        if S.keep_last_raster:
            S.last_raster = dict(mode=mode, width=width, height=height, means2d=means2d, conics=conics, opacities=opacities,
                               colors=colors, flatten_ids=flatten_ids, offsets=offsets, last_ids=last_ids)
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
            hit = torch.zeros((N,), dtype=torch.uint8, device=dev) if S.track_hit_pixels else None
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
