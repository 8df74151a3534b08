"""The Gaussian-sharded renderer's pieces: the 48-byte visible-splat records and the step as three autograd nodes."""
from __future__ import annotations

import os
from typing import NamedTuple, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .. import _lib as L
from ._state import STATE as S
from ._common import (_SUPPORTED_D, _packed_row_stride, _guarded, _f32c, _rows, _raw_ptr, _grad_or_zeros, _side_stream, colour_stream,
                      join_pending_updates, _await_updates, _take_event)
from .projection import _ProjectFn, _SHBatchedFn
from .binning import bin_gaussians
from .compositing import _CompositeFn

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
        pool = S.pinned_ends.setdefault(C, [])
        host_ends = pool.pop() if pool else torch.empty((C,), dtype=torch.int64).pin_memory()
        ws_bytes = lib.gspl_records_workspace_bytes(C, N)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        L.call("gspl_records_pack_fwd", C, N, L.ptr(radii), L.ptr(means2d), L.ptr(depths), L.ptr(conics), L.ptr(comps), L.ptr(opac), L.ptr(rgbs),
               L.ptr(records), L.ptr(slots), L.ptr(ends), host_ends.data_ptr(), L.ptr(ws), ws_bytes, L.stream())
        ev = _take_event(dev)
        ev.record()
        ev.synchronize()                 # the split sizes of the all-to-all are needed on the host (as in the reference)
        S.events[dev.index].append(ev)
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
        ident = S.identity_slots.get((C, N, dev))
        if ident is None:
            if len(S.identity_slots) > 8:
                S.identity_slots.clear()
            ident = S.identity_slots[(C, N, dev)] = torch.arange(C * N, dtype=torch.int32, device=dev).reshape(C, N)
        slots = torch.where(vis, ident, -1)
        ctx.save_for_backward(slots)
        ctx.cfg = (C, N, has_comp, tuple(opacities.shape))
        ctx.set_materialize_grads(False)
        return rec.reshape(C * N, L.GSPL_RECORD_FLOATS)

    @staticmethod
    @_guarded(0)
    def backward(ctx, v_records):
        return _PackRecordsFn.backward(ctx, v_records, None)




def _zero_scalar(dev):
    z = S.zero_scalars.get(dev)
    if z is None:
        z = S.zero_scalars[dev] = torch.zeros((), dtype=torch.float32, device=dev)
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
            pool = S.pinned_ends.setdefault(C, [])
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
            S.events[dev.index].append(ev)
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
