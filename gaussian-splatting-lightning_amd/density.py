"""Densification statistics of the density controller, fused.

`update_densification_stats` is the in-place equivalent of the PyTorch lines of
`VanillaDensityControllerImpl.update_states` + `_add_densification_stats`
(internal/density_controllers/vanilla_density_controller.py:101-123) in one kernel launch; `HipDensityStatsMixin`
puts it behind the same method name, so a controller of the reference picks it up by inheritance:

    class HipVanillaDensityControllerImpl(HipDensityStatsMixin, VanillaDensityControllerImpl):
        pass
"""
from typing import Optional, Union

import torch
from torch import Tensor

from . import _lib as L


@torch.no_grad()
def update_densification_stats(grad: Tensor, visibility_filter: Optional[Tensor], radii: Optional[Tensor],
                               xyz_gradient_accum: Tensor, denom: Tensor, max_radii2D: Optional[Tensor],
                               scale: Union[None, float, int, Tensor] = None) -> None:
    """max_radii2D[v] = max(max_radii2D[v], radii[v]); xyz_gradient_accum[v] += |grad[v, :2] * scale|; denom[v] += 1
    with v = visibility_filter (None: radii > 0).  All state tensors are updated in place.
    grad [N, >=2] f32; radii [N] int32/float32; state tensors f32 with N elements; scale: None, a number, or a device
    tensor of 1 or 2 elements (the renderers' `viewspace_points_grad_scale`) — read on the device, no host sync."""
    if not grad.is_cuda:
        raise RuntimeError("update_densification_stats runs on the GPU only; there is no CPU fallback")
    N = grad.shape[0]
    if N == 0:
        return
    g = grad if (grad.dtype == torch.float32 and grad.is_contiguous()) else grad.float().contiguous()
    g = g.reshape(N, -1)
    for t, name in ((xyz_gradient_accum, "xyz_gradient_accum"), (denom, "denom"), (max_radii2D, "max_radii2D")):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != N):
            raise RuntimeError(f"{name}: a contiguous float32 tensor with one element per Gaussian is needed")
    vis = None
    if visibility_filter is not None:
        vis = visibility_filter.reshape(-1)
        vis = (vis if vis.dtype in (torch.bool, torch.uint8) else vis != 0).contiguous().view(torch.uint8)
    r_i = r_f = None
    if radii is not None:
        r = radii.reshape(-1)
        if r.dtype == torch.int32:
            r_i = r.contiguous()
        else:
            r_f = r.float().contiguous()
    sx = sy = 1.0
    s_dev = None
    if isinstance(scale, Tensor):
        s_dev = scale.detach().to(device=g.device, dtype=torch.float32).reshape(-1)
        s_dev = (s_dev.expand(2) if s_dev.numel() == 1 else s_dev[:2]).contiguous()
    elif scale is not None:
        sx = sy = float(scale)
    with torch.cuda.device(g.device):
        L.call("gspl_densify_stats", N, L.ptr(g), g.shape[1], sx, sy, L.ptr(s_dev), L.ptr(vis), L.ptr(r_i), L.ptr(r_f),
               L.ptr(xyz_gradient_accum), L.ptr(denom), L.ptr(max_radii2D), L.stream())


@torch.no_grad()
def update_densification_stats_views(grads, visibility_filters, radii, xyz_gradient_accum: Tensor, denom: Tensor,
                                     max_radii2D: Optional[Tensor], scale: Union[None, float, int, Tensor] = None) -> None:
    """`update_densification_stats` for the cameras of ONE Gaussian-sharded step — the loop of
    `DistributedVanillaDensityControllerImpl.update_states` (distributed_vanilla_density_controller.py:22-47) — in one launch
    instead of one per camera (eight at W = 8): grads / visibility_filters / radii are lists with one entry per camera
    (`projection_results_list[i][1].grad`, `visible_mask_list[i]`, `projection_results_list[i][0]`).  The views are applied in list
    order per Gaussian: bit-identical to the sequential calls.  More than 16 views, or inputs the kernel does not take (float radii,
    non-contiguous gradients), fall back to the per-view launches."""
    import ctypes
    n_views = len(grads)
    if n_views == 0:
        return
    ok = 1 <= n_views <= 16 and all(isinstance(g, Tensor) and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.dim() == 2 for g in grads)
    ok = ok and all(r is not None and r.dtype == torch.int32 and r.is_contiguous() and r.numel() == grads[0].shape[0] for r in radii)
    ok = ok and len({tuple(g.shape) for g in grads}) == 1 and (scale is None or isinstance(scale, (Tensor, float, int)))
    if not ok:
        for g, v, r in zip(grads, visibility_filters, radii):
            update_densification_stats(g, v, r, xyz_gradient_accum, denom, max_radii2D, scale=scale)
        return
    N, stride = grads[0].shape
    if N == 0:
        return
    for t, name in ((xyz_gradient_accum, "xyz_gradient_accum"), (denom, "denom"), (max_radii2D, "max_radii2D")):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != N):
            raise RuntimeError(f"{name}: a contiguous float32 tensor with one element per Gaussian is needed")
    vis = []
    for v in visibility_filters:
        if v is None:
            vis.append(None)
            continue
        v = v.reshape(-1)
        vis.append((v if v.dtype in (torch.bool, torch.uint8) else v != 0).contiguous().view(torch.uint8))
    sx = sy = 1.0
    s_dev = None
    if isinstance(scale, Tensor):
        s_dev = scale.detach().to(device=grads[0].device, dtype=torch.float32).reshape(-1)
        s_dev = (s_dev.expand(2) if s_dev.numel() == 1 else s_dev[:2]).contiguous()
    elif scale is not None:
        sx = sy = float(scale)
    PA = ctypes.c_void_p * n_views
    addr = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(grads[0].device):
        L.call("gspl_densify_stats_views", N, n_views, PA(*[addr(g) for g in grads]), stride, sx, sy, L.ptr(s_dev),
               PA(*[addr(v) for v in vis]) if any(v is not None for v in vis) else None, PA(*[addr(r.reshape(-1)) for r in radii]),
               L.ptr(xyz_gradient_accum), L.ptr(denom), L.ptr(max_radii2D), L.stream())


class StatsRequest:
    """A frame's statistics handed to that frame's backward: `applied` turns True once the fused Inria backward that owns `radii`
    has run the update of `update_densification_stats(viewspace.grad, None, radii, ...)` inside its last kernel."""
    __slots__ = ("radii", "radii_ptr", "n", "accum", "denom", "max_radii", "applied")

    def __init__(self, radii, accum, denom, max_radii):
        self.radii = radii      # kept: while the request is pending no later frame's radii can be allocated at this address
        self.radii_ptr, self.n = radii.data_ptr(), radii.numel()
        self.accum, self.denom, self.max_radii = accum, denom, max_radii
        self.applied = False


def request_stats_in_backward(radii: Tensor, xyz_gradient_accum: Tensor, denom: Tensor,
                              max_radii2D: Optional[Tensor]) -> Optional[StatsRequest]:
    """Ask the backward of the frame that returned `radii` (ops.GaussianRasterizer, the fused call) to apply this frame's
    statistics itself: for every Gaussian with radii > 0, max_radii2D = max(max_radii2D, radii), xyz_gradient_accum += |the
    screen-space gradient the backward hands to `means2D` (first two columns)|, denom += 1 — what
    `update_densification_stats(means2D.grad, None, radii, ...)` does after the backward, with the same arithmetic, minus its launch
    (the preprocess-backward kernel has the row's gradient in registers).  Call it between the render and `loss.backward()`;
    afterwards `request.applied` says whether that backward ran and did it (it does so once) — if not, call
    `update_densification_stats` as before.  None: not possible here (switched off, not the fused call's radii, unfit buffers).
    One request is pending at a time; a newer one replaces it, `withdraw_stats_request` drops it.
    Single-consumer assumption: the backward adds the gradient the RASTERIZER hands to `means2D`.  The reference reads the leaf's
    accumulated `viewspace_points.grad` (vanilla_density_controller.py:111-117); the two are the same thing as long as nothing else
    back-propagates into that tensor — true for the reference's loop and renderers (the tensor exists only to carry this gradient).  A
    loss that also differentiates `viewspace_points` must not make this request (GSPL_STATS_IN_BACKWARD=0, or call
    `update_densification_stats` itself)."""
    from .ops._state import STATE
    if not (STATE.stats_in_backward and STATE.fused_inria):
        return None
    if not (isinstance(radii, Tensor) and radii.is_cuda and radii.dtype == torch.int32 and radii.is_contiguous() and radii.dim() == 1):
        return None
    if not getattr(radii, "_gspl_fused_inria", False):       # set by the fused call on the radii it returns
        return None
    n = radii.numel()
    for t in (xyz_gradient_accum, denom, max_radii2D):
        if t is None:
            continue
        if not (t.is_cuda and t.device == radii.device and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
            return None
    if n == 0 or xyz_gradient_accum is None or denom is None:
        return None
    req = StatsRequest(radii, xyz_gradient_accum, denom, max_radii2D)
    STATE.backward_stats = req
    return req


def withdraw_stats_request(req: Optional[StatsRequest]) -> None:
    from .ops._state import STATE
    if req is not None and STATE.backward_stats is req:
        STATE.backward_stats = None


class HipDensityStatsMixin:
    """`update_states` of the reference's density controllers on the fused kernel (same reads of `outputs`, same state
    buffers: `max_radii2D`, `xyz_gradient_accum`, `denom`; `config.absgrad` selects `.absgrad`).  With the package's vanilla renderer
    in front (its `visibility_filter` is `radii > 0`, no gradient scale, no absgrad) `before_backward` hands the buffers to the
    frame's backward, which applies the update itself; `update_states` then has nothing left to launch."""

    def before_backward(self, outputs, *args, **kwargs):
        super().before_backward(outputs, *args, **kwargs)
        withdraw_stats_request(getattr(self, "_stats_request", None))
        self._stats_request = None
        # global_step is the reference's fourth positional argument after `outputs` (density_controller.py:13)
        global_step = kwargs.get("global_step", args[3] if len(args) > 3 else None)
        if global_step is not None and global_step >= self.config.densify_until_iter:
            return                                        # after_backward will not update the statistics either
        if getattr(self.config, "absgrad", False) is True or outputs.get("viewspace_points_grad_scale", None) is not None:
            return
        radii, vis = outputs.get("radii"), outputs.get("visibility_filter")
        if not getattr(vis, "_gspl_radii_positive", False):   # the package's vanilla renderer marks its `radii > 0`
            return
        self._stats_request = request_stats_in_backward(radii, self.xyz_gradient_accum, self.denom, self.max_radii2D)

    def update_states(self, outputs):
        req, self._stats_request = getattr(self, "_stats_request", None), None
        withdraw_stats_request(req)
        if req is not None and req.applied and isinstance(outputs.get("radii"), Tensor) and outputs["radii"].data_ptr() == req.radii_ptr:
            return                                        # this frame's backward has applied them
        vp = outputs["viewspace_points"]
        grad = vp.absgrad if getattr(self.config, "absgrad", False) is True else vp.grad
        update_densification_stats(grad, outputs["visibility_filter"], outputs["radii"], self.xyz_gradient_accum, self.denom,
                                   self.max_radii2D, scale=outputs.get("viewspace_points_grad_scale", None))


class HipDistributedDensityStatsMixin:
    """`update_states` of the reference's `DistributedVanillaDensityControllerImpl`
    (internal/density_controllers/distributed_vanilla_density_controller.py:22-47: a Python loop over the step's cameras, a dozen torch
    launches each) on ONE launch for all cameras of the step (`update_densification_stats_views`); same reads of `outputs` —
    `cameras`, `projection_results_list`, `visible_mask_list`, `xys_grad_scale_required`, `config.absgrad` — same buffers:

        class HipDistributedVanillaDensityControllerImpl(HipDistributedDensityStatsMixin, DistributedVanillaDensityControllerImpl):
            pass

    Cameras of different sizes (a per-view gradient scale) take one launch per view."""

    def update_states(self, outputs):
        cameras, results, masks = outputs["cameras"], outputs["projection_results_list"], outputs["visible_mask_list"]
        absgrad = getattr(self.config, "absgrad", False) is True
        grads = [(r[1].absgrad if absgrad else r[1].grad) for r in results]
        radii = [r[0] for r in results]
        scale, per_view = None, None
        if outputs.get("xys_grad_scale_required", False) is True:
            dev = results[0][1].device
            sizes = [(int(c.width), int(c.height)) for c in cameras]
            if len(set(sizes)) == 1:
                scale = 0.5 * torch.tensor(sizes[0], dtype=torch.float32, device=dev)
            else:
                per_view = [0.5 * torch.tensor(wh, dtype=torch.float32, device=dev) for wh in sizes]
        accum, denom = self.xyz_gradient_accum.reshape(-1), self.denom.reshape(-1)
        if per_view is None:
            update_densification_stats_views(grads, masks, radii, accum, denom, self.max_radii2D, scale=scale)
        else:
            for g, m, r, s_ in zip(grads, masks, radii, per_view):
                update_densification_stats(g, m, r, accum, denom, self.max_radii2D, scale=s_)
