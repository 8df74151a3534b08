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


class HipDensityStatsMixin:
    """`update_states` of the reference's density controllers on the fused kernel (same reads of `outputs`, same state
    buffers: `max_radii2D`, `xyz_gradient_accum`, `denom`; `config.absgrad` selects `.absgrad`)."""

    def update_states(self, outputs):
        vp = outputs["viewspace_points"]
        grad = vp.absgrad if getattr(self.config, "absgrad", False) is True else vp.grad
        update_densification_stats(grad, outputs["visibility_filter"], outputs["radii"], self.xyz_gradient_accum, self.denom,
                                   self.max_radii2D, scale=outputs.get("viewspace_points_grad_scale", None))
