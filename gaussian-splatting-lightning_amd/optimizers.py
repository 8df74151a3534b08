"""Fused optimizers for the per-Gaussian parameter tensors.

`SelectiveAdam` mirrors gsplat's optimizer of that name as the reference wraps it (internal/optimizers.py:26-58,
`configs/gsplat_v1-accel_more.yaml`): Adam without bias correction whose `step(visibility)` only touches the rows of
visible Gaussians (parameters AND moments of the others stay as they are), all parameter groups in one HIP launch.
`FusedAdam` is the unmasked, bias-corrected update of `torch.optim.Adam` (the reference's default,
internal/models/vanilla_gaussian.py:266-300) through the same kernel.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch

from . import _lib as L


class _FusedAdamBase(torch.optim.Optimizer):
    _bias_correction = True

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False, maximize: bool = False, **unsupported):
        """The keyword arguments of `torch.optim.Adam` are accepted so that configurations written for it construct this
        class; the ones the kernel does not implement must keep their default values."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        if weight_decay != 0.0 or amsgrad or maximize:
            raise NotImplementedError("fused Adam: weight_decay / amsgrad / maximize are not implemented (the reference never sets them)")
        for k, v in unsupported.items():      # foreach / capturable / differentiable / fused: execution hints of torch.optim.Adam
            if k not in ("foreach", "capturable", "differentiable", "fused"):
                raise TypeError(f"fused Adam: unexpected argument {k!r}")
            if k in ("capturable", "differentiable") and v:
                raise NotImplementedError(f"fused Adam: {k}=True is not implemented")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def _prepare(self, p, group):
        """State of one parameter with its step counter advanced (once per optimizer step)."""
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("fused Adam: parameters must be contiguous fp32 tensors on the GPU")
        if p.grad.is_sparse:
            raise RuntimeError("fused Adam: sparse gradients are not supported")
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        # a state loaded from a `torch.optim.Adam` checkpoint keeps `step` as a tensor: a python int from here on
        st["step"] = int(st["step"]) + 1
        return st

    def _run(self, batches, visibility: Optional[torch.Tensor]):
        """batches: {(device, rows, b1, b2, eps, step): [(param, grad, exp_avg, exp_avg_sq, lr, row_elems), ...]} -> launches."""
        for (dev, N, b1, b2, eps, step), items in batches.items():
            vis = None
            if visibility is not None:
                if visibility.shape[0] != N:
                    raise ValueError(f"visibility has {visibility.shape[0]} rows, parameters {N}")
                vis = visibility.to(device=dev, dtype=torch.uint8).contiguous()
            bc1 = 1.0 - b1 ** step if self._bias_correction else 1.0
            bc2s = math.sqrt(1.0 - b2 ** step) if self._bias_correction else 1.0
            for i in range(0, len(items), L.GSPL_ADAM_MAX_TENSORS):
                chunk = items[i:i + L.GSPL_ADAM_MAX_TENSORS]
                table = (L.AdamTensor * len(chunk))()
                for k, (p, g, m, v, lr, row) in enumerate(chunk):
                    if m.shape != p.shape or v.shape != p.shape or not m.is_contiguous() or not v.is_contiguous() or m.dtype != torch.float32:
                        raise RuntimeError("fused Adam: exp_avg / exp_avg_sq must be contiguous fp32 tensors of the parameter's shape")
                    table[k] = L.AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lr, row)
                with torch.cuda.device(dev):
                    L.call("gspl_selective_adam", len(chunk), ctypes.cast(table, ctypes.c_void_p), N, L.ptr(vis),
                           float(b1), float(b2), float(eps), float(bc1), float(bc2s), L.stream())

    def _launch(self, visibility: Optional[torch.Tensor]):
        L.lib()
        # tensors are batched per (N, betas, eps, step count): the reference has one parameter per group, all [N, ...]
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._prepare(p, group)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                N = p.shape[0] if p.dim() > 0 else 1
                row = p.numel() // max(N, 1)
                key = (p.device, N, b1, b2, group["eps"], st["step"] if self._bias_correction else 0)
                batches.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), row))
        self._run(batches, visibility)

    # ---- the same update in ROW CHUNKS (multi-GPU: a chunk is updated as soon as the all-reduce of its gradient rows has
    #      finished, while the collectives of the following chunks are still on the wire; distributed.all_reduce_and_step) ----
    @torch.no_grad()
    def begin_chunked_step(self):
        """Advances the step counters of every parameter that has a gradient; returns {parameter: (group, state)}."""
        L.lib()
        ready = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.grad.is_contiguous():          # (row chunks are reduced and consumed in place)
                    p.grad = p.grad.contiguous()
                ready[p] = (group, self._prepare(p, group))
        return ready

    @torch.no_grad()
    def step_rows(self, ready, p, lo: int, hi: int):
        """Adam update of rows [lo, hi) of parameter `p` (after `begin_chunked_step`)."""
        if hi <= lo:
            return
        group, st = ready[p]
        b1, b2 = group["betas"]
        row = p.numel() // max(p.shape[0], 1)
        key = (p.device, hi - lo, b1, b2, group["eps"], st["step"] if self._bias_correction else 0)
        item = (p[lo:hi], p.grad[lo:hi], st["exp_avg"][lo:hi], st["exp_avg_sq"][lo:hi], float(group["lr"]), row)
        self._run({key: [item]}, None)


class SelectiveAdam(_FusedAdamBase):
    """`SelectiveAdam(params, eps, betas).step(visibility)`: gsplat's visibility-masked Adam (no bias correction)."""
    _bias_correction = False

    def __init__(self, params, eps: float = 1e-8, betas=(0.9, 0.999), lr: float = 1e-3):
        super().__init__(params, lr=lr, betas=betas, eps=eps)

    @torch.no_grad()
    def step(self, visibility: torch.Tensor):
        self._launch(visibility)


class FusedAdam(_FusedAdamBase):
    """`torch.optim.Adam` semantics (bias-corrected, every row) for [N, ...] fp32 parameters, one launch per step."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._launch(None)
        return loss


# ---- configuration classes in the shape of internal/optimizers.py (selected with e.g.
#      `--model.gaussian.optimization.optimizer gspl_amd.optimizers.HipSelectiveAdam`) ---------------------------------
try:                                                            # inside the reference repository: its own base class
    from internal.optimizers import OptimizerConfig as _OptimizerConfig     # type: ignore
except Exception:                                               # stand-alone (tests, bench): interface-identical stub
    class _OptimizerConfig:                                     # internal/optimizers.py:8-11
        def instantiate(self, params, lr: float, *args, **kwargs):
            raise NotImplementedError()

from dataclasses import dataclass, field  # noqa: E402
from typing import Tuple  # noqa: E402


@dataclass
class HipFusedAdam(_OptimizerConfig):
    """Drop-in for `internal.optimizers.Adam` (internal/optimizers.py:14-22): same update, one launch per step."""

    def instantiate(self, params, lr: float, *args, **kwargs):
        return FusedAdam(params, lr, *args, **kwargs)


@dataclass
class HipSelectiveAdam(_OptimizerConfig):
    """Drop-in for `internal.optimizers.SelectiveAdam` (internal/optimizers.py:25-58)."""
    betas: Tuple[float, float] = field(default_factory=lambda: (0.9, 0.999))

    def instantiate(self, params, lr: float, *args, **kwargs):
        params = list(params)
        for group in params:
            if isinstance(group, dict) and "lr" not in group:
                group["lr"] = lr
        from . import ops
        ops.TRACK_HIT_PIXELS = True      # the compositing backward now reports `viewspace_points.has_hit_any_pixels`

        class Adapter(SelectiveAdam):
            def on_after_backward(self, outputs, batch, gaussian_model, global_step, pl_module):
                vs = outputs["viewspace_points"]
                # the fork's rasterizer tags Gaussians that reached a pixel; without the tag, the projected-visibility mask
                self.visibility = getattr(vs, "has_hit_any_pixels", None)
                if self.visibility is None:
                    self.visibility = outputs["visibility_filter"]

            @torch.no_grad()
            def step(self, closure=None):
                loss = None
                if closure is not None:
                    with torch.enable_grad():
                        loss = closure()
                SelectiveAdam.step(self, self.visibility)
                return loss

        return Adapter(params, betas=tuple(self.betas), lr=lr, *args, **kwargs)
