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


# Workgroups (of 256 threads) per tensor of a deferred update, 0 = no limit.  The update shares the device with the next frame's
# geometry and binning kernels; a bounded grid leaves them wave slots on every CU.  Measured (round 3, S-1080p-1M, colour stream at the
# default priority): 512 / 1024 / 2048 / unlimited -> 1.251 / 1.264 / 1.255 / 1.250 ms per step: no gain from the limit, default off.
DEFERRED_BLOCKS = int(__import__("os").environ.get("GSPL_ADAM_DEFERRED_BLOCKS", "0"))


class _FusedAdamBase(torch.optim.Optimizer):
    _bias_correction = True

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False, maximize: bool = False, deferred=None, fuse_into_backward: bool = False, **unsupported):
        """The keyword arguments of `torch.optim.Adam` are accepted so that configurations written for it construct this
        class; the ones the kernel does not implement must keep their default values.

        deferred (extension, off by default): parameter-group names (the reference names its groups: "shs_rest", ...,
        vanilla_gaussian.py:266-300) and / or parameter tensors whose update is launched on the rasterizer's COLOUR stream
        (`ops.colour_stream`) instead of the caller's.  Meant for the SH coefficients: 45 of the 59 floats of a Gaussian, i.e. three
        quarters of the step's memory traffic, and the one parameter the next frame needs LAST — geometry and binning (a dozen small
        latency-bound launches that leave HBM idle) run on the caller's stream at the same time, and the colour kernel of the next
        frame follows the update in stream order.  Same kernel, same inputs: bit-identical parameters.  The contract that comes
        with it: (1) the optimizer takes the gradients of the deferred parameters (`p.grad` is None after `step()`), (2) kernels of
        this package that read a deferred parameter wait for its update by themselves (`ops._await_updates`); anything else that reads
        one between `step()` and the next render — a checkpoint written right after the step, foreign torch code — calls `join()`
        first (`state_dict()` and the next `step()` do).

        fuse_into_backward (extension, off by default): the update of a parameter the package's fused Inria rasterizer differentiates
        is applied BY ITS BACKWARD — the per-Gaussian kernels that end it update the rows they have just produced the gradient of
        (`gspl_rasterize_inria_bwd_adam`): the gradient never reaches HBM (236 B per Gaussian written and read back otherwise), the
        moments are read and written once.  Same arithmetic, bit-identical parameters for identical gradients.  What changes for the
        caller, and why it is opt-in (the default stays what internal/opt_strategies/vanilla.py:41-44 and internal/optimizers.py:14-22
        expect): (1) such a parameter is updated DURING `loss.backward()` and has no `.grad` afterwards; `step()` finds nothing
        left to do for it (it still advances nothing twice: the step counter moved when the backward claimed the update);
        (2) exactly one backward per `step()`: a second one (gradient accumulation) raises; (3) it happens only when EVERY parameter
        the rasterizer differentiates (means, scales, rotations, opacities, SH) belongs to optimizers built with this flag and is
        handed to the rasterizer as it is stored (the renderer plugins do that for the reference's model) — otherwise the backward
        writes gradients as always and `step()` applies them; (4) hyper-parameters are read when the backward runs (a scheduler that
        stepped after the previous `step()` is seen, as with torch.optim.Adam); (5) the rasterizer must be the ONLY autograd consumer
        of those parameters in a step: a second path into them — a scale or opacity regulariser, a depth loss on the means — leaves a
        gradient in `.grad` next to the update the backward has already applied (and may have been differentiated at parameters that
        had already moved); `step()` then RAISES instead of applying a second update with a second step count — train
        such a loss with the default two-kernel path; (6) on a densification step the reference DROPS the step's update (the density
        controller replaces the Parameters in `after_backward`, before `step()` runs: internal/density_controllers/
        vanilla_density_controller.py:125-178 behind gaussian_splatting.py:380-397) — here the rows were updated during the backward,
        before the surgery copies them: one more Adam update on those steps than the reference applies (a few steps in thirty
        thousand; the bench's reference-shaped loop reports both forms)."""
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
        self._deferred_names = {d for d in (deferred or ()) if isinstance(d, str)}
        self._deferred_ids = {id(d) for d in (deferred or ()) if isinstance(d, torch.Tensor)}
        self._inflight = []      # [(event, tensors kept alive for the launch, data_ptrs registered in ops.PENDING_UPDATES)]
        self._fuse_into_backward = bool(fuse_into_backward)
        self._claimed = set()    # id() of the parameters a backward has updated since the last step()
        if self._fuse_into_backward:
            import weakref
            from .ops._state import STATE
            STATE.backward_optimizers[:] = [r for r in STATE.backward_optimizers if r() is not None]
            STATE.backward_optimizers.append(weakref.ref(self))

    # ---- the update applied by the rasterizer's backward (fuse_into_backward) ------------------------------------------------------
    @property
    def fuse_into_backward(self) -> bool:
        return self._fuse_into_backward

    @fuse_into_backward.setter
    def fuse_into_backward(self, on: bool):
        """Switch the in-backward update off / on again (e.g. around backward passes that no `step()` follows)."""
        on = bool(on)
        if on and not any(r() is self for r in __import__("gspl_amd.ops._state", fromlist=["STATE"]).STATE.backward_optimizers):
            import weakref
            from .ops._state import STATE
            STATE.backward_optimizers.append(weakref.ref(self))
        self._fuse_into_backward = on

    def _owner_of(self, t: torch.Tensor):
        """(group, parameter) of the parameter stored where `t` is (same memory, same number of elements), or None."""
        ptr, n = t.data_ptr(), t.numel()
        for group in self.param_groups:
            for p in group["params"]:
                if p.data_ptr() == ptr and p.numel() == n and p.dtype == t.dtype:
                    return group, p
        return None

    def _claimable(self, group, p) -> bool:
        if not self._fuse_into_backward or p.grad is not None or not p.requires_grad:
            return False          # (a gradient is already waiting: accumulate into it and let step() apply it)
        if group.get("name") in self._deferred_names or id(p) in self._deferred_ids:
            return False
        return p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()

    def _claim(self, group, p):
        """Moments + hyper-parameters of `p` for the update the backward is about to apply; advances the step counter."""
        if id(p) in self._claimed:
            raise RuntimeError("fused Adam (fuse_into_backward): a second backward before optimizer.step() — the first one has already "
                               "updated this parameter; gradient accumulation needs the default two-kernel path")
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        m, v = st["exp_avg"], st["exp_avg_sq"]
        if m.shape != p.shape or v.shape != p.shape or not m.is_contiguous() or not v.is_contiguous() or m.dtype != torch.float32:
            raise RuntimeError("fused Adam: exp_avg / exp_avg_sq must be contiguous fp32 tensors of the parameter's shape")
        st["step"] = int(st["step"]) + 1
        b1, b2 = group["betas"]
        bc1 = 1.0 - b1 ** st["step"] if self._bias_correction else 1.0
        bc2s = math.sqrt(1.0 - b2 ** st["step"]) if self._bias_correction else 1.0
        self._claimed.add(id(p))
        return L.BwdAdamTensor(m.data_ptr(), v.data_ptr(), float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(bc1), float(bc2s))

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

    def _run(self, batches, visibility: Optional[torch.Tensor], max_blocks: int = 0):
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
                    L.call("gspl_selective_adam_limited", len(chunk), ctypes.cast(table, ctypes.c_void_p), N, L.ptr(vis),
                           float(b1), float(b2), float(eps), float(bc1), float(bc2s), int(max_blocks), L.stream())

    def _launch(self, visibility: Optional[torch.Tensor]):
        L.lib()
        self.join()
        if self._claimed:
            # a parameter the backward has ALREADY updated must not carry a gradient now: something else differentiated it too, and
            # applying that as a second update (a second step count, possibly evaluated at the moved parameter) would be silently wrong
            stray = [group.get("name", "?") for group in self.param_groups for p in group["params"] if id(p) in self._claimed and p.grad is not None]
            if stray:
                self._claimed.clear()
                raise RuntimeError("fused Adam (fuse_into_backward): parameter(s) " + ", ".join(map(str, stray)) + " were updated by the rasterizer's "
                                   "backward AND received a gradient from a second autograd path (a regulariser / an extra loss on the same "
                                   "parameters); use the default two-kernel path (fuse_into_backward=False) for such a step")
        self._claimed.clear()      # parameters a backward updated have no .grad: the loop below passes them by
        # tensors are batched per (N, betas, eps, step count): the reference has one parameter per group, all [N, ...]
        batches, later = {}, {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            defer_group = group.get("name") in self._deferred_names
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._prepare(p, group)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                N = p.shape[0] if p.dim() > 0 else 1
                row = p.numel() // max(N, 1)
                key = (p.device, N, b1, b2, group["eps"], st["step"] if self._bias_correction else 0)
                dst = later if (defer_group or id(p) in self._deferred_ids) else batches
                dst.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), row))
        self._run(batches, visibility)
        if later:
            self._run_deferred(later, visibility)

    def _run_deferred(self, later, visibility):
        """The update of the deferred parameters on the colour stream of their device (see `__init__`)."""
        from . import ops
        by_dev = {}
        for key, items in later.items():
            by_dev.setdefault(key[0], {})[key] = items
        for dev, dev_batches in by_dev.items():
            raw, side = ops.colour_stream(dev)
            if side is None:                                   # GSPL_SIDE_STREAM=0: everything on the caller's stream
                self._run(dev_batches, visibility)
                continue
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)                              # the gradients (and everything else enqueued so far) are ready
            with torch.cuda.stream(side):
                self._run(dev_batches, visibility, max_blocks=DEFERRED_BLOCKS)
                done = torch.cuda.Event()
                done.record(side)
            keep, ptrs = [visibility], []
            for items in dev_batches.values():
                for (p, g, m, v, _lr, _row) in items:
                    keep.append(g)                             # alive until the caller's stream has waited for the update: the
                    p.grad = None                              # allocator must not hand the block to later work on that stream
                    ops.PENDING_UPDATES[p.data_ptr()] = (done, raw, dev)
                    ptrs.append(p.data_ptr())
            self._inflight.append((done, keep, ptrs, dev))

    def join(self):
        """Make the current stream wait for the deferred updates in flight (no host synchronisation) and retire them."""
        if not self._inflight:
            return
        from . import ops
        for done, _keep, ptrs, dev in self._inflight:
            torch.cuda.current_stream(dev).wait_event(done)
            for ptr in ptrs:
                if ops.PENDING_UPDATES.get(ptr, (None,))[0] is done:
                    del ops.PENDING_UPDATES[ptr]
        self._inflight = []

    def state_dict(self):
        self.join()
        return super().state_dict()

    def zero_grad(self, set_to_none: bool = True):
        if not set_to_none:
            self.join()
        return super().zero_grad(set_to_none=set_to_none)

    # ---- the same update in ROW CHUNKS (multi-GPU: a chunk is updated as soon as the all-reduce of its gradient rows has
    #      finished, while the collectives of the following chunks are still on the wire; distributed.all_reduce_and_step) ----
    @torch.no_grad()
    def begin_chunked_step(self):
        """Advances the step counters of every parameter that has a gradient; returns {parameter: (group, state)}."""
        L.lib()
        self.join()
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


def claim_backward_update(named: dict):
    """Called by the fused Inria backward (ops/inria.py) with the tensors it is about to differentiate ({field of gspl_bwd_adam_plan:
    tensor or None}).  If EVERY one of them is a parameter of an optimizer built with `fuse_into_backward=True` and has no gradient
    waiting, returns the filled `BwdAdamPlan` (the step counters have then moved: the caller must apply the update); else None."""
    from .ops._state import STATE
    refs = [r() for r in STATE.backward_optimizers]
    opts = [o for o in refs if o is not None]
    if not opts:
        return None
    owners = {}
    for field, t in named.items():
        if t is None:
            continue
        found = None
        for o in opts:
            hit = o._owner_of(t)
            if hit is not None:
                found = (o, *hit)
                break
        if found is None or not found[0]._claimable(found[1], found[2]):
            return None
        owners[field] = found
    if len({id(f[2]) for f in owners.values()}) != len(owners):
        return None                                  # two inputs in one parameter's memory: not a layout the kernels update in place
    plan = L.BwdAdamPlan()
    for field, (o, group, p) in owners.items():
        setattr(plan, field, o._claim(group, p))
    return plan


class SelectiveAdam(_FusedAdamBase):
    """`SelectiveAdam(params, eps, betas).step(visibility)`: gsplat's visibility-masked Adam (no bias correction)."""
    _bias_correction = False

    def __init__(self, params, eps: float = 1e-8, betas=(0.9, 0.999), lr: float = 1e-3, deferred=None, fuse_into_backward: bool = False):
        if fuse_into_backward:
            raise NotImplementedError("SelectiveAdam: fuse_into_backward is implemented for the unmasked update (FusedAdam) only")
        super().__init__(params, lr=lr, betas=betas, eps=eps, deferred=deferred)

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
    """Drop-in for `internal.optimizers.Adam` (internal/optimizers.py:14-22): same update, one launch per step.
    `overlap_sh_update`: run the update of the "shs_rest" group under the next frame's geometry / binning (`FusedAdam(deferred=...)`,
    read its contract first); off by default.  `fuse_into_backward`: see `FusedAdam`."""
    overlap_sh_update: bool = False
    # the rasterizer's backward applies the update (FusedAdam(fuse_into_backward=True): read its contract first); off by default
    fuse_into_backward: bool = False

    def instantiate(self, params, lr: float, *args, **kwargs):
        if self.overlap_sh_update:
            kwargs.setdefault("deferred", ("shs_rest",))
        if self.fuse_into_backward:
            kwargs.setdefault("fuse_into_backward", True)
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
