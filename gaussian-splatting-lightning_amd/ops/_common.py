"""Helpers shared by the operator modules: device guards, tensor normalisation, the colour (side) stream and the
book-keeping of parameter updates in flight."""
from __future__ import annotations

import os
from typing import NamedTuple, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .. import _lib as L
from ._state import STATE as S

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
    if S.pending_updates and (t.dtype != torch.float32 or not t.is_contiguous()):
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
            raw = (L.lib().gspl_low_priority_stream() or 0) if S.side_low_priority else 0
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


def join_pending_updates(device=None):
    """Make the current stream of `device` (default: every device with an entry) wait for ALL parameter updates in flight
    (`FusedAdam(deferred=...)`).  `_await_updates` recognises a parameter by its data pointer, which covers the kernels of this
    package reading it in place; a TORCH read — `torch.cat` inside `get_features`, a dtype / layout copy, user code in
    `on_train_batch_end` — produces a new tensor that no pointer table can tie to the update, so every place of this package that
    reads a possibly-deferred parameter through torch calls this first (renderers/renderer.py: `model_sh_pair`, `_f32c` above)."""
    if not S.pending_updates:
        return
    seen = set()
    for done, _raw, dev in list(S.pending_updates.values()):
        if id(done) in seen:
            continue
        seen.add(id(done))
        if device is not None and torch.device(dev) != torch.device(device):
            continue
        torch.cuda.current_stream(dev).wait_event(done)


def _await_updates(*tensors, on_raw_stream=None):
    """Make the current stream wait for the in-flight updates of `tensors` (no-op for updates launched on `on_raw_stream`, which
    stream order already covers)."""
    if not S.pending_updates:
        return
    for t in tensors:
        if t is None:
            continue
        ent = S.pending_updates.get(t.data_ptr())
        if ent is not None and ent[1] != on_raw_stream:
            torch.cuda.current_stream(t.device).wait_event(ent[0])


def _take_event(dev):
    pool = S.events.setdefault(dev.index, [])
    return pool.pop() if pool else S.new_event()


def set_deterministic(on: bool = True) -> bool:
    """Bit-reproducible gradients from the compositing backward (`gspl_set_deterministic`: per-splat rows added in list order instead
    of by atomics in dispatch order) — for tests that compare two runs and for debugging; slower (three extra passes over the list
    entries).  Process-wide; returns the previous setting."""
    return bool(L.lib().gspl_set_deterministic(1 if on else 0))
