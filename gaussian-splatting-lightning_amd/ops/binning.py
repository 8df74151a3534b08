"""Tile binning: the 64-bit keyed `isect_tiles` / `isect_offset_encode` and the list-only `bin_gaussians` with its speculative
emission and lists whose length stays on the device (`LazyLists`)."""
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


class _PendingBins:
    """Binning in flight: the count/depth-sort half has been launched and the number of intersections is on its
    way to a pinned host word; `bin_gaussians_end` waits for it and launches the emit/sort half."""
    __slots__ = ("N", "mode", "means2d", "radii", "cull_c", "cull_o", "order", "cum", "spans", "offsets", "tile_w", "tile_h",
                 "block_width", "host_count", "event", "dev", "capacity", "ws2", "ws2_bytes", "big_list", "depths", "count", "offsets_buf")



MAX_ISECTS = 2 ** 30 - 1      # RADIX_MAX_ITEMS of csrc/gspl_sort.h: the list positions and the sort's workgroup spans are 32-bit
# How the guesses fared (bench.py reports the miss rate): frames binned, frames without a guess (first of a size: the host waits),
# frames whose guess was too low (emission, sort and — in the fused call — compositing are repeated).


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
        p.host_count = S.pinned_words.pop() if S.pinned_words else torch.empty((2,), dtype=torch.int64).pin_memory()
        L.call("gspl_bin_count", N, mode, L.ptr(p.means2d), L.ptr(p.radii), L.ptr(depths), L.ptr(p.cull_c), L.ptr(p.cull_o),
               block_width, p.tile_w, p.tile_h, L.ptr(p.order), L.ptr(p.cum), L.ptr(p.big_list), L.ptr(p.spans), p.host_count.data_ptr(),
               L.ptr(ws), ws_bytes, L.stream())
        p.event = _take_event(dev)
        p.event.record()
        # Speculative emission: the emit kernel's grid depends on N only, so it is launched NOW with room for a guess of
        # the list length (the last frame's, plus a margin) and runs while the host waits for the real number; a guess
        # that turns out too low costs one repeated emission in `bin_gaussians_end`.
        guess = S.capacity.hint((dev.index, p.tile_w, p.tile_h), N)      # decayed running maximum per splat x N x margin (ops._state.ListCapacity)
        if S.speculative_emit and guess > 0:
            p.capacity = min(guess, MAX_ISECTS)
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
    S.events[p.dev.index].append(p.event)
    n_isects = int(p.host_count[0])
    S.pinned_words.append(p.host_count)
    if n_isects > MAX_ISECTS:
        raise RuntimeError(f"{n_isects} (tile, Gaussian) intersections in one frame: the per-tile lists of this library hold at most "
                           f"2^30-1 = {MAX_ISECTS} entries (fewer / smaller Gaussians, a larger tile size or a lower resolution)")
    S.last_isects[(p.dev.index, p.tile_w, p.tile_h)] = n_isects
    S.capacity.observe((p.dev.index, p.tile_w, p.tile_h), p.N, n_isects)
    S.speculation["frames"] += 1
    if p.capacity == 0:
        S.speculation["cold"] += 1
    elif n_isects > p.capacity:
        S.speculation["misses"] += 1
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
    if N > 0 and p.ws2 is not None and S.device_side_list_length:
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
