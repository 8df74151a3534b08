"""
gspl_amd.ops — torch.autograd.Function wrappers over the C-ABI, mirroring the operator interfaces the
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

One module per concern: projection (+ SH colours), binning, compositing, inria (the fused rasterizer), sharded (records and the
three-node step of the Gaussian-sharded renderer), side (radix sort wrappers, distCUDA2, the photometric loss); `_common` holds the
shared helpers and `_state.STATE` every piece of mutable run-time state.  This package re-exports all of them under the names the
single-file version had, including the module-level switches (ops.FUSED_INRIA = False still works: they are properties that read
and write STATE).
"""
import sys as _sys
import types as _types

from ._state import STATE, RuntimeState
from ._common import (_SUPPORTED_D, _packed_row_stride, _guarded, _f32c, _rows, _raw_ptr, _grad_or_zeros, _side_stream, colour_stream,
                      join_pending_updates, _await_updates, _take_event, set_deterministic)
from .projection import (_ProjectFn, fully_fused_projection, project_gaussians, _SHFn, spherical_harmonics, spherical_harmonics_decomposed,
                         sh_view_colors, _SHBatchedFn, sh_view_colors_batched)
from .binning import (isect_tiles, isect_offset_encode, _PendingBins, MAX_ISECTS, bin_gaussians_begin, bin_gaussians_end, bin_gaussians, LazyLists,
                      _bin_count_arrived, _bin_finish, _emit)
from .compositing import (_CompositeFn, _composite, rasterize_to_pixels, composite_scores, hit_pixel_count, rasterize_to_weights, rasterize_gaussians)
from .sharded import (unbind_cameras, pack_visible_records, pack_all_records, unpack_visible_records, _PackRecordsFn, _PackAllRecordsFn,
                      _UnpackRecordsFn, _StageCtx, _ShardFrontFn, _ShardExchangeFn, _ShardBackFn, sharded_front, sharded_exchange, sharded_back)
from .inria import GaussianRasterizationSettings, GaussianRasterizer, _InriaRasterizeFn, _InriaFusedFn, _split_sh
from .side import radix_sort_pairs, radix_sort_keys64, distCUDA2, l1_ssim, fused_ssim, photometric_loss


def _forward(name):
    return property(lambda self: getattr(STATE, name), lambda self, value: setattr(STATE, name, value))


class _OpsPackage(_types.ModuleType):
    """The module-level spellings of the run-time state (read AND write: tests and bench assign to them)."""
    FUSED_INRIA = _forward("fused_inria")
    DEVICE_SIDE_LIST_LENGTH = _forward("device_side_list_length")
    SPECULATIVE_EMIT = _forward("speculative_emit")
    TRACK_HIT_PIXELS = _forward("track_hit_pixels")
    KEEP_LAST_RASTER = _forward("keep_last_raster")
    SIDE_LOW_PRIORITY = _forward("side_low_priority")
    SEGMENTED_BACKWARD = _forward("segmented_backward")
    LAST_RASTER = _forward("last_raster")
    SPECULATION = _forward("speculation")
    PENDING_UPDATES = _forward("pending_updates")
    _LAST_ISECTS = _forward("last_isects")
    _EVENTS = _forward("events")
    _PINNED_WORDS = _forward("pinned_words")
    _PINNED_ENDS = _forward("pinned_ends")


_sys.modules[__name__].__class__ = _OpsPackage
