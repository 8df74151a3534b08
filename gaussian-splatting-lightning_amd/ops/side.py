"""Side functions of the path: radix sort wrappers, `distCUDA2`, the fused photometric loss."""
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
