"""
_lib.py — ctypes binding of libgspl_hip.so (the C-ABI declared in include/gspl_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every compute call goes
through the C-ABI with raw device pointers.  There is NO fallback: if the library is missing, or
a tensor is not on the GPU, the call raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSPL_HIP_LIB", os.path.join(_PKG_DIR, "libgspl_hip.so"))   # override: A/B builds of the same ABI
ABI_VERSION = 35

GSPL_RECORD_FLOATS = 12
GSPL_CAMERA_PINHOLE, GSPL_CAMERA_ORTHO, GSPL_CAMERA_FISHEYE = 0, 1, 2
CAMERA_MODELS = {"pinhole": 0, "ortho": 1, "fisheye": 2}
GSPL_MODE_GSPLAT = 0
GSPL_MODE_INRIA = 1
GSPL_LAYOUT_HWC = 0
GSPL_LAYOUT_CHW = 1
GSPL_SH_ADD_HALF_CLAMP = 1
GSPL_INRIA_GEOMETRY, GSPL_INRIA_COLOURS, GSPL_INRIA_ALL = 1, 2, 3
GSPL_INRIA_RAW_PARAMS = 1      # gspl_inria_state.flags: scales / rotations / opacities are the model's raw parameters
GSPL_INRIA_NO_SEGMENTS = 2     # ... never segment the backward (the plain one-workgroup-per-tile walk)
GSPL_INRIA_FORCE_SEGMENTS = 4  # ... always (default: adaptively, while walks longer than a segment are being met)
GSPL_INRIA_WILL_BACKWARD = 8   # ... IN: a backward follows: the forward clears the backward's packed rows (GSPL_BUF_PACKED)
GSPL_INRIA_PACKED_READY = 16   # ... OUT: it did
GSPL_BIN_SPAN_BYTES = 64
GSPL_ADAM_MAX_TENSORS = 16


class AdamTensor(ctypes.Structure):
    """`gspl_adam_tensor` of include/gspl_hip.h."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("lr", ctypes.c_float), ("row_elems", ctypes.c_int32)]


class BwdAdamTensor(ctypes.Structure):
    """`gspl_bwd_adam_tensor` of include/gspl_hip.h (moments + hyper-parameters of one parameter updated inside the backward)."""
    _fields_ = [("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p), ("lr", ctypes.c_float), ("beta1", ctypes.c_float),
                ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("bias_correction1", ctypes.c_float), ("bias_correction2_sqrt", ctypes.c_float)]


class BwdAdamPlan(ctypes.Structure):
    """`gspl_bwd_adam_plan`."""
    _fields_ = [(n, BwdAdamTensor) for n in ("means", "scales", "rotations", "opacities", "shs", "shs_rest")]


class HipLibraryError(RuntimeError):
    pass


# `gspl_alloc_fn` / `gspl_inria_state` of include/gspl_hip.h (the fused Inria entry points)
GSPL_BUF_GEOMETRY, GSPL_BUF_BINNING, GSPL_BUF_IMAGE, GSPL_BUF_LISTS_WORK, GSPL_BUF_LISTS, GSPL_BUF_CHECKPOINTS, GSPL_BUF_PACKED = 1, 2, 3, 4, 5, 6, 7
ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t)


class InriaState(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int), ("n_isects", ctypes.c_int64),
                ("means2d", ctypes.c_void_p), ("depths", ctypes.c_void_p), ("conics", ctypes.c_void_p), ("colors", ctypes.c_void_p),
                ("clamped", ctypes.c_void_p), ("cov3d", ctypes.c_void_p), ("sh_jac", ctypes.c_void_p),
                ("alphas", ctypes.c_void_p), ("final_Ts", ctypes.c_void_p), ("last_ids", ctypes.c_void_p), ("offsets", ctypes.c_void_p),
                ("flatten_ids", ctypes.c_void_p), ("opacities", ctypes.c_void_p), ("flags", ctypes.c_int),
                ("seg_ckpt", ctypes.c_void_p), ("seg_words", ctypes.c_void_p), ("seg_slots", ctypes.c_uint32), ("seg_reserved", ctypes.c_uint32),
                ("stats_accum", ctypes.c_void_p), ("stats_denom", ctypes.c_void_p), ("stats_max_radii", ctypes.c_void_p)]


_P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "gspl_abi_version": (c_int, []),
    "gspl_last_error": (ctypes.c_char_p, []),
    "gspl_composite_bwd_kernel_name": (ctypes.c_char_p, []),
    "gspl_set_deterministic": (c_int, [c_int]),
    "gspl_get_deterministic": (c_int, []),
    "gspl_project_fwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int,
                                 c_float, c_float, c_float, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_project_bwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_int,
                                 _P, _P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "gspl_low_priority_stream": (c_void_p, []),
    "gspl_records_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gspl_records_pack_fwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gspl_records_count_fwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gspl_records_scatter_fwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_records_pad_fwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_records_pack_bwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_records_unpack_fwd": (c_int, [ctypes.c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_records_unpack_bwd": (c_int, [ctypes.c_int64, c_int, _P, _P, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P]),
    "gspl_peer_alloc": (c_int, [c_size_t, _P, _P]),
    "gspl_peer_open": (c_int, [_P, _P]),
    "gspl_peer_close": (c_int, [_P]),
    "gspl_peer_free": (c_int, [_P]),
    "gspl_peer_put_rows": (c_int, [c_int, _P, _P, _P, c_int, _P]),
    "gspl_peer_signal": (c_int, [c_int, _P, ctypes.c_uint64, _P]),
    "gspl_peer_wait": (c_int, [_P, c_int, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    "gspl_sh_fwd": (c_int, [c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P]),
    "gspl_sh_bwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P, _P, _P, _P]),
    "gspl_sh_fwd_batched": (c_int, [c_int, c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P]),
    "gspl_sh_bwd_batched": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P]),
    "gspl_isect_workspace_bytes": (c_size_t, [c_int, c_int64]),
    "gspl_isect_count": (c_int, [c_int, c_int, _P, _P, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "gspl_isect_emit_sort": (c_int, [c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "gspl_isect_offsets": (c_int, [c_int64, _P, c_int, c_int, _P, _P]),
    "gspl_bin_workspace_bytes": (c_size_t, [c_int, c_int64]),
    "gspl_loss_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gspl_loss_l1_ssim_fwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gspl_loss_photometric_fwd": (c_int, [c_int, c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gspl_loss_l1_ssim_bwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P]),
    "gspl_selective_adam": (c_int, [c_int, _P, c_int, _P, c_float, c_float, c_float, c_float, c_float, _P]),
    "gspl_selective_adam_limited": (c_int, [c_int, _P, c_int, _P, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "gspl_radix_sort_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "gspl_radix_sort_pairs_u32": (c_int, [c_int64, _P, _P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    "gspl_radix_sort_keys_u64": (c_int, [c_int64, _P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    "gspl_composite_scores": (c_int, [c_int, c_int64, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P,
                                      _P, _P, _P, _P, _P, _P, _P]),
    "gspl_densify_stats": (c_int, [c_int, _P, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_densify_stats_views": (c_int, [c_int, c_int, _P, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_knn_workspace_bytes": (c_size_t, [c_int]),
    "gspl_knn3_mean_dist2": (c_int, [c_int, _P, _P, _P, c_size_t, _P]),
    "gspl_bin_count": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gspl_bin_emit": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, _P, c_size_t, _P]),
    "gspl_bin_sort": (c_int, [c_int, c_int, c_int, c_int64, c_int64, _P, _P, _P, c_size_t, _P]),
    "gspl_bin_sort_device_count": (c_int, [c_int, c_int, c_int, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "gspl_bin_emit_sort": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "gspl_composite_fwd": (c_int, [c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, _P,
                                   c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_composite_bwd": (c_int, [c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, _P,
                                   c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P,
                                   _P, _P, _P, _P, _P, _P, _P]),
    "gspl_composite_bwd_packed": (c_int, [c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, _P,
                                          c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "gspl_inria_preprocess_fwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                          c_int, c_int, c_int, c_float, c_float, c_float,
                                          _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "gspl_rasterize_inria_fwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_float,
                                         ALLOC_FN, _P, c_int64, _P, _P, ctypes.POINTER(InriaState), _P, _P]),
    "gspl_rasterize_inria_bwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float,
                                         _P, ctypes.POINTER(InriaState), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gspl_rasterize_inria_bwd_adam": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float,
                                              _P, ctypes.POINTER(InriaState), _P, _P, _P, _P, _P, ctypes.POINTER(BwdAdamPlan), _P]),
    "gspl_profile_enable": (c_int, [c_int]),
    "gspl_profile_enable2": (c_int, [c_int, c_int]),
    "gspl_profile_read": (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_float)]),
    "gspl_rasterize_inria_geometry_bytes": (c_size_t, [c_int]),
    "gspl_rasterize_inria_image_bytes": (c_size_t, [c_int, c_int]),
    "gspl_inria_state_bytes": (c_size_t, []),
    "gspl_inria_preprocess_bwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                          c_int, c_int, c_float, c_float, c_float,
                                          _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}

_LIB = None


def exported_symbols():
    """Names include/gspl_hip.h declares (used by the no-GPU ABI test)."""
    return sorted(_SIGNATURES)


def build(verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_PKG_DIR, "csrc"), "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise HipLibraryError("building libgspl_hip.so failed (see output above)")
    return LIB_PATH


def lib():
    """Load libgspl_hip.so (once).  Raises HipLibraryError when it is absent — there is no CPU path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the HIP extension is the only compute path of this package. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C "
            f"{os.path.join(_PKG_DIR, 'csrc')}`.")
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"could not load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in _SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    got = handle.gspl_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library {got}, python binding {ABI_VERSION}; rebuild the extension")
    handle.gspl_inria_state_bytes.restype = c_size_t
    if handle.gspl_inria_state_bytes() != ctypes.sizeof(InriaState):
        raise HipLibraryError(f"gspl_inria_state: the library's struct has {handle.gspl_inria_state_bytes()} bytes, the binding's "
                              f"{ctypes.sizeof(InriaState)}; rebuild the extension")
    _LIB = handle
    return _LIB


# Optional per-call device timing (bench.py): a list that receives (name, start_event, end_event).
# Events are recorded on torch's current stream, which is the stream every kernel is launched on.
_PROFILE = None
_PROFILE_PERIOD = 1
_PROFILE_SEEN: dict = {}
_PROFILE_NAMES = None


def profile_start(names=None, period: int = 1):
    """Time C-ABI calls with events on the current stream; `names`: only these entry points (None: all); `period`: every
    period-th call of a name is timed (an event pair costs the stream ~6 us of idle time on either side of the call)."""
    global _PROFILE, _PROFILE_NAMES, _PROFILE_PERIOD, _PROFILE_SEEN
    _PROFILE = []
    _PROFILE_NAMES = None if names is None else frozenset(names)
    _PROFILE_PERIOD, _PROFILE_SEEN = max(int(period), 1), {}
    on = lambda name: _PROFILE_PERIOD if (_PROFILE_NAMES is None or name in _PROFILE_NAMES) else 0
    lib().gspl_profile_enable2(on("gspl_composite_fwd"), on("gspl_composite_bwd_packed"))


def profile_stop():
    """Returns {name: [ms, ...]} (synchronises).  The compositing launches made inside the fused Inria calls are reported
    under the names of their stage entry points (as one entry holding the mean, repeated per launch)."""
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in rec or []:
        out.setdefault(name, []).append(e0.elapsed_time(e1))
    for which, name in ((0, "gspl_composite_fwd"), (1, "gspl_composite_bwd_packed")):
        n, ms = c_int(0), c_float(0.0)
        check(lib().gspl_profile_read(which, ctypes.byref(n), ctypes.byref(ms)), "gspl_profile_read")
        if n.value > 0 and (_PROFILE_NAMES is None or name in _PROFILE_NAMES):
            out.setdefault(name, []).extend([ms.value / n.value] * n.value)
    lib().gspl_profile_enable(0)
    return out


def call(name: str, *args):
    """Invoke one C-ABI entry point and raise on a non-zero status."""
    fn = getattr(lib(), name)
    timed = _PROFILE is not None and (_PROFILE_NAMES is None or name in _PROFILE_NAMES)
    if timed and _PROFILE_PERIOD > 1:
        seen = _PROFILE_SEEN.get(name, 0)
        _PROFILE_SEEN[name] = seen + 1
        timed = seen % _PROFILE_PERIOD == 0
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        _PROFILE.append((name, e0, e1))
    else:
        rc = fn(*args)
    check(rc, name)


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().gspl_last_error()
        raise RuntimeError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")


def ptr(t, dtype=None, offset_bytes: int = 0):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("gspl ops run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("gspl ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    return c_void_p(t.data_ptr() + offset_bytes)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


class device_guard:
    """`with device_guard(tensor):` makes the tensor's device current for the enclosed launches (so that `stream()` hands
    out THAT device's current stream) and restores the previous one; free when it already is the current device."""
    __slots__ = ("idx", "prev")

    def __init__(self, t):
        dev = t.device if hasattr(t, "device") else t
        self.idx = dev.index if dev.index is not None else (_raw_device() if _raw_device is not None else torch.cuda.current_device())
        self.prev = -1

    def __enter__(self):
        cur = _raw_device() if _raw_device is not None else torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def stream():
    """Raw handle of torch's current stream on the current device.  `torch.cuda.current_stream().cuda_stream` costs ~10 us
    of Python per call (device-index plumbing, a Stream object) and every op wrapper needs it: the C accessors do the
    same in well under a microsecond."""
    if _raw_stream is not None and _raw_device is not None:
        return c_void_p(_raw_stream(_raw_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)
