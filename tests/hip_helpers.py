"""Direct C-ABI call helpers for the GPU parity tests (tests only)."""
import numpy as np
import torch

from gspl_amd import _lib as L


def dev():
    return torch.device("cuda:0")


def t32(a, device=None):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).contiguous().to(device or dev())


def hip_composite_fwd(mode, means2d, conics, colors, opac, bg, W, H, offsets, flat, layout=L.GSPL_LAYOUT_HWC, hits=None):
    lib = L.lib()
    N, D = colors.shape
    tw, th = (W + 15) // 16, (H + 15) // 16
    shape = (H, W, D) if layout == L.GSPL_LAYOUT_HWC else (D, H, W)
    out = torch.empty(shape, dtype=torch.float32, device=means2d.device)
    alphas = torch.empty((H, W), dtype=torch.float32, device=means2d.device)
    final_T = torch.empty((H, W), dtype=torch.float32, device=means2d.device)
    last = torch.empty((H, W), dtype=torch.int32, device=means2d.device)
    n_isects = flat.shape[0]
    L.check(lib.gspl_composite_fwd(N, n_isects, D, mode, layout, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opac),
                                   L.ptr(bg), W, H, 16, tw, th, L.ptr(offsets), L.ptr(flat) if n_isects else None,
                                   L.ptr(out), L.ptr(alphas), L.ptr(final_T), L.ptr(last), L.ptr(hits), L.stream()), "composite_fwd")
    return out, alphas, final_T, last


def hip_composite_bwd(mode, means2d, conics, colors, opac, bg, W, H, offsets, flat, final_T, last, v_out, v_alpha=None,
                      absgrad=False, layout=L.GSPL_LAYOUT_HWC):
    lib = L.lib()
    N, D = colors.shape
    tw, th = (W + 15) // 16, (H + 15) // 16
    d = means2d.device
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=d)
    v_xy, v_con, v_col, v_op = z(N, 2), z(N, 3), z(N, D), z(N)
    v_abs = z(N, 2) if absgrad else None
    hit = torch.zeros(N, dtype=torch.uint8, device=d)
    L.check(lib.gspl_composite_bwd(N, flat.shape[0], D, mode, layout, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opac),
                                   L.ptr(bg), W, H, 16, tw, th, L.ptr(offsets), L.ptr(flat), L.ptr(final_T), L.ptr(last),
                                   L.ptr(v_out), L.ptr(v_alpha), L.ptr(v_xy), L.ptr(v_abs), L.ptr(v_con), L.ptr(v_col),
                                   L.ptr(v_op), L.ptr(hit), L.stream()), "composite_bwd")
    return dict(v_means2d=v_xy, v_means2d_abs=v_abs, v_conics=v_con, v_colors=v_col, v_opacities=v_op, hit=hit)


def assert_close_scaled(got, ref, rel, name="", frac_ok=1.0):
    """|got - ref| <= rel * (|ref| + rms(ref)) elementwise (for at least `frac_ok` of the elements)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if ref.size == 0:
        return
    rms = float(np.sqrt(np.mean(ref * ref))) + 1e-30
    err = np.abs(got - ref)
    bound = rel * (np.abs(ref) + rms)
    bad = err > bound
    frac = 1.0 - bad.mean()
    assert frac >= frac_ok, f"{name}: {bad.sum()} / {bad.size} outside rel={rel} (worst {np.max(err / (np.abs(ref) + rms)):.3e})"
