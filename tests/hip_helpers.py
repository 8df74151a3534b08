"""Direct C-ABI call helpers for the GPU parity tests (tests only)."""
import numpy as np
import torch

from gspl_amd import _lib as L


def dev():
    return torch.device("cuda:0")


def t32(a, device=None):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).contiguous().to(device or dev())


def hip_composite_fwd(mode, means2d, conics, colors, opac, bg, W, H, offsets, flat, layout=L.GSPL_LAYOUT_HWC, hits=None, tile=16):
    lib = L.lib()
    N, D = colors.shape
    tw, th = (W + tile - 1) // tile, (H + tile - 1) // tile
    shape = (H, W, D) if layout == L.GSPL_LAYOUT_HWC else (D, H, W)
    out = torch.empty(shape, dtype=torch.float32, device=means2d.device)
    alphas = torch.empty((H, W), dtype=torch.float32, device=means2d.device)
    final_T = torch.empty((H, W), dtype=torch.float32, device=means2d.device)
    last = torch.empty((H, W), dtype=torch.int32, device=means2d.device)
    n_isects = flat.shape[0]
    L.check(lib.gspl_composite_fwd(N, n_isects, D, mode, layout, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opac),
                                   L.ptr(bg), W, H, tile, tw, th, L.ptr(offsets), L.ptr(flat) if n_isects else None,
                                   L.ptr(out), L.ptr(alphas), L.ptr(final_T), L.ptr(last), L.ptr(hits), L.stream()), "composite_fwd")
    return out, alphas, final_T, last


def hip_composite_bwd(mode, means2d, conics, colors, opac, bg, W, H, offsets, flat, final_T, last, v_out, v_alpha=None,
                      absgrad=False, layout=L.GSPL_LAYOUT_HWC, tile=16):
    lib = L.lib()
    N, D = colors.shape
    tw, th = (W + tile - 1) // tile, (H + tile - 1) // tile
    d = means2d.device
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=d)
    v_xy, v_con, v_col, v_op = z(N, 2), z(N, 3), z(N, D), z(N)
    v_abs = z(N, 2) if absgrad else None
    hit = torch.zeros(N, dtype=torch.uint8, device=d)
    L.check(lib.gspl_composite_bwd(N, flat.shape[0], D, mode, layout, L.ptr(means2d), L.ptr(conics), L.ptr(colors), L.ptr(opac),
                                   L.ptr(bg), W, H, tile, tw, th, L.ptr(offsets), L.ptr(flat), L.ptr(final_T), L.ptr(last),
                                   L.ptr(v_out), L.ptr(v_alpha), L.ptr(v_xy), L.ptr(v_abs), L.ptr(v_con), L.ptr(v_col),
                                   L.ptr(v_op), L.ptr(hit), L.stream()), "composite_bwd")
    return dict(v_means2d=v_xy, v_means2d_abs=v_abs, v_conics=v_con, v_colors=v_col, v_opacities=v_op, hit=hit)


def assert_close_scaled(got, ref, rel, name="", frac_ok=1.0, rel_all=None, frac_1e2=None, outliers=0, rel_outliers=0.5):
    """ratio = |got - ref| / (|ref| + rms(ref)) elementwise, with three tiers:
         ratio <= rel       for at least `frac_ok` of the elements (the claimed tolerance);
         ratio <= 1e-2      for at least `frac_1e2` of them (default: all when rel_all <= 1e-2, else 1 - 5e-5; measured up to 3e-5);
         ratio <= rel_all   for EVERY element (the hard cap on the excused tail; None: no cap unless frac_ok == 1).
    Why a tail exists at all: a splat whose alpha meets the 1/255 skip threshold (or a pixel its transmittance stop) within the
    ~1e-6 relative difference between the hardware exp/rcp and the fp64 oracle gains or loses ONE pixel's whole contribution; for
    a splat seen by few pixels that is a sizeable share of a small gradient.  The tiers keep such flips rare (counted, printed)
    and bounded, while a systematically wrong gradient (ratio ~ 1 or more on many elements) fails all of them."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if ref.size == 0:
        return
    assert np.isfinite(got).all(), f"{name}: non-finite values"
    rms = float(np.sqrt(np.mean(ref * ref))) + 1e-30
    ratio = np.abs(got - ref) / (np.abs(ref) + rms)
    bad = ratio > rel
    worst = float(ratio.max())
    n2 = int((ratio > 1e-2).sum())
    if bad.any():
        print(f"[tail] {name}: {int(bad.sum())} of {bad.size} elements outside rel={rel:g}, {n2} outside 1e-2 (worst {worst:.3e})")
    frac = 1.0 - bad.mean()
    assert frac >= frac_ok, f"{name}: {bad.sum()} / {bad.size} outside rel={rel} (worst {worst:.3e})"
    if rel_all is not None:
        n_over = int((ratio > rel_all).sum())
        if n_over:
            print(f"[tail] {name}: {n_over} element(s) beyond the cap {rel_all:g} (worst {worst:.3e}); allowed: {outliers} up to {rel_outliers:g}")
        # `outliers` elements (a counted handful among millions: single flipped decisions on splats that few pixels see) may pass the
        # cap, up to `rel_outliers`
        assert n_over <= outliers and worst <= (rel_outliers if outliers else rel_all), \
            f"{name}: worst element {worst:.3e}, {n_over} beyond the tail cap {rel_all:g} (allowed {outliers} up to {rel_outliers:g})"
        if rel_all > 1e-2 and rel < 1e-2:
            allowed = (5e-5 if frac_1e2 is None else 1.0 - frac_1e2) * ratio.size
            assert n2 <= max(allowed, 1.0), f"{name}: {n2} elements outside 1e-2 (allowed {allowed:.1f} of {ratio.size})"


def assert_pixels_close(got, ref, tol=1e-5, frac_ok=0.999, tol_all=4e-3, name="render"):
    """|got - ref| <= tol for at least frac_ok of the pixels and <= tol_all for EVERY pixel.  The tail bound is the size of ONE
    flipped discrete decision: a splat whose alpha meets the 1/255 skip threshold within the ~1e-6 relative difference between
    the hardware exp/rcp and fp64 is blended by one side and skipped by the other, which moves the pixel by
    alpha T |c - C_behind| <= 1/255 = 3.9e-3 for colours in [0, 1] (measured maxima at 1-6 M splats: 1.1e-3 ... 2.3e-3).
    Counts are reported."""
    d = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
    bad = d > tol
    if bad.any():
        print(f"[tail] {name}: {int(bad.sum())} of {bad.size} values outside {tol:g} (max {d.max():.3e})")
    assert 1.0 - bad.mean() >= frac_ok, f"{name}: {bad.sum()} / {bad.size} outside {tol} (max {d.max():.3e})"
    assert d.max() <= tol_all, f"{name}: max |diff| {d.max():.3e} exceeds the tail bound {tol_all:g}"
