"""HIP against the fp64 oracle AT the sizes BASELINE.json quotes (not only at sizes the oracle finishes instantly):

  * S-1080p-1M (the metric point: 1 M Gaussians, 1920x1080), both APIs: every pixel and all five parameter gradients;
  * S-1080p-6M (configs[2] proxy, ~6 M Gaussians): projection, tile lists (bit-exact against the oracle's stable sort on the
    same projected inputs — the >1 M sort path) and the composited image;
  * a configs[4] proxy (MatrixCity: SH degree 0, absgrad densification, millions of Gaussians): 5 M splats through the gsplat
    API with `.absgrad`.

Tolerances: forward >= 99.9 % of the pixels within 1e-5 and ALL within 4e-3 (= one flipped 1/255 decision); gradients >= 99.5 % of the elements within
1e-4 * (|ref| + rms), all but 5e-5 of them within 1e-2, all but a counted handful (<= 8) within 0.05 and ALL within 0.5 (hip_helpers.assert_close_scaled:
the tail is counted and printed).  These tiers belong to the FREE-RUNNING oracle; tests/test_locked_parity.py repeats the metric point on the GPU's own
discrete decisions with no tail at all.
The fp64 oracle pass takes 10-60 s of host time per case."""
import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as O
from hip_helpers import assert_close_scaled, assert_pixels_close, assert_pipeline_attributed, cov2d_condition, cov_chain_slack, footprint_slack, means2d_slack

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# FREE-RUNNING comparison (the oracle re-takes every discrete decision in fp64): cap on the gradient elements relative to
# |ref| + rms(ref), with at most OUTLIERS elements (of 3-59 million; measured 1-4 of them, worst 1.5e-1) beyond it, none beyond 0.5; in
# addition at most 5e-5 of the elements may exceed 1e-2.  The comparison that excuses NOTHING — the oracle on the GPU's own discrete
# decisions, every pixel within 1e-5 and every gradient element within 1e-4 — is tests/test_locked_parity.py.
TAIL = 0.05
OUTLIERS = 8


def _hip_gsplat(params, cam, deg, bg, absgrad=False):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    W, H = cam["width"], cam["height"]
    leaves = [t.to(DEV).requires_grad_(True) for t in params]
    m, s, q, o, c = leaves
    vm = cam["world_to_camera"].T.contiguous().to(DEV)
    xys, depths, radii, conics, comp, tiles, _ = ops.project_gaussians(m, s, 1.0, q, vm[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, 16)
    if xys.requires_grad:
        xys.retain_grad()
    rgbs = ops.sh_view_colors(deg, m, cam["camera_center"].to(DEV), c, None, radii > 0)
    op = o * comp[:, None]
    img = ops.rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, op, H, W, 16, bg.to(DEV), absgrad=absgrad, channels_first=True)
    return img, leaves, dict(xys=xys, depths=depths, radii=radii, conics=conics, comp=comp, tiles=tiles, rgbs=rgbs, op=op)


def _oracle_gsplat(params, cam, deg, bg):
    W, H = cam["width"], cam["height"]
    dl = [t.double().requires_grad_(True) for t in params]
    r = O.render_gsplat(*dl, deg, cam["world_to_camera"].double(), cam["fx"], cam["fy"], cam["cx"], cam["cy"], W, H, bg.double(),
                        cam["camera_center"].double())
    return r, dl


# Free-running comparisons at 1080p: the per-splat inputs the two sides composite differ by a few fp32 ulps of their own magnitude
# (1e-4 pixels at x ~ 1900), which moves a pixel under a bright splat's edge by up to ~1.5e-5 with every decision identical
# (measured: 30 unflagged pixels of 2 M between 1e-5 and 1.49e-5 through the Inria API; the same pixels are within 4e-7 of the oracle
# composited AT the GPU's values — tests/test_locked_parity.py holds every unflagged pixel to 1e-5 there).
PIXEL_TOL_FREE_1080P = 2e-5


def _compare_grads(leaves, dl, mode, r, W, H, bg, render, gpu_radii, extra=(), slack=None, pixel_tol=PIXEL_TOL_FREE_1080P):
    """Attributed: every pixel the oracle does not flag within the pixel tolerance; every gradient element beyond
    1e-4 (|ref| + rms) belongs to a splat a flagged decision reaches (hip_helpers.assert_pipeline_attributed) — the fragile rows keep
    the old caps (every element within 0.5, at least 99.5 % of all elements within 1e-4)."""
    pairs = []
    for got, ref, name in zip(leaves, dl, ("means", "scales", "quats", "opacities", "shs")):
        g = got.grad if hasattr(got, "grad") and not isinstance(got, np.ndarray) else got
        assert g is not None, name
        pairs.append((name, g.cpu().numpy() if hasattr(g, "cpu") else g, ref.grad.numpy()))
    return assert_pipeline_attributed(mode, r, W, H, bg.double(), render, pairs + list(extra), opacities=dl[3], gpu_radii=gpu_radii,
                                      pixel_tol=pixel_tol, slack=slack)


def _vanilla_against_the_oracle(params, cam, W, H, wimg, bg, radii_frac=0.9995, conditioned=False):
    """Inria API (`ops.GaussianRasterizer`, as vanilla_renderer.py:25-129 calls it), SH degree 3: image, the five parameter
    gradients and `viewspace_points.grad` (NDC units) against `O.render_inria`.
    conditioned: the per-row allowances of tests/test_locked_parity.py (hip_helpers.cov_chain_slack / footprint_slack: the fp32
    conditioning of conic -> cov2D -> cov3D and of a sum over a large footprint) under the tiers — for scenes with long anisotropic
    splats, where the tiers alone would be measuring that conditioning instead of flipped decisions."""
    from gspl_amd import ops
    leaves = [t.to(DEV).requires_grad_(True) for t in params]
    m, s, q, o, c = leaves
    settings = ops.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(DEV), scale_modifier=1.0,
        viewmatrix=cam["world_to_camera"].to(DEV), projmatrix=cam["full_projection"].to(DEV), sh_degree=3, campos=cam["camera_center"].to(DEV))
    screen = torch.zeros_like(m, requires_grad=True)
    render, radii = ops.GaussianRasterizer(settings)(means3D=m, means2D=screen, opacities=o, shs=c, scales=s, rotations=q)
    (render * wimg.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    got = dict(render=render.detach().cpu().numpy(), radii=radii.cpu().numpy(), screen=screen.grad[:, :2].cpu().numpy(),
               grads=[t.grad.cpu().numpy() for t in leaves])
    del leaves, m, s, q, o, c, screen, render, radii
    torch.cuda.empty_cache()
    dl = [t.double().requires_grad_(True) for t in params]
    r = O.render_inria(*dl, 3, cam["world_to_camera"].double(), cam["full_projection"].double(), cam["camera_center"].double(),
                       cam["tanfovx"], cam["tanfovy"], W, H, bg.double())
    (r["render"] * wimg.double()).sum().backward()
    assert np.mean(got["radii"] == r["radii"].numpy()) > radii_frac
    ref_ndc = r["xy"].grad.numpy() * np.array([0.5 * W, 0.5 * H])
    slack = None
    if conditioned:
        kappa, extent = cov2d_condition(r["conics"].detach().numpy()), r["radii"].numpy().astype(np.float64)
        sl = lambda ref, cov: footprint_slack(ref, extent) + (cov_chain_slack(ref, kappa) if cov else 0.0)
        slack = {name: sl(ref.grad.numpy(), name in ("means", "scales", "quats")) for name, ref in zip(("means", "scales", "quats", "opacities", "shs"), dl)}
        # (the screen-space gradient of a needle is A Sx + B Sy with the two products far larger than their sum: hip_helpers.means2d_slack,
        # as the locked test allows it)
        slack["viewspace_points.grad"] = sl(ref_ndc, False) + means2d_slack(r["xy"].grad.numpy(), r["conics"].detach().numpy(), r["radii"].numpy()) * np.array([0.5 * W, 0.5 * H])
    _compare_grads(got["grads"], dl, O.MODE_INRIA, r, W, H, bg, got["render"], got["radii"],
                   extra=[("viewspace_points.grad", got["screen"], ref_ndc)], slack=slack, pixel_tol=None if conditioned else PIXEL_TOL_FREE_1080P)


def test_config2_proxy_S_1080p_6M_inria_api_gradients():
    """BASELINE.json configs[2] (garden-sized model, ~6 M Gaussians, the VANILLA renderer, SH degree 3: vanilla_renderer.py:25-129)
    at the metric resolution: forward, all five parameter gradients and `viewspace_points.grad` through the Inria API — the stages
    that dominate a 6 M step (masked SH / preprocess backward over 6 M rows, the > 1 M depth sort) under a gradient check
    (missing #2)."""
    import gspl_amd  # noqa: F401
    from gspl_amd import synthetic
    wl = synthetic.WORKLOADS["S-1080p-6M"]
    W, H = wl["width"], wl["height"]
    params = O.synthetic_scene(wl["n"], seed=42)
    cam = O.synthetic_camera(W, H, wl["fx"])
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(4))
    _vanilla_against_the_oracle(params, cam, W, H, wimg, torch.tensor([0.1, 0.2, 0.3]), radii_frac=0.99999)


def test_trained_scene_shaped_workload_S_1080p_1M_surfaces_against_the_oracle():
    """`synthetic.scene_surfaces` (opaque surfaces, saturating pixels, heavy-tailed lists) free-running: the oracle re-takes every
    decision in fp64.  The locked form is tests/test_locked_parity.py::test_trained_scene_shaped_workload_locked."""
    import gspl_amd  # noqa: F401
    from gspl_amd import synthetic
    wl = synthetic.WORKLOADS["S-1080p-1M-surfaces"]
    W, H = wl["width"], wl["height"]
    params = synthetic.workload_scene(wl, seed=42)
    cam = O.synthetic_camera(W, H, wl["fx"])
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(6))
    _vanilla_against_the_oracle(params, cam, W, H, wimg, torch.tensor([0.1, 0.2, 0.3]), conditioned=True)


@pytest.mark.parametrize("api", ["vanilla", "gsplat"])
def test_metric_point_S_1080p_1M_against_the_oracle(api):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops, synthetic
    wl = synthetic.WORKLOADS["S-1080p-1M"]
    W, H = wl["width"], wl["height"]
    params = O.synthetic_scene(wl["n"], seed=42)
    cam = O.synthetic_camera(W, H, wl["fx"])
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    bg = torch.tensor([0.1, 0.2, 0.3])
    if api == "vanilla":
        leaves = [t.to(DEV).requires_grad_(True) for t in params]
        m, s, q, o, c = leaves
        settings = ops.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(DEV), scale_modifier=1.0,
            viewmatrix=cam["world_to_camera"].to(DEV), projmatrix=cam["full_projection"].to(DEV), sh_degree=3, campos=cam["camera_center"].to(DEV))
        screen = torch.zeros_like(m, requires_grad=True)
        render, radii = ops.GaussianRasterizer(settings)(means3D=m, means2D=screen, opacities=o, shs=c, scales=s, rotations=q)
        (render * wimg.to(DEV)).sum().backward()
        dl = [t.double().requires_grad_(True) for t in params]
        r = O.render_inria(*dl, 3, cam["world_to_camera"].double(), cam["full_projection"].double(), cam["camera_center"].double(),
                           cam["tanfovx"], cam["tanfovy"], W, H, bg.double())
        (r["render"] * wimg.double()).sum().backward()
        assert np.mean(radii.cpu().numpy() == r["radii"].numpy()) > 0.9995
        ref_ndc = r["xy"].grad.numpy() * np.array([0.5 * W, 0.5 * H])
        mode, extra = O.MODE_INRIA, [("viewspace_points.grad", screen.grad[:, :2].cpu().numpy(), ref_ndc)]
    else:
        render, leaves, mid = _hip_gsplat(params, cam, 3, bg)
        (render * wimg.to(DEV)).sum().backward()
        r, dl = _oracle_gsplat(params, cam, 3, bg)
        (r["render"] * wimg.double()).sum().backward()
        mode, extra, radii = O.MODE_GSPLAT, [("xys.grad", mid["xys"].grad.cpu().numpy(), r["xys"].grad.numpy())], mid["radii"]
    _compare_grads(leaves, dl, mode, r, W, H, bg, render.detach().cpu().numpy(), radii, extra=extra)


def test_config2_proxy_S_1080p_6M_projection_lists_and_image():
    """~6 M Gaussians at 1080p (the garden-sized model of configs[2]): the depth sort and scan run on the >1 M code path."""
    import gspl_amd  # noqa: F401
    from gspl_amd import ops, synthetic
    wl = synthetic.WORKLOADS["S-1080p-6M"]
    W, H = wl["width"], wl["height"]
    params = O.synthetic_scene(wl["n"], seed=42)
    cam = O.synthetic_camera(W, H, wl["fx"])
    bg = torch.zeros(3)
    with torch.no_grad():
        render, _, mid = _hip_gsplat(params, cam, 3, bg)
        # projection against the fp64 oracle
        m, s, q, o, c = [t.double() for t in params]
        xys, depths, radii, conics, comp, n_tiles, _, mask, _, _ = O.project_gaussians(
            m, s, 1.0, q, cam["world_to_camera"].double(), cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
        assert np.mean(mid["radii"].cpu().numpy() == radii.numpy()) > 0.99999      # (measured: 3 in a million on a rounding boundary, profiles/r05b_extent_probe.txt)
        assert_close_scaled(mid["xys"].cpu().numpy(), xys.numpy(), 1e-5, "xys", frac_ok=0.9999, rel_all=1e-3)
        assert_close_scaled(mid["conics"].cpu().numpy(), conics.numpy(), 1e-4, "conics", frac_ok=0.9995, rel_all=TAIL, outliers=OUTLIERS)
        # tile lists of the HIP projection's own outputs: bit-exact against the oracle's stable (tile | depth) sort
        flat, offs = ops.bin_gaussians(mid["xys"], mid["depths"], mid["radii"], H, W, 16)
        _, _, flat_ref, offs_ref = O.isect_tiles(O.MODE_GSPLAT, mid["xys"].cpu(), mid["radii"].cpu(), mid["depths"].cpu(), W, H)
        assert flat.numel() == int(mid["tiles"].sum()) and flat.numel() > 60_000_000
        assert np.array_equal(offs.cpu().numpy(), offs_ref) and np.array_equal(flat.cpu().numpy(), flat_ref)
        # image against the C oracle on the same per-splat inputs and lists
        ref, _, _, frag = O.composite_fwd(O.MODE_GSPLAT, mid["xys"].cpu(), mid["conics"].cpu(), mid["rgbs"].cpu(), mid["op"].reshape(-1).cpu(),
                                          bg, W, H, offs_ref, flat_ref)
        assert_pixels_close(render.permute(1, 2, 0).cpu().numpy(), ref)


def test_config4_proxy_sh0_absgrad_5M():
    """MatrixCity-like settings (configs/matrixcity/gsplat-aerial.yaml: SH degree 0; configs/gsplat-absgrad.yaml: densification on
    the absolute screen-space gradient) at 5 M Gaussians: image, parameter gradients and `.absgrad` against the oracle."""
    N, W, H = 5_000_000, 1920, 1080
    means, scales, quats, opac, shs = O.synthetic_scene(N, seed=7, sh_degree=0)
    scales = scales * 0.6
    params = (means, scales, quats, opac, shs)
    cam = O.synthetic_camera(W, H, 1600.0)
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2))
    bg = torch.zeros(3)
    render, leaves, mid = _hip_gsplat(params, cam, 0, bg, absgrad=True)
    (render * wimg.to(DEV)).sum().backward()
    r, dl = _oracle_gsplat(params, cam, 0, bg)
    (r["render"] * wimg.double()).sum().backward()
    rows, _ = _compare_grads(leaves, dl, O.MODE_GSPLAT, r, W, H, bg, render.detach().cpu().numpy(), mid["radii"],
                             extra=[("xys.grad", mid["xys"].grad.cpu().numpy(), r["xys"].grad.numpy())])
    # absgrad: sum over pixels of |per-pixel gradient|; the oracle's analytic backward on the same lists
    ab = mid["xys"].absgrad
    assert ab.shape == (N, 2) and bool((ab >= mid["xys"].grad.abs() - 1e-6).all())
    d = lambda t: t.detach()
    _, alpha_ref, last_ref, _ = O.composite_fwd(O.MODE_GSPLAT, d(r["xys"]), d(r["conics"]), d(r["rgbs"]), d(r["opacities"]), bg.double(), W, H,
                                                r["offsets"], r["flatten_ids"])
    g = O.composite_bwd(O.MODE_GSPLAT, d(r["xys"]), d(r["conics"]), d(r["rgbs"]), d(r["opacities"]), bg.double(), W, H, r["offsets"],
                        r["flatten_ids"], alpha_ref, last_ref, wimg.permute(1, 2, 0).double().numpy(), None, absgrad=True)
    from hip_helpers import assert_close_attributed
    assert_close_attributed(ab.cpu().numpy(), g["v_means2d_abs"], 1e-4, "xys.absgrad", rows)


def test_config4_scale_20M_sh0_lists_and_image():
    """BASELINE.json configs[4] at its SCALE (configs/matrixcity/gsplat-aerial.yaml:25: ~20 M Gaussians, SH degree 0): the bench
    workload S-1080p-20M-sh0-absgrad through the gsplat API.  Projection against the fp64 oracle; the tile lists (> 50 M entries)
    against the independent 64-bit (tile | depth) device-wide sort behind `isect_tiles` and by their defining property (every
    tile's list ascending in depth bits, ties in id order); the image against the C oracle compositing the same lists."""
    import gspl_amd  # noqa: F401
    from gspl_amd import ops, synthetic
    wl = synthetic.WORKLOADS["S-1080p-20M-sh0-absgrad"]
    W, H = wl["width"], wl["height"]
    params = synthetic.workload_scene(wl, seed=42)
    cam = O.synthetic_camera(W, H, wl["fx"])
    bg = torch.zeros(3)
    with torch.no_grad():
        render, _, mid = _hip_gsplat(params, cam, 0, bg)
        m, s, q, o, c = [t.double() for t in params]
        xys, depths, radii, conics, comp, n_tiles, _, mask, _, _ = O.project_gaussians(
            m, s, 1.0, q, cam["world_to_camera"].double(), cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
        assert np.mean(mid["radii"].cpu().numpy() == radii.numpy()) > 0.99999
        assert_close_scaled(mid["xys"].cpu().numpy(), xys.numpy(), 1e-5, "xys", frac_ok=0.9999, rel_all=1e-3)
        assert_close_scaled(mid["conics"].cpu().numpy(), conics.numpy(), 1e-4, "conics", frac_ok=0.9995, rel_all=TAIL, outliers=OUTLIERS)
        del xys, conics, comp, m, s, q, o, c
        tw, th = (W + 15) // 16, (H + 15) // 16
        flat, offs = ops.bin_gaussians(mid["xys"], mid["depths"], mid["radii"], H, W, 16)
        assert flat.numel() == int(mid["tiles"].sum()) and flat.numel() > 50_000_000
        _, ids, flat_ref = ops.isect_tiles(mid["xys"][None], mid["radii"][None], mid["depths"][None], 16, tw, th)
        assert torch.equal(flat, flat_ref) and torch.equal(offs, ops.isect_offset_encode(ids, 1, tw, th).reshape(-1))
        # the defining property, checked on the device: inside a tile (depth bits, id) ascends strictly
        tile_of = torch.repeat_interleave(torch.arange(tw * th, device=DEV), torch.diff(torch.cat([offs, offs.new_tensor([flat.numel()])])).long())
        key = (mid["depths"][flat.long()].view(torch.int32).long() << 32) | flat.long()
        same = tile_of[1:] == tile_of[:-1]
        assert bool((key[1:][same] > key[:-1][same]).all())
        del tile_of, key, same, ids, flat_ref
        ref, _, _, frag = O.composite_fwd(O.MODE_GSPLAT, mid["xys"].cpu(), mid["conics"].cpu(), mid["rgbs"].cpu(), mid["op"].reshape(-1).cpu(),
                                          bg, W, H, offs.cpu().numpy(), flat.cpu().numpy())
        assert_pixels_close(render.permute(1, 2, 0).cpu().numpy(), ref)
