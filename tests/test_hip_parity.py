"""GPU parity tests: the HIP path (through the C-ABI / the autograd wrappers over it) against the
oracle on identical seeded inputs, against the committed golden fixtures, and — at BASELINE size —
through size-independent properties.

Tolerances (BASELINE.json north_star): forward <= 1e-5 abs per pixel, gradients <= 1e-4 relative.
Integer / index outputs (radii, tile counts, sort keys, offsets) are compared bit-exactly.
Pixels / splats whose discrete skip-stop-clamp decision sits within 2e-5 relative of its threshold
("fragile", flagged by the fp64 oracle) are excluded from the strict bound and must be rare.
"""
import os

import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gspl_amd  # noqa: F401
    from gspl_amd import _lib, ops
    _lib.lib()       # raises if the extension is missing: no silent fallback
    return ops


def _dev():
    return torch.device("cuda:0")


def _cuda(*ts):
    return [t.detach().float().contiguous().to(_dev()) for t in ts]


def _cam_tensors(cam):
    vm = cam["world_to_camera"].T.contiguous().float().to(_dev())
    K = torch.tensor([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1.0]], dtype=torch.float32, device=_dev())
    return vm, K


from hip_helpers import assert_pixels_close, assert_close_scaled, assert_pipeline_attributed, hip_composite_bwd, hip_composite_fwd, t32  # noqa: E402


# ---------------------------------------------------------------------------------------------
# projection
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cam", [0, 1])
def test_projection_vs_reference_golden(hip, golden_dir, cam):
    """HIP projection against the reference Python's own outputs and gradients (fixtures)."""
    z = np.load(os.path.join(golden_dir, "ref_projection.npz"))
    p = f"cam{cam}_"
    fx, fy, cx, cy, W, H = z[p + "intr"]
    means, scales, quats = [t32(z[k]).requires_grad_(True) for k in ("means", "scales", "quats")]
    viewmat = t32(z[p + "w2c"]).T.contiguous()
    xys, depths, radii, conics, comp, tiles, cov3d = hip.project_gaussians(
        means, scales, 1.0, quats, viewmat, float(fx), float(fy), float(cx), float(cy), int(H), int(W), 16)
    # cov3d [N,3,3] as the reference returns it (gaussian_projection.py:47,137): (R S)(R S)^T, zeros for culled Gaussians
    assert cov3d.shape == (means.shape[0], 3, 3) and not cov3d.requires_grad
    np.testing.assert_allclose(cov3d.cpu().numpy(), z[p + "cov3d"], rtol=2e-5, atol=1e-9)
    # integer outputs: bit-exact (the reference compares them with torch.equal, tests/gaussian_projection_test.py:185-265).  The
    # radius of a splat whose fp32 extent sits on a rounding boundary comes from the fp64 value of the chain (csrc/gspl_device.h:
    # extent_f64); the fixture's radii equal the fp64 oracle's for every splat (tests/test_oracle_golden.py).
    assert np.array_equal(radii.cpu().numpy(), z[p + "radii"])
    assert np.array_equal((radii > 0).cpu().numpy(), z[p + "mask"])
    assert np.array_equal(tiles.cpu().numpy(), z[p + "tiles"])
    np.testing.assert_allclose(xys.detach().cpu().numpy(), z[p + "xys"], rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(depths.detach().cpu().numpy(), z[p + "depths"], rtol=1e-5, atol=1e-6)
    assert_close_scaled(conics.detach().cpu().numpy(), z[p + "conics"], 2e-4, "conics", frac_ok=0.999, rel_all=5e-2)
    np.testing.assert_allclose(comp.detach().cpu().numpy(), z[p + "comp"], rtol=2e-4, atol=1e-6)
    loss = (xys * t32(z[p + "w_xy"])).sum() + (depths * t32(z[p + "w_d"])).sum() + (conics * t32(z[p + "w_c"])).sum() \
        + (comp * t32(z[p + "w_k"])).sum()
    loss.backward()
    for got, name in ((means.grad, "g_means"), (scales.grad, "g_scales"), (quats.grad, "g_quats")):
        assert_close_scaled(got.cpu().numpy(), z[p + name], 1e-4, name, frac_ok=0.995, rel_all=5e-2)      # north_star: 1e-4 rel


def test_projection_known_answer_vector(hip, golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_kat.npz"))
    fx, fy, cx, cy, W, H = z["intr"]
    xys, depths, radii, conics, comp, tiles, cov3d = hip.project_gaussians(
        t32(z["means"]), t32(z["scales"]), 1.0, t32(z["quats"]), t32(z["w2c"]).T.contiguous(),
        float(fx), float(fy), float(cx), float(cy), int(H), int(W), 16)
    assert radii.tolist() == [0, 4, 0, 16783]
    m = (radii > 0).cpu().numpy()
    # the reference's literal expected upper triangles (tests/gaussian_projection_test.py:30-113) and zeros for the culled rows
    upper = cov3d.cpu().numpy().reshape(4, 9)[:, [0, 1, 2, 4, 5, 8]]
    np.testing.assert_allclose(upper[m], z["exp_cov3d_upper_masked"].reshape(int(m.sum()), -1), rtol=2e-4)
    assert not upper[~m].any()
    assert hip.project_gaussians(t32(z["means"]), t32(z["scales"]), 1.0, t32(z["quats"]), t32(z["w2c"]).T.contiguous(),
                                 float(fx), float(fy), float(cx), float(cy), int(H), int(W), 16, return_cov3d=False)[6] is None
    np.testing.assert_allclose(conics.cpu().numpy()[m], z["exp_conics_masked"], rtol=1e-4)
    np.testing.assert_allclose(comp.cpu().numpy()[m], z["exp_comp_masked"], rtol=1e-5)
    assert tiles.cpu().numpy()[m].tolist() == [4, 4346]
    np.testing.assert_allclose(xys.cpu().numpy()[[1, 3]] - 0.5, z["exp_xys_rows13_ndc_convention"], rtol=3e-6, atol=3e-3)


def test_projection_vs_oracle_fp64_batched_cameras(hip):
    """v1 API, C = 3 cameras in one launch, gradients accumulated across cameras."""
    means, scales, quats, _, _ = O.synthetic_scene(5000, seed=3)
    scales = scales * 3
    W, H = 400, 304
    cams = [O.synthetic_camera(W, H, 380.0, 375.0, distance=d) for d in (3.0, 4.0, 5.5)]
    vms = torch.stack([c["world_to_camera"].T for c in cams]).float().to(_dev())
    Ks = torch.stack([torch.tensor([[c["fx"], 0, c["cx"]], [0, c["fy"], c["cy"]], [0, 0, 1.0]]) for c in cams]).float().to(_dev())
    m, s, q = [t.requires_grad_(True) for t in _cuda(means, scales, quats)]
    radii, means2d, depths, conics, comps = hip.fully_fused_projection(m, None, q, s, vms, Ks, W, H, calc_compensations=True)
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(means2d.shape, generator=g), torch.randn(depths.shape, generator=g),
          torch.randn(conics.shape, generator=g) * 0.1, torch.randn(comps.shape, generator=g)]
    (sum((a * w.to(_dev())).sum() for a, w in zip((means2d, depths, conics, comps), ws))).backward()

    md, sd, qd = [t.double().requires_grad_(True) for t in (means, scales, quats)]
    loss = 0
    for ci, c in enumerate(cams):
        xys, dep, rad, con, cmp_, tiles, _, mask, _, _ = O.project_gaussians(
            md, sd, 1.0, qd, c["world_to_camera"].double(), c["fx"], c["fy"], c["cx"], c["cy"], H, W)
        same = rad.numpy() == radii[ci].cpu().numpy()
        assert same.all(), f"camera {ci}: {int((~same).sum())} radii differ from the fp64 oracle's"
        np.testing.assert_allclose(means2d[ci].detach().cpu().numpy()[same], xys.detach().numpy()[same], rtol=1e-5, atol=2e-3)
        assert_close_scaled(conics[ci].detach().cpu().numpy()[same], con.detach().numpy()[same], 2e-4, "conics", 0.999, rel_all=5e-2)
        loss = loss + (xys * ws[0][ci].double()).sum() + (dep * ws[1][ci].double()).sum() \
            + (con * ws[2][ci].double()).sum() + (cmp_ * ws[3][ci].double()).sum()
    loss.backward()
    for got, ref, name in ((m.grad, md.grad, "means"), (s.grad, sd.grad, "scales"), (q.grad, qd.grad, "quats")):
        assert_close_scaled(got.cpu().numpy(), ref.numpy(), 1e-4, name, frac_ok=0.995, rel_all=5e-2)      # north_star: 1e-4 rel


# ---------------------------------------------------------------------------------------------
# spherical harmonics
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("form", ["merged", "decomposed"])
def test_sh_vs_reference_golden(hip, golden_dir, deg, form):
    z = np.load(os.path.join(golden_dir, "ref_sh.npz"))
    K = (deg + 1) ** 2
    dirs = t32(z["dirs"] * 2.5).requires_grad_(True)          # un-normalised on purpose
    w = t32(z["w"])
    if form == "merged":
        c = t32(z["coeffs"][:, :K]).requires_grad_(True)
        rgb = hip.spherical_harmonics(deg, dirs, c)
        (rgb * w).sum().backward()
        gc = c.grad.cpu().numpy()
    else:
        dc = t32(z["coeffs"][:, :1]).requires_grad_(True)
        rest = t32(z["coeffs"][:, 1:K]).requires_grad_(True)
        rgb = hip.spherical_harmonics_decomposed(deg, dirs, dc, rest)
        (rgb * w).sum().backward()
        gc = np.concatenate([dc.grad.cpu().numpy(), rest.grad.cpu().numpy()], axis=1)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), z[f"deg{deg}_rgb"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(gc, z[f"deg{deg}_g_coeffs"], rtol=2e-5, atol=2e-6)
    if deg > 0:
        # golden dirs gradient is w.r.t. unit dirs; ours is w.r.t. 2.5*dirs through the normalisation
        d = torch.from_numpy(z["dirs"]).double().requires_grad_(True)
        cc = torch.from_numpy(z["coeffs"][:, :K]).double()
        (O.eval_sh(deg, cc, d * 2.5) * torch.from_numpy(z["w"]).double()).sum().backward()
        assert_close_scaled(dirs.grad.cpu().numpy(), d.grad.numpy() / 2.5, 2e-4, "v_dirs")


def test_sh_masks_partial_degree_and_fused_clamp(hip):
    n, K = 3001, 16                       # ragged (not a multiple of the 256-row tile)
    g = torch.Generator().manual_seed(1)
    means = torch.randn(n, 3, generator=g)
    center = torch.tensor([0.3, -0.2, 4.0])
    dc = torch.randn(n, 1, 3, generator=g) * 0.5
    rest = torch.randn(n, K - 1, 3, generator=g) * 0.5
    mask = torch.rand(n, generator=g) > 0.3
    w = torch.randn(n, 3, generator=g)
    for deg in (0, 2, 3):
        dcg, rg = [t.requires_grad_(True) for t in _cuda(dc, rest)]
        col = hip.sh_view_colors(deg, means.to(_dev()), center.to(_dev()), dcg, rg, mask.to(_dev()))
        (col * w.to(_dev())).sum().backward()
        dcd, rd = dc.double().requires_grad_(True), rest.double().requires_grad_(True)
        ref = O.sh_colors(deg, torch.cat([dcd, rd], 1), means.double(), center.double(), detach_dirs=True)
        ref = torch.where(mask[:, None], ref, torch.zeros((), dtype=torch.float64))
        (ref * w.double()).sum().backward()
        np.testing.assert_allclose(col.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=3e-6)
        np.testing.assert_allclose(dcg.grad.cpu().numpy(), dcd.grad.numpy(), rtol=2e-5, atol=3e-6)
        np.testing.assert_allclose(rg.grad.cpu().numpy(), rd.grad.numpy(), rtol=2e-5, atol=3e-6)
        assert torch.all(rg.grad[:, (deg + 1) ** 2 - 1:] == 0)          # above the active degree


# ---------------------------------------------------------------------------------------------
# binning: bit-exact
# ---------------------------------------------------------------------------------------------
def _projected_scene(n, W, H, fx, seed=5, scale_mul=3.0, mode=O.MODE_GSPLAT):
    means, scales, quats, opac, shs = O.synthetic_scene(n, seed=seed)
    cam = O.synthetic_camera(W, H, fx)
    res = O.project_gaussians(means, scales * scale_mul, 1.0, quats, cam["world_to_camera"], cam["fx"], cam["fy"],
                              cam["cx"], cam["cy"], H, W)
    return res, opac, shs, cam


@pytest.mark.parametrize("mode", [O.MODE_GSPLAT, O.MODE_INRIA])
@pytest.mark.parametrize("wh", [(320, 240), (333, 211), (16, 16)])
def test_binning_bit_exact(hip, mode, wh):
    W, H = wh
    res, _, _, _ = _projected_scene(4000, W, H, 300.0)
    xys, depths, radii = res[0], res[1], res[2]
    tiles_ref, ids_ref, flat_ref, offs_ref = O.isect_tiles(mode, xys, radii, depths, W, H)
    tw, th = (W + 15) // 16, (H + 15) // 16
    tiles, ids, flat = hip.isect_tiles(xys.to(_dev())[None], radii.to(_dev())[None], depths.to(_dev())[None], 16, tw, th, mode=mode)
    offs = hip.isect_offset_encode(ids, 1, tw, th)
    assert np.array_equal(tiles[0].cpu().numpy(), tiles_ref)
    assert np.array_equal(ids.cpu().numpy(), ids_ref)
    assert np.array_equal(flat.cpu().numpy(), flat_ref)          # stable: ties keep Gaussian order
    assert np.array_equal(offs.reshape(-1).cpu().numpy(), offs_ref)
    # the two-level "depth first" path used by the fused rasterizers must give the very same lists
    flat2, offs2 = hip.bin_gaussians(xys.to(_dev()), depths.to(_dev()), radii.to(_dev()), H, W, 16, mode=mode)
    assert np.array_equal(flat2.cpu().numpy(), flat_ref)
    assert np.array_equal(offs2.cpu().numpy(), offs_ref)


@pytest.mark.parametrize("mode", [O.MODE_GSPLAT, O.MODE_INRIA])
@pytest.mark.parametrize("tile", [8, 32])
@pytest.mark.parametrize("wh", [(320, 240), (333, 211)])
def test_binning_bit_exact_tile_sizes_8_and_32(hip, mode, tile, wh):
    """`block_size` 8 / 32 (gsplat_v1_renderer.py:23-41 passes it as `tile_size`): same keys, same lists as the oracle."""
    W, H = wh
    res, _, _, _ = _projected_scene(4000, W, H, 300.0)
    xys, depths, radii = res[0], res[1], res[2]
    tiles_ref, ids_ref, flat_ref, offs_ref = O.isect_tiles(mode, xys, radii, depths, W, H, block=tile)
    tw, th = (W + tile - 1) // tile, (H + tile - 1) // tile
    tiles, ids, flat = hip.isect_tiles(xys.to(_dev())[None], radii.to(_dev())[None], depths.to(_dev())[None], tile, tw, th, mode=mode)
    offs = hip.isect_offset_encode(ids, 1, tw, th)
    assert np.array_equal(tiles[0].cpu().numpy(), tiles_ref)
    assert np.array_equal(ids.cpu().numpy(), ids_ref)
    assert np.array_equal(flat.cpu().numpy(), flat_ref)
    assert np.array_equal(offs.reshape(-1).cpu().numpy(), offs_ref)
    flat2, offs2 = hip.bin_gaussians(xys.to(_dev()), depths.to(_dev()), radii.to(_dev()), H, W, tile, mode=mode)
    assert np.array_equal(flat2.cpu().numpy(), flat_ref)
    assert np.array_equal(offs2.cpu().numpy(), offs_ref)


def test_binning_two_level_ties_and_empty(hip):
    """Equal depths (ties must stay in Gaussian-id order), all-culled and N == 0 inputs."""
    d = _dev()
    n, W, H = 3000, 200, 120
    g = torch.Generator().manual_seed(2)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W, H])
    radii = torch.randint(0, 30, (n,), generator=g, dtype=torch.int32)
    depths = torch.randint(1, 6, (n,), generator=g).float()          # only 5 distinct depths
    _, _, flat_ref, offs_ref = O.isect_tiles(O.MODE_GSPLAT, xy, radii, depths, W, H)
    flat, offs = hip.bin_gaussians(xy.to(d), depths.to(d), radii.to(d), H, W, 16)
    assert np.array_equal(flat.cpu().numpy(), flat_ref) and np.array_equal(offs.cpu().numpy(), offs_ref)
    flat, offs = hip.bin_gaussians(xy.to(d), depths.to(d), torch.zeros_like(radii).to(d), H, W, 16)
    assert flat.numel() == 0 and int(offs.abs().sum()) == 0
    flat, offs = hip.bin_gaussians(torch.zeros(0, 2, device=d), torch.zeros(0, device=d), torch.zeros(0, dtype=torch.int32, device=d), H, W, 16)
    assert flat.numel() == 0 and offs.numel() == ((W + 15) // 16) * ((H + 15) // 16) and int(offs.abs().sum()) == 0


def test_speculative_emission_recovers_from_a_low_guess(hip):
    """`bin_gaussians` launches the emit kernel with room for the LAST frame's list length before it knows this frame's
    (ops.bin_gaussians_begin).  A frame with far more intersections than the last one must still give the exact lists
    (the emission is repeated), and so must a frame with far fewer."""
    d = _dev()
    W, H = 320, 240
    tw, th = (W + 15) // 16, (H + 15) // 16
    res_small, _, _, _ = _projected_scene(300, W, H, 300.0, seed=11, scale_mul=1.0)
    res_big, _, _, _ = _projected_scene(6000, W, H, 300.0, seed=12, scale_mul=6.0)
    # both orders of the second half: the sort enqueued before the host has the list length (read on the device, checked
    # afterwards) and the round-1 order (wait for the count, then sort)
    for device_side in (True, False):
        hip.DEVICE_SIDE_LIST_LENGTH = device_side
        hip.STATE.capacity.clear()          # (the room is a running maximum per splat, ops._state.ListCapacity: start without history)
        key = (torch.device(d).index, tw, th)
        low_guesses, misses0 = 0, hip.SPECULATION["misses"]
        try:
            for res in (res_small, res_big, res_small, res_big, res_big):
                xys, depths, radii = res[0], res[1], res[2]
                _, _, flat_ref, offs_ref = O.isect_tiles(O.MODE_GSPLAT, xys, radii, depths, W, H)
                room = hip.STATE.capacity.hint(key, xys.shape[0])
                low_guesses += 0 < room < flat_ref.shape[0]
                flat, offs = hip.bin_gaussians(xys.to(d), depths.to(d), radii.to(d), H, W, 16)
                assert np.array_equal(flat.cpu().numpy(), flat_ref) and np.array_equal(offs.cpu().numpy(), offs_ref)
                assert offs.shape == (tw * th,) and offs.is_contiguous() and flat.is_contiguous()
        finally:
            hip.DEVICE_SIDE_LIST_LENGTH = True
        # the big frame after the small one overflows its room and is repeated; the running maximum then covers every later frame
        assert low_guesses == 1 and hip.SPECULATION["misses"] - misses0 == 1, (low_guesses, hip.SPECULATION["misses"] - misses0)
        assert hip._LAST_ISECTS[key] == flat_ref.shape[0]


def test_lazy_lists_composite_before_the_host_has_the_list_length(hip):
    """`bin_gaussians(..., lazy=True)` hands compositing an `ops.LazyLists`: the compositing launch goes out on the capacity-sized
    buffer with the device-side list end BEFORE the host looks at the count (no blocking wait in the frame); a guess that was too
    low repeats emission, sort and that launch.  Frames whose guess is far too low, about right and far too high must equal the
    eager path bit for bit — image, alpha, the lists the object settles to — and their gradients within the atomics' spread; the
    v0 entry point (which bins by itself, lazily) and the hit flags as well."""
    d = _dev()
    W, H, D = 640, 480, 3
    cases = []
    for n, seed, mul in ((400, 31, 1.0), (30000, 32, 8.0), (400, 31, 1.0), (30000, 32, 8.0), (30000, 33, 8.0)):
        res, opac, _, _ = _projected_scene(n, W, H, 600.0, seed=seed, scale_mul=mul)
        g = torch.Generator().manual_seed(seed)
        cases.append(dict(xys=res[0].float().to(d), depths=res[1].float().to(d), radii=res[2].to(d), conics=res[3].float().to(d),
                          op=(opac.reshape(-1) * res[4]).float().to(d), col=torch.rand(n, D, generator=g).to(d)))
    bg = torch.tensor([0.2, 0.4, 0.1], device=d)
    w = torch.randn(H, W, D, generator=torch.Generator().manual_seed(3)).to(d)
    frames0, misses0 = hip.SPECULATION["frames"], hip.SPECULATION["misses"]
    key = (torch.device(d).index, (W + 15) // 16, (H + 15) // 16)
    n_lazy, lazy_misses, prev = 0, 0, (1, 1)
    for c in cases:
        def run(lazy):
            leaves = [c[k].clone().requires_grad_(True) for k in ("xys", "conics", "col", "op")]
            m2, con, col, op = leaves
            flat, offs = hip.bin_gaussians(c["xys"], c["depths"], c["radii"], H, W, 16, conics=c["conics"], opacities=c["op"], lazy=lazy)
            was_lazy = isinstance(flat, hip.LazyLists)
            out, alphas = hip.rasterize_to_pixels(m2, con[None], col[None], op[None], W, H, 16, offs.reshape(1, (H + 15) // 16, (W + 15) // 16), flat,
                                                  backgrounds=bg[None], track_hits=True)
            if was_lazy:
                assert flat.settled
                flat = flat.resolve()[0]
            (out[0] * w).sum().backward()
            return out.detach(), alphas.detach(), flat, offs, [t.grad for t in leaves], m2.has_hit_any_pixels, was_lazy
        o0, a0, f0, of0, g0, h0, _ = run(False)
        # the lazy run starts from the guess the PREVIOUS case left (far too low for a big frame after a small one, far too high the
        # other way round), not from the one the eager run of the same scene has just stored
        hip.STATE.capacity.set(key, *prev)
        lazy_misses += f0.shape[0] > hip.STATE.capacity.hint(key, c["xys"].shape[0])
        prev = (c["xys"].shape[0], f0.shape[0])
        o1, a1, f1, of1, g1, h1, was_lazy = run(True)
        n_lazy += was_lazy
        assert torch.equal(o0, o1) and torch.equal(a0, a1) and torch.equal(f0, f1) and torch.equal(of0, of1) and torch.equal(h0, h1)
        for x, y in zip(g0, g1):
            assert_close_scaled(y.cpu().numpy(), x.cpu().numpy(), 2e-5, "lazy vs eager gradients")
        # v0: bins by itself (lazily) inside the call
        img = hip.rasterize_gaussians(c["xys"], c["depths"], c["radii"], c["conics"], None, c["col"], c["op"][:, None], H, W, 16, bg)
        assert torch.equal(img, o0[0])
    assert n_lazy == len(cases), "the speculative path must have been taken"
    assert lazy_misses >= 2, f"the big frames after the small ones must overflow their room (last: {prev})"
    assert hip.SPECULATION["misses"] - misses0 >= lazy_misses
    assert hip.SPECULATION["frames"] - frames0 == 3 * len(cases)


@pytest.mark.parametrize("mode", [O.MODE_GSPLAT, O.MODE_INRIA])
def test_binning_big_splats(hip, mode):
    """Splats taller than 16 tile rows (radius > 128 px) are ranked by the scan of the counts and dealt out to the workgroups
    of the emit kernel (phase B): lists must stay bit-exact with screen-filling splats mixed in (consecutive in depth, as
    close-ups are), with and without the ellipse culling."""
    d = _dev()
    W, H = 640, 400                              # 40 x 25 tiles
    g = torch.Generator().manual_seed(3)
    n = 5000
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W, H])
    radii = torch.randint(1, 40, (n,), generator=g, dtype=torch.int32)
    depths = torch.rand(n, generator=g) * 5 + 1
    near = torch.argsort(depths)[:150]           # the 150 nearest splats are huge
    radii[near] = torch.randint(140, 900, (150,), generator=g, dtype=torch.int32)
    _, _, flat_ref, offs_ref = O.isect_tiles(mode, xy, radii, depths, W, H)
    flat, offs = hip.bin_gaussians(xy.to(d), depths.to(d), radii.to(d), H, W, 16, mode=mode)
    assert np.array_equal(flat.cpu().numpy(), flat_ref) and np.array_equal(offs.cpu().numpy(), offs_ref)
    # with culling: isotropic conics sized so that the alpha >= 1/255 ellipse is well inside the 3-sigma radius
    sig = radii.float() / 3.0
    conics = torch.stack([1 / sig ** 2, torch.zeros(n), 1 / sig ** 2], 1)
    opac = torch.full((n,), 0.9)
    flat_c, offs_c = hip.bin_gaussians(xy.to(d), depths.to(d), radii.to(d), H, W, 16, mode=mode, conics=conics.to(d), opacities=opac.to(d))
    full = set(zip(np.repeat(np.arange(offs_ref.size), np.diff(np.append(offs_ref, flat_ref.size))).tolist(), flat_ref.tolist()))
    fc, oc = flat_c.cpu().numpy(), offs_c.cpu().numpy()
    culled = list(zip(np.repeat(np.arange(oc.size), np.diff(np.append(oc, fc.size))).tolist(), fc.tolist()))
    assert 0 < len(culled) < len(full) and set(culled) <= full          # a subset of the rect lists, tile by tile
    # depth order inside every tile is kept
    dn = depths.numpy()
    for t in range(0, oc.size, 97):
        seg = fc[oc[t]:(oc[t + 1] if t + 1 < oc.size else fc.size)]
        assert np.all(np.diff(dn[seg]) >= 0)


@pytest.mark.parametrize("n", [1_200_000, 2_500_000])
def test_binning_above_one_million_splats(hip, n):
    """1.2 M splats (several sort workgroups per pass, multi-block scans) with a few screen-filling ones dealt out to the emit
    kernel's workgroups: the two-level lists must equal those of the independent 64-bit (tile | depth) sort behind `isect_tiles`
    (itself checked against the oracle at small sizes).  2.5 M: more than SCAN_RAW_SUMS_BLOCKS (1024) block sums, i.e. the
    three-launch form of the count scan (gspl_sort.h); below that the middle launch is folded into the last one."""
    d = _dev()
    W, H = 640, 400
    g = torch.Generator().manual_seed(8)
    xy = (torch.rand(n, 2, generator=g) * torch.tensor([W, H])).to(d)
    radii = torch.randint(0, 3, (n,), generator=g, dtype=torch.int32)
    radii[:50] = torch.randint(150, 700, (50,), generator=g, dtype=torch.int32)      # a few screen-filling ones
    radii = radii.to(d)
    depths = (torch.rand(n, generator=g) * 9 + 0.5).to(d)
    tw, th = (W + 15) // 16, (H + 15) // 16
    _, ids, flat_ref = hip.isect_tiles(xy[None], radii[None], depths[None], 16, tw, th)
    offs_ref = hip.isect_offset_encode(ids, 1, tw, th).reshape(-1)
    flat, offs = hip.bin_gaussians(xy, depths, radii, H, W, 16)
    assert torch.equal(flat, flat_ref) and torch.equal(offs, offs_ref)


@pytest.mark.parametrize("mode", [O.MODE_GSPLAT, O.MODE_INRIA])
def test_tile_culling_is_lossless(hip, mode):
    """Exact ellipse-vs-tile culling in the list-only binning path: shorter lists, every tile's list an
    order-preserving subsequence of the un-culled one, bit-identical image, same gradients."""
    from gspl_amd import _lib as L
    W, H, D = 333, 211, 3
    xys, conics, colors, op, bg, flat_ref, offs_ref = _composite_case(mode, D, W, H, n=8000, seed=17)
    res, _, _, _ = _projected_scene(8000, W, H, 260.0, seed=17)
    depths, radii = res[1], res[2]
    d = _dev()
    c = lambda a: torch.as_tensor(a).contiguous().to(d)
    dxy, dcon, dcol, dop, dbg = c(xys), c(conics), c(colors), c(op), c(bg)
    flat0, offs0 = hip.bin_gaussians(dxy, c(depths), c(radii), H, W, 16, mode=mode)
    flat1, offs1 = hip.bin_gaussians(dxy, c(depths), c(radii), H, W, 16, mode=mode, conics=dcon, opacities=dop)
    assert np.array_equal(flat0.cpu().numpy(), flat_ref)
    assert 0.3 * flat0.numel() < flat1.numel() < 0.9 * flat0.numel()
    f0, f1, o0, o1 = flat0.cpu().numpy(), flat1.cpu().numpy(), offs0.cpu().numpy(), offs1.cpu().numpy()
    n_tiles = o0.shape[0]
    for t in range(0, n_tiles, 7):
        a = f0[o0[t]:(o0[t + 1] if t + 1 < n_tiles else len(f0))]
        b = f1[o1[t]:(o1[t + 1] if t + 1 < n_tiles else len(f1))]
        it = iter(a.tolist())
        assert all(any(x == y for y in it) for x in b.tolist()), "culled list is not a subsequence"
    out0, a0, T0, l0 = hip_composite_fwd(mode, dxy, dcon, dcol, dop, dbg, W, H, offs0, flat0)
    out1, a1, T1, l1 = hip_composite_fwd(mode, dxy, dcon, dcol, dop, dbg, W, H, offs1, flat1)
    assert torch.equal(out0, out1) and torch.equal(T0, T1)
    g = torch.Generator().manual_seed(3)
    v_out = c(torch.randn(H, W, D, generator=g))
    g0 = hip_composite_bwd(mode, dxy, dcon, dcol, dop, dbg, W, H, offs0, flat0, T0, l0, v_out)
    g1 = hip_composite_bwd(mode, dxy, dcon, dcol, dop, dbg, W, H, offs1, flat1, T1, l1, v_out)
    for k in ("v_means2d", "v_conics", "v_colors", "v_opacities"):
        assert_close_scaled(g1[k].cpu().numpy(), g0[k].cpu().numpy(), 1e-5, k)
    # and against the oracle on the culled lists
    out_ref, _, _, frag = O.composite_fwd(mode, xys, conics, colors, op, bg, W, H, offs_ref, flat_ref)
    assert np.abs(out1.cpu().numpy() - out_ref)[frag == 0].max() <= 1e-5


@pytest.mark.parametrize("mode", [O.MODE_GSPLAT, O.MODE_INRIA])
@pytest.mark.parametrize("seed,wh", [(1, (640, 400)), (2, (333, 211)), (3, (1000, 48)), (4, (48, 700))])
def test_culling_is_lossless_with_every_span_class(hip, seed, wh, mode):
    """The emit kernel has three classes of splats — up to 8 tile rows (one span record), 9-16 rows (extension record) and
    taller or very wide ones (ranked by the scan, emitted in phase B) — and anisotropic, rotated footprints exercise the
    per-row column spans.  Whatever the mix: un-culled lists bit-exact against the oracle, culled lists order-preserving
    subsequences, and compositing the culled lists gives the bit-identical image."""
    W, H = wh
    d = _dev()
    g = torch.Generator().manual_seed(seed)
    n = 3000
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W * 1.2, H * 1.2]) - torch.tensor([W * 0.1, H * 0.1])      # some off-screen centres
    # standard deviations along the principal axes (pixels) from three scales, random orientation
    cls = torch.randint(0, 3, (n,), generator=g)
    s_major = torch.where(cls == 0, torch.rand(n, generator=g) * 12 + 0.5, torch.where(cls == 1, torch.rand(n, generator=g) * 30 + 25,
                                                                                            torch.rand(n, generator=g) * 250 + 60))
    s_minor = s_major * (torch.rand(n, generator=g) * 0.9 + 0.1)
    th = torch.rand(n, generator=g) * 3.14159
    cx, sx = torch.cos(th), torch.sin(th)
    cov_a = cx * cx * s_major ** 2 + sx * sx * s_minor ** 2
    cov_b = cx * sx * (s_major ** 2 - s_minor ** 2)
    cov_c = sx * sx * s_major ** 2 + cx * cx * s_minor ** 2
    det = cov_a * cov_c - cov_b * cov_b
    conics = torch.stack([cov_c / det, -cov_b / det, cov_a / det], 1)
    radii = torch.ceil(3 * s_major).to(torch.int32)
    depths = torch.rand(n, generator=g) * 8 + 0.2
    opac = torch.rand(n, generator=g) * 0.98 + 0.01
    colors = torch.rand(n, 3, generator=g)
    _, _, flat_ref, offs_ref = O.isect_tiles(mode, xy, radii, depths, W, H)
    c = lambda t: t.contiguous().to(d)
    flat0, offs0 = hip.bin_gaussians(c(xy), c(depths), c(radii), H, W, 16, mode=mode)
    assert np.array_equal(flat0.cpu().numpy(), flat_ref) and np.array_equal(offs0.cpu().numpy(), offs_ref)
    flat1, offs1 = hip.bin_gaussians(c(xy), c(depths), c(radii), H, W, 16, mode=mode, conics=c(conics), opacities=c(opac))
    assert flat1.numel() < flat0.numel()
    f0, f1, o0, o1 = flat0.cpu().numpy(), flat1.cpu().numpy(), offs0.cpu().numpy(), offs1.cpu().numpy()
    nt = o0.shape[0]
    for t in range(nt):
        a = f0[o0[t]:(o0[t + 1] if t + 1 < nt else len(f0))]
        b = f1[o1[t]:(o1[t + 1] if t + 1 < nt else len(f1))]
        it = iter(a.tolist())
        assert all(any(x == y for y in it) for x in b.tolist()), "culled list is not a subsequence"
    bg = torch.tensor([0.1, 0.2, 0.3])
    out0, a0, T0, l0 = hip_composite_fwd(mode, c(xy), c(conics), c(colors), c(opac), c(bg), W, H, offs0, flat0)
    out1, a1, T1, l1 = hip_composite_fwd(mode, c(xy), c(conics), c(colors), c(opac), c(bg), W, H, offs1, flat1)
    assert torch.equal(out0, out1) and torch.equal(T0, T1)


def test_binning_empty_inputs(hip):
    d = _dev()
    tiles, ids, flat = hip.isect_tiles(torch.zeros(1, 7, 2, device=d), torch.zeros(1, 7, dtype=torch.int32, device=d),
                                       torch.zeros(1, 7, device=d), 16, 4, 3)
    assert ids.numel() == 0 and flat.numel() == 0 and int(tiles.sum()) == 0
    offs = hip.isect_offset_encode(ids, 1, 4, 3)
    assert offs.shape == (1, 3, 4) and int(offs.abs().sum()) == 0
    tiles, ids, flat = hip.isect_tiles(torch.zeros(1, 0, 2, device=d), torch.zeros(1, 0, dtype=torch.int32, device=d),
                                       torch.zeros(1, 0, device=d), 16, 4, 3)
    assert ids.numel() == 0


# ---------------------------------------------------------------------------------------------
# compositing: forward <= 1e-5 abs / pixel, backward <= 1e-4 rel
# ---------------------------------------------------------------------------------------------
def _composite_case(mode, D, W, H, n=6000, seed=11, big=False, tile=16):
    res, opac, shs, cam = _projected_scene(n, W, H, 260.0, seed=seed, scale_mul=(12.0 if big else 3.0))
    xys, depths, radii, conics, comp = res[0], res[1], res[2], res[3], res[4]
    g = torch.Generator().manual_seed(seed)
    colors = torch.rand(n, D, generator=g)
    op = (opac.reshape(-1) * comp).float()
    op[:20] = 1.0                                    # exercise the clamp
    bg = torch.rand(D, generator=g)
    if mode == O.MODE_INRIA:
        xys = xys - 0.5
    tiles, ids, flat, offs = O.isect_tiles(mode, xys, radii, depths, W, H, block=tile)
    return xys.float(), conics.float(), colors, op, bg, flat, offs


@pytest.mark.parametrize("mode", [O.MODE_GSPLAT, O.MODE_INRIA])
@pytest.mark.parametrize("D,wh,big,tile", [(3, (320, 240), False, 16), (1, (333, 211), False, 16), (4, (160, 96), True, 16), (8, (97, 65), False, 16),
                                           (2, (64, 48), True, 16),
                                           # lists cut on 8- and 32-pixel tiles (`block_size` of the reference's renderers)
                                           (3, (320, 240), False, 8), (3, (333, 211), True, 8), (1, (97, 65), False, 8),
                                           (3, (320, 240), False, 32), (4, (333, 211), True, 32), (8, (97, 65), False, 32)])
def test_composite_fwd_bwd_vs_oracle(hip, mode, D, wh, big, tile):
    from gspl_amd import _lib as L
    W, H = wh
    xys, conics, colors, op, bg, flat, offs = _composite_case(mode, D, W, H, big=big, tile=tile)
    out_ref, alpha_ref, last_ref, frag = O.composite_fwd(mode, xys, conics, colors, op, bg, W, H, offs, flat, tile=tile)
    c = lambda a: torch.as_tensor(a).contiguous().to(_dev())
    dxy, dcon, dcol, dop, dbg, dflat, doffs = c(xys), c(conics), c(colors), c(op), c(bg), c(flat), c(offs)
    out, alphas, final_T, last = hip_composite_fwd(mode, dxy, dcon, dcol, dop, dbg, W, H, doffs, dflat, tile=tile)
    ok = frag == 0
    assert ok.mean() > 0.995
    assert np.abs(out.cpu().numpy() - out_ref)[ok].max() <= 1e-5
    assert np.abs(alphas.cpu().numpy() - alpha_ref)[ok].max() <= 1e-5
    assert np.abs(final_T.cpu().numpy() - (1.0 - alpha_ref))[ok].max() <= 1e-5
    assert np.array_equal(last.cpu().numpy()[ok], last_ref[ok])

    g = torch.Generator().manual_seed(4)
    v_out = torch.randn(H, W, D, generator=g)
    v_alpha = torch.randn(H, W, generator=g)
    # Pixels with a decision (1/255 skip, transmittance stop, clamp) within 2e-5 of its threshold — where fp32 and fp64 may decide
    # differently — carry NO loss, on both sides; then nothing is excused: every splat, every gradient element within 1e-4.
    fragile = torch.from_numpy(frag != 0)
    v_out[fragile] = 0.0
    v_alpha[fragile] = 0.0
    got = hip_composite_bwd(mode, dxy, dcon, dcol, dop, dbg, W, H, doffs, dflat, final_T, last, c(v_out), c(v_alpha), absgrad=True, tile=tile)
    # the oracle differentiates the discrete path the GPU took (its alphas / last_ids)
    ref = O.composite_bwd(mode, xys, conics, colors, op, bg, W, H, offs, flat, 1.0 - final_T.cpu().double().numpy(),
                          last.cpu().numpy(), v_out.double().numpy(), v_alpha.double().numpy(), fragile_px=frag, absgrad=True, tile=tile)
    for k in ("v_means2d", "v_means2d_abs", "v_conics", "v_colors", "v_opacities"):
        assert_close_scaled(got[k].cpu().numpy(), ref[k], 1e-4, f"{k} mode={mode} D={D} tile={tile}", frac_ok=1.0)
    keep = ref["fragile_g"] == 0          # (only for the hit flags below: a splat seen by a fragile pixel alone has no oracle gradient)
    # ... and so are the fragile pixels of the forward
    assert np.abs(out.cpu().numpy() - out_ref).max() <= 1e-3
    # hit flags (the fork's `has_hit_any_pixels`): a splat some pixel composited has a colour weight alpha*T > 0 there, so with a
    # random dL/dout its oracle colour gradient is non-zero — and zero for every splat no pixel took
    hit = got["hit"].cpu().numpy().astype(bool)
    ref_hit = (np.abs(ref["v_colors"]) > 0).any(axis=1)
    assert np.array_equal(hit[keep], ref_hit[keep])
    assert 0 < hit.sum() < hit.size or not big


def test_composite_layout_chw_equals_hwc(hip):
    from gspl_amd import _lib as L
    W, H, D = 130, 70, 3
    xys, conics, colors, op, bg, flat, offs = _composite_case(O.MODE_INRIA, D, W, H)
    c = lambda a: torch.as_tensor(a).contiguous().to(_dev())
    args = (c(xys), c(conics), c(colors), c(op), c(bg), W, H, c(offs), c(flat))
    o1, a1, _, l1 = hip_composite_fwd(O.MODE_INRIA, *args, layout=L.GSPL_LAYOUT_HWC)
    o2, a2, _, l2 = hip_composite_fwd(O.MODE_INRIA, *args, layout=L.GSPL_LAYOUT_CHW)
    assert torch.equal(o1.permute(2, 0, 1), o2) and torch.equal(a1, a2) and torch.equal(l1, l2)


def test_composite_no_intersections_gives_background(hip):
    d = _dev()
    W, H = 50, 20
    offs = torch.zeros(((W + 15) // 16) * ((H + 15) // 16), dtype=torch.int32, device=d)
    bg = torch.tensor([0.1, 0.7, 0.3], device=d)
    out, alphas, _, last = hip_composite_fwd(O.MODE_GSPLAT, torch.zeros(0, 2, device=d), torch.zeros(0, 3, device=d),
                                          torch.zeros(0, 3, device=d), torch.zeros(0, device=d), bg, W, H, offs,
                                          torch.zeros(0, dtype=torch.int32, device=d))
    assert torch.all(out == bg) and torch.all(alphas == 0) and torch.all(last == 0)


def test_rasterize_to_pixels_wrapper_channels_and_absgrad(hip):
    """The reference's v1 call shape ([1,N,*]), a 7-channel feature stack (rgb+depth+normal,
    gsplat_v1_renderer.py:226-285 -> zero-padded to 8) and the `.absgrad` side channel."""
    W, H, D = 150, 100, 7
    xys, conics, colors, op, bg, flat, offs = _composite_case(O.MODE_GSPLAT, D, W, H, n=3000)
    means2d = xys.to(_dev()).requires_grad_(True)
    col = colors.to(_dev()).requires_grad_(True)
    opd = op.to(_dev()).requires_grad_(True)
    out, alphas = hip.rasterize_to_pixels(means2d, conics.to(_dev())[None], col[None], opd[None], W, H, 16,
                                          torch.as_tensor(offs).to(_dev()).reshape(1, (H + 15) // 16, (W + 15) // 16),
                                          torch.as_tensor(flat).to(_dev()), backgrounds=bg.to(_dev())[None], absgrad=True)
    assert out.shape == (1, H, W, D) and alphas.shape == (1, H, W, 1)
    ref, aref, _, frag = O.composite_fwd(O.MODE_GSPLAT, xys, conics, colors, op, bg, W, H, offs, flat)
    ok = frag == 0
    assert np.abs(out[0].detach().cpu().numpy() - ref)[ok].max() <= 1e-5
    g = torch.Generator().manual_seed(8)
    w = torch.randn(H, W, D, generator=g).to(_dev())
    (out[0] * w).sum().backward()
    assert hasattr(means2d, "absgrad") and means2d.absgrad.shape == (xys.shape[0], 2)
    assert torch.all(means2d.absgrad >= means2d.grad.abs() - 1e-6)
    assert col.grad.shape == (xys.shape[0], D)


@pytest.mark.parametrize("tile", [8, 32])
def test_rasterize_ops_with_tile_sizes_8_and_32(hip, tile):
    """The op-level entry points with lists cut on 8- / 32-pixel tiles: `rasterize_to_pixels` (v1, gsplat_v1_renderer.py:588-601) on
    the oracle's lists and `rasterize_gaussians` (v0, gsplat_renderer.py:86-99) binning by itself, forward against the oracle and
    the packed backward (the path the autograd wrappers take) against the separate-array one."""
    W, H, D = 210, 130, 3
    xys, conics, colors, op, bg, flat, offs = _composite_case(O.MODE_GSPLAT, D, W, H, n=5000, tile=tile)
    ref, aref, _, frag = O.composite_fwd(O.MODE_GSPLAT, xys, conics, colors, op, bg, W, H, offs, flat, tile=tile)
    ok = frag == 0
    tw, th = (W + tile - 1) // tile, (H + tile - 1) // tile
    leaves = [t.to(_dev()).requires_grad_(True) for t in (xys, conics, colors, op)]
    m2, con, col, opd = leaves
    out, alphas = hip.rasterize_to_pixels(m2, con[None], col[None], opd[None], W, H, tile,
                                          torch.as_tensor(offs).to(_dev()).reshape(1, th, tw), torch.as_tensor(flat).to(_dev()),
                                          backgrounds=bg.to(_dev())[None], absgrad=True)
    assert np.abs(out[0].detach().cpu().numpy() - ref)[ok].max() <= 1e-5
    assert np.abs(alphas[0, ..., 0].detach().cpu().numpy() - aref)[ok].max() <= 1e-5
    g = torch.Generator().manual_seed(8)
    w = torch.randn(H, W, D, generator=g)
    w[torch.from_numpy(frag != 0)] = 0.0
    (out[0] * w.to(_dev())).sum().backward()
    # the oracle differentiates the path the GPU took
    c = lambda a: torch.as_tensor(a).contiguous().to(_dev())
    o2, a2, final_T, last = hip_composite_fwd(O.MODE_GSPLAT, c(xys), c(conics), c(colors), c(op), c(bg), W, H, c(offs), c(flat), tile=tile)
    gref = O.composite_bwd(O.MODE_GSPLAT, xys, conics, colors, op, bg, W, H, offs, flat, 1.0 - final_T.cpu().double().numpy(),
                           last.cpu().numpy(), w.double().numpy(), None, fragile_px=frag, absgrad=True, tile=tile)
    for t, k in zip(leaves, ("v_means2d", "v_conics", "v_colors", "v_opacities")):
        assert_close_scaled(t.grad.cpu().numpy().reshape(gref[k].shape), gref[k], 1e-4, f"{k} tile={tile}", frac_ok=1.0)
    assert_close_scaled(m2.absgrad.cpu().numpy(), gref["v_means2d_abs"], 1e-4, f"absgrad tile={tile}", frac_ok=1.0)
    # v0 entry point: bins by itself (exact row spans, the culled lists) -> the same image within the fp32 bar
    res, _, _, _ = _projected_scene(5000, W, H, 260.0, seed=11, scale_mul=3.0)
    radii, depths = res[2], res[1]
    img = hip.rasterize_gaussians(c(xys), c(depths), c(radii), c(conics), None, c(colors), c(op)[:, None], H, W, tile, c(bg))
    assert np.abs(img.cpu().numpy() - ref)[ok].max() <= 1e-5


# ---------------------------------------------------------------------------------------------
# end-to-end pipelines through the autograd wrappers
# ---------------------------------------------------------------------------------------------
def _e2e_scene(n=8000, W=320, H=208, deg=3, seed=21):
    means, scales, quats, opac, shs = O.synthetic_scene(n, seed=seed, sh_degree=deg)
    scales = scales * 4
    cam = O.synthetic_camera(W, H, 300.0, 295.0)
    g = torch.Generator().manual_seed(seed)
    wimg = torch.randn(3, H, W, generator=g)
    bg = torch.tensor([0.25, 0.5, 0.125])
    return means, scales, quats, opac, shs, cam, wimg, bg


def test_end_to_end_gsplat_api(hip):
    means, scales, quats, opac, shs, cam, wimg, bg = _e2e_scene()
    W, H = cam["width"], cam["height"]
    leaves = [t.requires_grad_(True) for t in _cuda(means, scales, quats, opac, shs)]
    m, s, q, o, c = leaves
    vm, K = _cam_tensors(cam)
    xys, depths, radii, conics, comp, tiles, _ = hip.project_gaussians(m, s, 1.0, q, vm[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, 16)
    rgbs = hip.sh_view_colors(3, m, cam["camera_center"].to(_dev()), c, None, radii > 0)
    img = hip.rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, o * comp[:, None], H, W, 16, bg.to(_dev()))
    render = img.permute(2, 0, 1)
    (render * wimg.to(_dev())).sum().backward()

    dl = [t.double().requires_grad_(True) for t in (means, scales, quats, opac, shs)]
    r = O.render_gsplat(*dl, 3, cam["world_to_camera"].double(), cam["fx"], cam["fy"], cam["cx"], cam["cy"], W, H,
                        bg.double(), cam["camera_center"].double())
    (r["render"] * wimg.double()).sum().backward()
    # free-running, with ATTRIBUTION: every pixel the oracle does not flag within 1e-5 (a flagged one within one 8-bit
    # step); every gradient element beyond the tolerance belongs to a splat that a flagged decision reaches (hip_helpers.fragile_rows)
    assert np.array_equal((radii > 0).cpu().numpy(), r["mask"].numpy())
    assert_pipeline_attributed(O.MODE_GSPLAT, r, W, H, bg.double(), render.detach().cpu().numpy(),
                               [(name, got.grad.cpu().numpy(), ref.grad.numpy()) for got, ref, name in zip(leaves, dl, ("means", "scales", "quats", "opacities", "shs"))],
                               gpu_radii=radii)


def test_end_to_end_inria_api(hip):
    means, scales, quats, opac, shs, cam, wimg, bg = _e2e_scene(seed=22)
    W, H = cam["width"], cam["height"]
    leaves = [t.requires_grad_(True) for t in _cuda(means, scales, quats, opac, shs)]
    m, s, q, o, c = leaves
    settings = hip.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(_dev()), scale_modifier=1.0,
        viewmatrix=cam["world_to_camera"].to(_dev()), projmatrix=cam["full_projection"].to(_dev()), sh_degree=3,
        campos=cam["camera_center"].to(_dev()))
    screen = torch.zeros_like(m, requires_grad=True)
    render, radii = hip.GaussianRasterizer(settings)(means3D=m, means2D=screen, opacities=o, shs=c, scales=s, rotations=q)
    (render * wimg.to(_dev())).sum().backward()

    dl = [t.double().requires_grad_(True) for t in (means, scales, quats, opac, shs)]
    r = O.render_inria(*dl, 3, cam["world_to_camera"].double(), cam["full_projection"].double(), cam["camera_center"].double(),
                       cam["tanfovx"], cam["tanfovy"], W, H, bg.double())
    (r["render"] * wimg.double()).sum().backward()
    assert np.mean(radii.cpu().numpy() == r["radii"].numpy()) > 0.999
    # viewspace gradient in Inria units: pixel gradient * 0.5 * (W, H)
    ref_ndc = r["xy"].grad.numpy() * np.array([0.5 * W, 0.5 * H])
    assert_pipeline_attributed(O.MODE_INRIA, r, W, H, bg.double(), render.detach().cpu().numpy(),
                               [(name, got.grad.cpu().numpy(), ref.grad.numpy()) for got, ref, name in zip(leaves, dl, ("means", "scales", "quats", "opacities", "shs"))]
                               + [("viewspace_points.grad", screen.grad[:, :2].cpu().numpy(), ref_ndc)], opacities=dl[3], gpu_radii=radii)
    assert torch.all(screen.grad[:, 2] == 0)


# ---------------------------------------------------------------------------------------------
# BASELINE size (1920x1080, 1 M Gaussians): size-independent properties
# ---------------------------------------------------------------------------------------------
def test_full_size_properties(hip):
    N, W, H = 1_000_000, 1920, 1080
    means, scales, quats, opac, shs = O.synthetic_scene(N, seed=42)
    cam = O.synthetic_camera(W, H, 1600.0)
    m, s, q, o, c = _cuda(means, scales, quats, opac, shs)
    vm, K = _cam_tensors(cam)
    xys, depths, radii, conics, comp, tiles, _ = hip.project_gaussians(m, s, 1.0, q, vm[:3], 1600.0, 1600.0, 960.0, 540.0, H, W, 16)
    tw, th = 120, 68
    tiles_pg, ids, flat = hip.isect_tiles(xys[None], radii[None], depths[None], 16, tw, th)
    offs = hip.isect_offset_encode(ids, 1, tw, th)
    I = ids.shape[0]
    assert I == int(tiles.sum()) and 10_000_000 < I < 20_000_000          # survey measured 13.8 M
    assert bool(torch.all(ids[1:] >= ids[:-1]))                            # sortedness
    o_flat = offs.reshape(-1)
    assert bool(torch.all(o_flat[1:] >= o_flat[:-1])) and int(o_flat[0]) == 0 and int(o_flat[-1]) <= I
    # every (tile id of key) matches the offsets partition: checksum of per-tile counts
    counts = torch.diff(torch.cat([o_flat, torch.tensor([I], device=o_flat.device, dtype=o_flat.dtype)]))
    assert int(counts.sum()) == I
    assert torch.equal(torch.bincount(flat, minlength=N).to(torch.int32), tiles)   # each splat appears tiles_hit times

    rgbs = hip.sh_view_colors(3, m, cam["camera_center"].to(_dev()), c, None, radii > 0)
    op = (o.reshape(-1) * comp)
    col2 = torch.rand_like(rgbs)
    f = lambda colors: hip.rasterize_to_pixels(xys, conics[None], colors[None], op[None], W, H, 16, offs, flat)
    img1, a1 = f(rgbs)
    img2, a2 = f(col2)
    img12, a12 = f(rgbs + col2)
    assert float(a1.min()) >= 0 and float(a1.max()) <= 1.0
    assert torch.equal(a1, a2)                                           # alpha does not depend on colour
    assert float((img12 - img1 - img2).abs().max()) <= 2e-5              # linear in colour (background 0)
    # adjoint identity: <render(colors), w> == <colors, v_colors>
    colg = rgbs.clone().requires_grad_(True)
    img, _ = f(colg)
    w = torch.randn_like(img)
    (img * w).sum().backward()
    lhs = float((img.detach().double() * w.double()).sum())
    rhs = float((colg.grad.double() * rgbs.double()).sum())
    # both sides are sums of ~6 M signed terms: the yardstick is the sum of their magnitudes (|lhs| itself can cancel to ~0)
    scale = float((img.detach().double().abs() * w.double().abs()).sum())
    assert abs(lhs - rhs) <= 1e-6 * scale, (lhs, rhs, scale)
    # idempotence / determinism of the forward
    img1b, _ = f(rgbs)
    assert torch.equal(img1, img1b)


# ---------------------------------------------------------------------------------------------
# Inria API: optional inputs of the reference's GaussianRasterizer call (vanilla_renderer.py:81-120)
# ---------------------------------------------------------------------------------------------
def _inria_settings(hip, cam, bg, W, H, scale_modifier=1.0, deg=3):
    return hip.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(_dev()), scale_modifier=scale_modifier,
        viewmatrix=cam["world_to_camera"].to(_dev()), projmatrix=cam["full_projection"].to(_dev()), sh_degree=deg,
        campos=cam["camera_center"].to(_dev()))


def test_inria_api_precomputed_cov3d_and_colors_and_scale_modifier(hip):
    means, scales, quats, opac, shs, cam, wimg, bg = _e2e_scene(n=4000, seed=23)
    W, H = cam["width"], cam["height"]
    mod = 1.7
    # (a) cov3D_precomp == scales/rotations path with the same modifier; gradients reach cov3D
    m, s, q, o, c = [t.requires_grad_(True) for t in _cuda(means, scales, quats, opac, shs)]
    r1, rad1 = hip.GaussianRasterizer(_inria_settings(hip, cam, bg, W, H, mod))(
        means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, shs=c, scales=s, rotations=q)
    cov = O.cov3d_from_scale_rot(scales.double(), mod, quats.double())
    cov6 = torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1).float().to(_dev()).requires_grad_(True)
    m2, o2, c2 = [t.requires_grad_(True) for t in _cuda(means, opac, shs)]
    r2, rad2 = hip.GaussianRasterizer(_inria_settings(hip, cam, bg, W, H, mod))(
        means3D=m2, means2D=torch.zeros_like(m2, requires_grad=True), opacities=o2, shs=c2, cov3D_precomp=cov6)
    assert torch.equal(rad1, rad2) or (rad1 == rad2).float().mean() > 0.999
    assert float((r1 - r2).abs().max()) <= 2e-5
    (r2 * wimg.to(_dev())).sum().backward()
    (r1 * wimg.to(_dev())).sum().backward()
    assert cov6.grad is not None and torch.isfinite(cov6.grad).all() and float(cov6.grad.abs().sum()) > 0
    # chain rule check: dL/dscales through cov6 equals the direct path
    sd, qd = scales.double().requires_grad_(True), quats.double().requires_grad_(True)
    cv = O.cov3d_from_scale_rot(sd, mod, qd)
    cv6 = torch.stack([cv[:, 0, 0], cv[:, 0, 1], cv[:, 0, 2], cv[:, 1, 1], cv[:, 1, 2], cv[:, 2, 2]], -1)
    (cv6 * cov6.grad.cpu().double()).sum().backward()
    assert_close_scaled(sd.grad.numpy(), s.grad.cpu().numpy(), 2e-4, "scales via cov3D_precomp", frac_ok=0.995, rel_all=0.5)
    assert_close_scaled(qd.grad.numpy(), q.grad.cpu().numpy(), 2e-4, "quats via cov3D_precomp", frac_ok=0.995, rel_all=0.5)

    # (b) colors_precomp == SH evaluated outside; gradient flows to the colours
    rgbs = O.sh_colors(3, shs.double(), means.double(), cam["camera_center"].double(), detach_dirs=True).float()
    col = rgbs.to(_dev()).requires_grad_(True)
    m3, s3, q3, o3 = [t.requires_grad_(True) for t in _cuda(means, scales, quats, opac)]
    r3, _ = hip.GaussianRasterizer(_inria_settings(hip, cam, bg, W, H, mod))(
        means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), opacities=o3, colors_precomp=col, scales=s3, rotations=q3)
    assert float((r3 - r1.detach()).abs().max()) <= 2e-5
    (r3 * wimg.to(_dev())).sum().backward()
    assert col.grad is not None and float(col.grad.abs().sum()) > 0
    with pytest.raises(Exception):
        hip.GaussianRasterizer(_inria_settings(hip, cam, bg, W, H))(means3D=m3, means2D=None, opacities=o3, shs=c, colors_precomp=col,
                                                                    scales=s3, rotations=q3)


@pytest.mark.parametrize("fused", [True, False])
def test_inria_rasterizer_reads_shs_dc_and_shs_rest_in_place(hip, fused):
    """`GaussianRasterizer(shs=shs_dc, shs_rest=shs_rest)` — the model's two SH parameters as stored (gaussian.py:218-254) — against
    `shs=torch.cat((shs_dc, shs_rest), 1)` (what the reference passes, vanilla_renderer.py:99-109): same image bit for bit, the
    coefficient gradients equal to the slices of the merged gradient (the SH backward has no atomics: bit for bit as well),
    for a partial active degree too."""
    means, scales, quats, opac, shs, cam, wimg, bg = _e2e_scene(n=6000)
    W, H = cam["width"], cam["height"]
    old = hip.FUSED_INRIA
    hip.FUSED_INRIA = fused
    try:
        for deg in (3, 1, 0):
            settings = hip.GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(_dev()), scale_modifier=1.0,
                viewmatrix=cam["world_to_camera"].to(_dev()), projmatrix=cam["full_projection"].to(_dev()), sh_degree=deg,
                campos=cam["camera_center"].to(_dev()))
            rast = hip.GaussianRasterizer(settings)
            m, s, q, o = [t.requires_grad_(True) for t in _cuda(means, scales, quats, opac)]
            merged = shs.float().to(_dev()).requires_grad_(True)
            img0, r0 = rast(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, shs=merged, scales=s, rotations=q)
            (img0 * wimg.to(_dev())).sum().backward()
            g_means0 = m.grad.clone()
            m.grad = None
            dc = shs[:, :1].float().contiguous().to(_dev()).requires_grad_(True)
            rest = shs[:, 1:].float().contiguous().to(_dev()).requires_grad_(True)
            img1, r1 = rast(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, shs=dc, shs_rest=rest, scales=s, rotations=q)
            assert torch.equal(img0, img1) and torch.equal(r0, r1)
            (img1 * wimg.to(_dev())).sum().backward()
            # The per-splat gradients the SH / projection backward consume come out of float atomics whose order differs between two
            # launches: compare scaled, within the spread tests/test_backward_spread.py allows for the compositing gradients (2e-5)
            # and — for the means, behind the conic -> cov2D -> cov3D chain that amplifies a last-bit difference of a near-degenerate
            # splat — within 1e-4 (measured over fresh processes: up to 1.1e-5, tools/diag/r03rccl2.py; one element of 18 000 at
            # 1.04e-5 once in eight runs of this file, profiles/r04_flaky_v_means.txt: a tolerance of 1e-5 was inside the spread).
            assert_close_scaled(dc.grad.cpu().numpy(), merged.grad[:, :1].cpu().numpy(), 2e-5, f"v_shs_dc deg={deg}")
            assert_close_scaled(rest.grad.cpu().numpy(), merged.grad[:, 1:].cpu().numpy(), 2e-5, f"v_shs_rest deg={deg}")
            assert_close_scaled(m.grad.cpu().numpy(), g_means0.cpu().numpy(), 1e-4, f"v_means deg={deg}")
            # ... and with the compositing backward's rows added in list order (`set_deterministic`: no atomics-ordered sum left on
            # this path) the two forms give the SAME BITS, for every gradient
            was = hip.set_deterministic(True)
            try:
                grads = []
                for form in ("merged", "split"):
                    for t in (m, s, q, o, merged, dc, rest):
                        t.grad = None
                    kw = dict(shs=merged) if form == "merged" else dict(shs=dc, shs_rest=rest)
                    img, _ = rast(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, scales=s, rotations=q, **kw)
                    (img * wimg.to(_dev())).sum().backward()
                    sh_grad = merged.grad if form == "merged" else torch.cat((dc.grad, rest.grad), 1)
                    grads.append([m.grad.clone(), s.grad.clone(), q.grad.clone(), o.grad.clone(), sh_grad.clone()])
                for a, b, name in zip(grads[0], grads[1], ("means", "scales", "quats", "opacities", "shs")):
                    assert torch.equal(a, b), f"deterministic mode, deg={deg}: {name} differs between the merged and the split form"
            finally:
                hip.set_deterministic(was)
            n_active = (deg + 1) ** 2
            assert float(rest.grad[:, n_active - 1:].abs().max()) == 0.0 if n_active < 16 else True
            # tuple form
            img2, _ = rast(means3D=m, means2D=torch.zeros_like(m), opacities=o, shs=(dc, rest), scales=s, rotations=q)
            assert torch.equal(img2, img0)
    finally:
        hip.FUSED_INRIA = old


def test_fused_inria_device_side_list_length_and_guesses(hip):
    """The fused Inria call (gspl_rasterize_inria_fwd) launches emission, sort and compositing BEFORE the host has the list length:
    the sort reads it on the device, sized by a guess from the last frame.  Frames whose guess is far too low (the frame is
    redone), about right, and far too high must all equal the staged path (which waits for the length) bit for bit — image, radii
    and every gradient (the backward is atomics-ordered: compared to the spread the staged path shows against itself)."""
    if not hip.FUSED_INRIA:
        pytest.skip("GSPL_FUSED_INRIA=0")
    small = _e2e_scene(n=500, seed=41)
    big = _e2e_scene(n=9000, seed=42)

    def render(scene, fused):
        means, scales, quats, opac, shs, cam, wimg, bg = scene
        W, H = cam["width"], cam["height"]
        saved = hip.FUSED_INRIA
        hip.FUSED_INRIA = fused
        try:
            leaves = [t.requires_grad_(True) for t in _cuda(means, scales * (6.0 if scene is big else 1.0), quats, opac, shs)]
            m, sc, q, o, c = leaves
            img, radii = hip.GaussianRasterizer(_inria_settings(hip, cam, bg, W, H))(
                means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, shs=c, scales=sc, rotations=q)
            (img * wimg.to(_dev())).sum().backward()
            return img.detach(), radii, [t.grad for t in leaves]
        finally:
            hip.FUSED_INRIA = saved

    ref = {id(sc): render(sc, False) for sc in (small, big)}
    hip.STATE.capacity.clear()
    misses0, cold0 = hip.SPECULATION["misses"], hip.SPECULATION["cold"]
    for sc in (small, big, big, small, small):          # room: none yet, far too low, right, far too high (twice)
        img, radii, grads = render(sc, True)
        r_img, r_radii, r_grads = ref[id(sc)]
        assert torch.equal(img, r_img) and torch.equal(radii, r_radii)
        for g, rg in zip(grads, r_grads):
            assert float((g - rg).abs().max()) <= 2e-5 * max(1.0, float(rg.abs().max()))
    assert hip.SPECULATION["cold"] - cold0 == 1 and hip.SPECULATION["misses"] - misses0 == 1
    # the same with the backward's sums in list order: fused and staged, whatever the guess, give the same bits
    was = hip.set_deterministic(True)
    try:
        ref = {id(sc): render(sc, False) for sc in (small, big)}
        for sc in (big, small, big):
            img, radii, grads = render(sc, True)
            r_img, r_radii, r_grads = ref[id(sc)]
            assert torch.equal(img, r_img) and all(torch.equal(g, rg) for g, rg in zip(grads, r_grads))
            again = render(sc, True)[2]
            assert all(torch.equal(g, rg) for g, rg in zip(grads, again)), "two runs of the deterministic mode differ"
    finally:
        hip.set_deterministic(was)


def test_a_fourfold_jump_in_list_length_misses_once_and_the_shuffled_stream_never_again(hip):
    """the room of the speculative emission is a decayed running maximum of the list entries per splat
    (ops._state.ListCapacity), not the previous frame's length.  A camera stream like the reference's — a fresh random permutation of
    views whose lists differ by a factor of four every epoch (internal/dataset.py:216-217) — costs ONE repeated frame, at the first
    view that outgrows everything seen so far, and none afterwards; the previous-frame policy of rounds 3-5 would have missed at every
    far-to-near transition.  Every frame equals the staged path (which waits for the length) bit for bit."""
    if not hip.FUSED_INRIA:
        pytest.skip("GSPL_FUSED_INRIA=0")
    means, scales, quats, opac, shs, cam, wimg, bg = _e2e_scene(n=9000, seed=44)
    W, H = cam["width"], cam["height"]
    params = _cuda(means, scales * 3.0, quats, opac, shs)

    def view(distance):
        c = O.synthetic_camera(W, H, 300.0, 295.0, distance=distance)
        return c

    def render(c, fused):
        saved, hip.FUSED_INRIA = hip.FUSED_INRIA, fused
        try:
            m, sc, q, o, sh = params
            img, radii = hip.GaussianRasterizer(_inria_settings(hip, c, bg, W, H))(
                means3D=m, means2D=torch.zeros_like(m), opacities=o, shs=sh, scales=sc, rotations=q)
            return img, radii
        finally:
            hip.FUSED_INRIA = saved

    key = (torch.device(_dev()).index, (W + 15) // 16, (H + 15) // 16)
    far, near, mid = view(9.0), view(2.6), view(4.5)
    lengths = {}
    for name, c in (("far", far), ("near", near), ("mid", mid)):
        render(c, True)
        lengths[name] = hip._LAST_ISECTS[key]
    assert lengths["near"] >= 4 * lengths["far"], lengths          # the jump the test is about
    ref = {id(c): render(c, False) for c in (far, near, mid)}
    hip.STATE.capacity.clear()
    frames0, misses0, cold0 = (hip.SPECULATION[k] for k in ("frames", "misses", "cold"))
    g = torch.Generator().manual_seed(0)
    stream = [far, far, near]                                      # cold, fits, the 4 x jump: one miss
    for _ in range(6):                                             # six "epochs" in shuffled order: no miss any more
        stream += [(far, near, mid)[i] for i in torch.randperm(3, generator=g).tolist()]
    for i, c in enumerate(stream):
        img, radii = render(c, True)
        assert torch.equal(img, ref[id(c)][0]) and torch.equal(radii, ref[id(c)][1]), i
        if i == 2:
            assert hip.SPECULATION["misses"] - misses0 == 1
    assert hip.SPECULATION["frames"] - frames0 == len(stream) and hip.SPECULATION["cold"] - cold0 == 1
    assert hip.SPECULATION["misses"] - misses0 == 1, "a view the running maximum already covers must not miss"


@pytest.mark.parametrize("segmented", [False, "always"])
def test_a_second_backward_through_a_retained_graph(hip, segmented):
    """VERDICT r5 #8b: the Inria op this replaces keeps its buffers with the graph and supports `retain_graph=True` — a regulariser
    that back-propagates twice through one frame.  The frame's device buffers are saved tensors of the node: a second backward
    through a retained graph gives the same gradients (deterministic mode: bit for bit), also with the segmented walk (whose work
    counter the forward cleared once), and a backward through a RELEASED graph gets autograd's own error."""
    if not hip.FUSED_INRIA:
        pytest.skip("GSPL_FUSED_INRIA=0")
    means, scales, quats, opac, shs, cam, wimg, bg = _e2e_scene(n=6000, seed=45)
    W, H = cam["width"], cam["height"]
    saved = hip.SEGMENTED_BACKWARD
    hip.SEGMENTED_BACKWARD = segmented
    was = hip.set_deterministic(not segmented)      # (the deterministic mode keeps the plain walk: bit-exact there, atomics' spread with segments)
    try:
        leaves = [t.requires_grad_(True) for t in _cuda(means, scales * 6.0, quats, opac, shs)]
        m, sc, q, o, c = leaves
        screen = torch.zeros_like(m, requires_grad=True)
        img, radii = hip.GaussianRasterizer(_inria_settings(hip, cam, bg, W, H))(means3D=m, means2D=screen, opacities=o, shs=c, scales=sc, rotations=q)
        loss = (img * wimg.to(_dev())).sum()
        g1 = torch.autograd.grad(loss, leaves + [screen], retain_graph=True)
        g2 = torch.autograd.grad(loss, leaves + [screen], retain_graph=True)
        for a, b in zip(g1, g2):
            if segmented:
                assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))
            else:
                assert torch.equal(a, b)
        assert float(g1[0].abs().sum()) > 0
        g3 = torch.autograd.grad(loss, leaves + [screen])              # releases the graph ...
        assert float((g3[0] - g1[0]).abs().max()) <= 2e-5 * max(1.0, float(g1[0].abs().max()))
        with pytest.raises(RuntimeError):                              # ... and then autograd refuses, as for any op
            torch.autograd.grad(loss, leaves + [screen])
    finally:
        hip.set_deterministic(was)
        hip.SEGMENTED_BACKWARD = saved


def test_degenerate_inputs(hip):
    """Everything behind the camera, a single huge splat covering the whole image, image smaller than a tile."""
    d = _dev()
    cam = O.synthetic_camera(40, 24, 30.0)
    bg = torch.tensor([0.2, 0.4, 0.6])
    # all splats behind the camera -> pure background, zero gradients, radii 0
    means = torch.tensor([[0.0, 0.0, -10.0], [0.5, 0.1, -8.0]])
    scales = torch.full((2, 3), 0.1)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 2)
    opac = torch.full((2, 1), 0.5)
    shs = torch.zeros(2, 16, 3)
    m = means.to(d).requires_grad_(True)
    r, radii = hip.GaussianRasterizer(_inria_settings(hip, cam, bg, 40, 24))(
        means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=opac.to(d), shs=shs.to(d), scales=scales.to(d),
        rotations=quats.to(d))
    assert int(radii.sum()) == 0 and torch.allclose(r, bg.to(d)[:, None, None].expand_as(r))
    r.sum().backward()
    assert float(m.grad.abs().sum()) == 0
    # one huge opaque splat in front: every pixel saturates to its colour (dc = (c - 0.5)/C0)
    means = torch.tensor([[0.0, 0.0, 0.0]])
    scales = torch.full((1, 3), 50.0)
    shs = torch.zeros(1, 16, 3)
    shs[0, 0] = (torch.tensor([0.9, 0.1, 0.5]) - 0.5) / 0.28209479177387814
    r, radii = hip.GaussianRasterizer(_inria_settings(hip, cam, bg, 40, 24))(
        means3D=means.to(d), means2D=torch.zeros(1, 3, device=d), opacities=torch.ones(1, 1, device=d), shs=shs.to(d),
        scales=scales.to(d), rotations=quats[:1].to(d))
    ref = O.render_inria(means.double(), scales.double(), quats[:1].double(), torch.ones(1, 1).double(), shs.double(), 3,
                         cam["world_to_camera"].double(), cam["full_projection"].double(), cam["camera_center"].double(),
                         cam["tanfovx"], cam["tanfovy"], 40, 24, bg.double())
    assert int(radii[0]) > 0 and float((r.cpu() - ref["render"].float()).abs().max()) <= 1e-5
    # image smaller than one tile, gsplat API
    xys, depths, radii, conics, comp, tiles, _ = hip.project_gaussians(
        torch.tensor([[0.0, 0.0, 0.0]], device=d), torch.full((1, 3), 0.3, device=d), 1.0, quats[:1].to(d),
        torch.eye(4, device=d)[:3] + torch.tensor([[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 4.0]], device=d), 20.0, 20.0, 5.0, 3.5, 7, 10, 16)
    img = hip.rasterize_gaussians(xys, depths, radii, conics, tiles, torch.ones(1, 3, device=d), torch.ones(1, 1, device=d), 7, 10, 16,
                                  torch.zeros(3, device=d))
    assert img.shape == (7, 10, 3) and float(img.max()) > 0.5
