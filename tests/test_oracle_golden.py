"""The oracle restatement vs the reference's golden vectors (CPU only).

Fixtures come from tests/golden/make_golden.py, which imports the reference's own Python
(internal/utils/gaussian_projection.py, internal/utils/sh_utils.py), and from the literal
known-answer vector in the reference's tests/gaussian_projection_test.py:30-113.
"""
import os

import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as O


def _cam_args(z, p):
    fx, fy, cx, cy, W, H = z[p + "intr"]
    return dict(world_to_camera=torch.from_numpy(z[p + "w2c"]), fx=fx, fy=fy, cx=cx, cy=cy, img_height=int(H), img_width=int(W))


@pytest.mark.parametrize("cam", [0, 1])
def test_projection_matches_reference_python(golden_dir, cam):
    z = np.load(os.path.join(golden_dir, "ref_projection.npz"))
    p = f"cam{cam}_"
    means = torch.from_numpy(z["means"]).requires_grad_(True)
    scales = torch.from_numpy(z["scales"]).requires_grad_(True)
    quats = torch.from_numpy(z["quats"]).requires_grad_(True)
    xys, depths, radii, conics, comp, tiles, cov3d, mask, rmin, rmax = O.project_gaussians(
        means, scales, 1.0, quats, **_cam_args(z, p))
    assert np.array_equal(mask.numpy(), z[p + "mask"])
    assert np.array_equal(radii.numpy(), z[p + "radii"])
    assert np.array_equal(tiles.numpy(), z[p + "tiles"])
    assert np.array_equal(rmin.numpy(), z[p + "rect_min"])
    assert np.array_equal(rmax.numpy(), z[p + "rect_max"])
    np.testing.assert_allclose(xys.detach().numpy(), z[p + "xys"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(depths.detach().numpy(), z[p + "depths"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(conics.detach().numpy(), z[p + "conics"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(comp.detach().numpy(), z[p + "comp"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(cov3d.detach().numpy(), z[p + "cov3d"], rtol=1e-5, atol=1e-9)
    # gradients of the same fixed linear loss
    loss = (xys * torch.from_numpy(z[p + "w_xy"])).sum() + (depths * torch.from_numpy(z[p + "w_d"])).sum() \
        + (conics * torch.from_numpy(z[p + "w_c"])).sum() + (comp * torch.from_numpy(z[p + "w_k"])).sum()
    loss.backward()
    for got, name in ((means.grad, "g_means"), (scales.grad, "g_scales"), (quats.grad, "g_quats")):
        ref = z[p + name]
        scale = np.abs(ref).max()
        assert np.abs(got.numpy() - ref).max() <= 2e-4 * scale, name


def test_projection_fp64_agrees_with_fp32_reference(golden_dir):
    """The fp64 run of the restatement (what the GPU parity tests use) sits within fp32 noise of the
    reference's fp32 output, and its discrete outputs are identical."""
    z = np.load(os.path.join(golden_dir, "ref_projection.npz"))
    p = "cam1_"
    args = _cam_args(z, p)
    args["world_to_camera"] = args["world_to_camera"].double()
    out = O.project_gaussians(torch.from_numpy(z["means"]).double(), torch.from_numpy(z["scales"]).double(), 1.0,
                              torch.from_numpy(z["quats"]).double(), **args)
    same = out[2].numpy() == z[p + "radii"]
    assert same.mean() > 0.998          # ceil() may flip on a rounding boundary
    np.testing.assert_allclose(out[0].numpy()[same], z[p + "xys"][same], rtol=1e-5, atol=2e-3)


def test_known_answer_vector(golden_dir):
    """Reference's own literal KAT (tests/gaussian_projection_test.py:30-113)."""
    z = np.load(os.path.join(golden_dir, "ref_kat.npz"))
    fx, fy, cx, cy, W, H = z["intr"]
    xys, depths, radii, conics, comp, tiles, cov3d, mask, _, _ = O.project_gaussians(
        torch.from_numpy(z["means"]), torch.from_numpy(z["scales"]), 1.0, torch.from_numpy(z["quats"]),
        torch.from_numpy(z["w2c"]), fx, fy, cx, cy, int(H), int(W))
    assert radii.tolist() == z["exp_radii"].tolist() == [0, 4, 0, 16783]
    assert mask.tolist() == [False, True, False, True]
    m = mask.numpy()
    np.testing.assert_allclose(conics.numpy()[m], z["exp_conics_masked"], rtol=2e-5)
    np.testing.assert_allclose(comp.numpy()[m], z["exp_comp_masked"], rtol=1e-6)
    assert tiles.numpy()[m].tolist() == z["exp_tiles_masked"].tolist()
    up = cov3d.numpy()[m].reshape(-1, 9)[:, [0, 1, 2, 4, 5, 8]]
    np.testing.assert_allclose(up, z["exp_cov3d_upper_masked"], rtol=2e-5)
    # pixel-centre convention: literals were produced by the old NDC path (centres at integers),
    # the current reference returns +0.5; we follow the current reference.
    np.testing.assert_allclose(xys.numpy()[[1, 3]] - 0.5, z["exp_xys_rows13_ndc_convention"], rtol=2e-6, atol=2e-3)
    np.testing.assert_allclose(xys.numpy(), z["cur_xys"], rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_matches_reference_python(golden_dir, deg):
    z = np.load(os.path.join(golden_dir, "ref_sh.npz"))
    K = (deg + 1) ** 2
    c = torch.from_numpy(z["coeffs"][:, :K]).clone().requires_grad_(True)
    d = torch.from_numpy(z["dirs"]).clone().requires_grad_(True)
    rgb = O.eval_sh(deg, c, d, normalize=False)
    np.testing.assert_allclose(rgb.detach().numpy(), z[f"deg{deg}_rgb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rgb.detach().numpy(), z[f"deg{deg}_rgb_decomposed"], rtol=1e-5, atol=1e-6)
    (rgb * torch.from_numpy(z["w"])).sum().backward()
    np.testing.assert_allclose(c.grad.numpy(), z[f"deg{deg}_g_coeffs"], rtol=1e-5, atol=1e-6)
    if deg > 0:
        np.testing.assert_allclose(d.grad.numpy(), z[f"deg{deg}_g_dirs"], rtol=1e-4, atol=1e-5)


def test_sort_keys_match_reference_python(golden_dir):
    """build_gaussian_sort_key (gaussian_projection.py:159-208): emission order, Gaussian ids and
    depth bits are pinned by the reference's output; its tile bits are lost to an int32 << 32 wrap
    in that python helper, so the tile ids are checked against a direct loop over the rects."""
    z = np.load(os.path.join(golden_dir, "ref_sortkey.npz"))
    W, H = [int(v) for v in z["wh"]]
    counts, keys, gid = O.emit_keys(O.MODE_GSPLAT, z["xys"], z["radii"], z["depths"], W, H)
    assert np.array_equal(counts, z["tiles"])
    assert np.array_equal(gid, z["gids_unsorted"])
    ref_keys = z["keys_unsorted"].view(np.uint64)
    assert np.all((ref_keys >> np.uint64(32)) == 0)          # the reference helper's wrap
    assert np.array_equal(keys & np.uint64(0xFFFFFFFF), ref_keys & np.uint64(0xFFFFFFFF))
    # tile ids: row-major loop as written in the reference helper
    minx, miny, maxx, maxy = O.tile_rects(O.MODE_GSPLAT, z["xys"], z["radii"], W, H)
    gx = int(z["tile_bounds"][0])
    exp = []
    for g in range(z["xys"].shape[0]):
        for i in range(miny[g], maxy[g]):
            for j in range(minx[g], maxx[g]):
                exp.append(gx * i + j)
    assert np.array_equal((keys >> np.uint64(32)).astype(np.int64), np.asarray(exp, np.int64))

    tiles, ids, flat, offs = O.isect_tiles(O.MODE_GSPLAT, z["xys"], z["radii"], z["depths"], W, H)
    assert np.all(np.diff(ids.view(np.uint64).astype(np.float64)) >= 0)
    tile_of = (ids.view(np.uint64) >> np.uint64(32)).astype(np.int64)
    n_tiles = offs.shape[0]
    for t in range(n_tiles):
        s, e = offs[t], (offs[t + 1] if t + 1 < n_tiles else ids.shape[0])
        assert np.all(tile_of[s:e] == t)
        # within a tile: depth ascending, ties in Gaussian-id order
        d = z["depths"][flat[s:e]]
        assert np.all(np.diff(d) >= 0)
