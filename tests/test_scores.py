"""Per-splat compositing statistics (include/gspl_hip.h §12: hit-pixel count / rasterize_to_weights) against the fp64
restatement in oracle/scores_oracle.py, and consistency with the renderer: the per-splat blending weights of a view add
up to the rendered alpha."""
import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as O
from oracle import scores_oracle as S


def _scene(n, W, H, seed, scale_mul=4.0):
    means, scales, quats, opac, shs = O.synthetic_scene(n, seed=seed)
    cam = O.synthetic_camera(W, H, 0.9 * W)
    res = O.project_gaussians(means, scales * scale_mul, 1.0, quats, cam["world_to_camera"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
    return res, opac.reshape(-1)


def test_oracle_blending_weights_sum_to_alpha():
    W, H = 64, 48
    res, opac = _scene(300, W, H, 3)
    xys, depths, radii, conics = res[0], res[1], res[2], res[3]
    _, _, flat, offs = O.isect_tiles(O.MODE_GSPLAT, xys, radii, depths, W, H)
    s = S.scores(O.MODE_GSPLAT, xys, conics, opac, W, H, offs, flat, pixel_weights=np.ones((H, W)))
    _, alpha, _, _ = O.composite_fwd(O.MODE_GSPLAT, xys, conics, torch.zeros(xys.shape[0], 1), opac, None, W, H, offs, flat)
    assert abs(s["visibility"].sum() - alpha.sum()) < 1e-9 * max(1.0, alpha.sum())
    assert np.allclose(s["weighted"], s["visibility"])
    assert (s["count"] > 0).sum() > 50 and np.all(s["alpha"] >= s["visibility"] - 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [O.MODE_GSPLAT, O.MODE_INRIA])
@pytest.mark.parametrize("wh", [(96, 80), (131, 77)])
def test_scores_vs_oracle(mode, wh):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    dev = "cuda:0"
    W, H = wh
    res, opac = _scene(1500, W, H, 7 + W)
    xys, depths, radii, conics = res[0], res[1], res[2], res[3]
    _, _, flat, offs = O.isect_tiles(mode, xys, radii, depths, W, H)
    g = torch.Generator().manual_seed(1)
    pw = torch.rand(H, W, generator=g)
    ref = S.scores(mode, xys, conics, opac, W, H, offs, flat, pixel_weights=pw.numpy())
    c = lambda t: t.to(dev)
    count, o_sum, a_sum, v_sum, w_sum, d_sum = ops.composite_scores(
        c(xys), c(conics), c(opac), W, H, 16, torch.from_numpy(offs).to(dev), torch.from_numpy(flat).to(dev), pixel_weights=c(pw), mode=mode,
        with_dist=True)
    # a pixel whose alpha sits within rounding of 1/255 or whose transmittance sits at the stop threshold may flip in fp32: allow
    # a handful of single-pixel differences in the counts, and compare the sums at fp32 accumulation accuracy
    dc = np.abs(count.cpu().numpy() - ref["count"])
    assert dc.max() <= 2 and (dc > 0).sum() <= 0.01 * len(dc) + 2
    for got, key in ((o_sum, "opacity"), (a_sum, "alpha"), (v_sum, "visibility"), (w_sum, "weighted"), (d_sum, "dist")):
        scale = max(1.0, float(np.abs(ref[key]).max()))
        err = np.abs(got.cpu().numpy() - ref[key]) / scale
        assert np.mean(err <= 2e-5) > 0.99 and err.max() < 2e-2, (key, err.max())


@pytest.mark.gpu
def test_reference_shaped_wrappers():
    """`hit_pixel_count` / `rasterize_to_weights` with the argument lists of the reference's call sites; the blending weights of
    all splats add up to the alpha image the renderer produces for the same inputs."""
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    dev = "cuda:0"
    W, H = 160, 112
    res, opac = _scene(4000, W, H, 21)
    xys, depths, radii, conics, tiles = (res[i].to(dev) for i in (0, 1, 2, 3, 5))
    opac = opac.to(dev)
    count, o_score, a_score, v_score = ops.hit_pixel_count(xys, depths, radii, conics, tiles, opac[:, None], H, W, 16)
    assert count.dtype == torch.int32 and count.shape == (4000,) and v_score.shape == (4000,)
    flat, offsets = ops.bin_gaussians(xys, depths, radii, H, W, 16, conics=conics, opacities=opac)
    img, alpha = ops.rasterize_to_pixels(xys, conics[None], torch.ones(1, 4000, 1, device=dev), opac[None], W, H, 16, offsets.reshape(1, (H + 15) // 16, (W + 15) // 16), flat)
    assert abs(float(v_score.sum()) - float(alpha.sum())) < 1e-3 * float(alpha.sum())
    assert torch.all(o_score >= 0) and torch.allclose(o_score, count.float() * opac, rtol=1e-4, atol=1e-4)
    aw, rc, bw, da = ops.rasterize_to_weights(xys[None], conics[None], opac[None], W, H, 16, offsets.reshape(1, (H + 15) // 16, (W + 15) // 16), flat,
                                              torch.full((1, H, W), 2.0, device=dev))
    assert aw.shape == rc.shape == bw.shape == da.shape == (1, 4000)
    assert torch.allclose(aw, 2.0 * bw, rtol=1e-5, atol=1e-6) and torch.equal(rc[0], count.float()) and torch.allclose(bw[0], v_score, rtol=1e-5, atol=1e-6)
