"""The package-level shims of gspl_amd.compat: `diff_gaussian_rasterization` and the yzslab `gsplat` fork are registered under
the module paths and names the reference imports, so that the reference's OWN renderer classes run unedited.

CPU, reference tree importable (lightning / viser stubbed): the reference's `VanillaRenderer`, `GSPlatRenderer` and
`GSplatV1Renderer` are imported against the shims and executed with the HIP ops replaced — in this test — by the oracle stages:
what is checked is the wiring (module paths, function names, argument lists, return conventions), end to end, against the
oracle pipelines of the same conventions."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as O

REF_ROOT = os.environ.get("GSPL_REFERENCE_ROOT", "/root/reference")
needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF_ROOT, "internal", "renderers", "gsplat_v1_renderer.py")),
                                     reason="reference tree not present")


def _stubs():
    if "lightning" not in sys.modules:
        Lm = types.ModuleType("lightning")
        Lm.LightningModule = type("LightningModule", (), {})
        sys.modules["lightning"] = Lm
    if "viser" not in sys.modules:
        V = types.ModuleType("viser")
        V.ViserServer = type("ViserServer", (), {})
        sys.modules["viser"] = V
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def test_shim_modules_expose_the_names_the_reference_imports():
    import gspl_amd  # noqa: F401
    from gspl_amd import compat, ops
    compat.install()
    import diff_gaussian_rasterization as dgr
    if "gspl_amd" in (dgr.__doc__ or ""):
        assert dgr.GaussianRasterizer is ops.GaussianRasterizer and dgr.GaussianRasterizationSettings is ops.GaussianRasterizationSettings
    import gsplat
    if "gspl_amd" not in (gsplat.__doc__ or ""):
        pytest.skip("a real gsplat package is installed")
    from gsplat.sh import spherical_harmonics  # noqa: F401
    from gsplat.rasterize import rasterize_gaussians  # noqa: F401
    from gsplat.project_gaussians import project_gaussians  # noqa: F401
    from gsplat.v0_interfaces import project_gaussians as p2, rasterize_to_pixels  # noqa: F401
    from gsplat.sh_decomposed import spherical_harmonics_decomposed  # noqa: F401
    from gsplat.cuda._wrapper import fully_fused_projection, isect_offset_encode, isect_tiles  # noqa: F401
    from gsplat.cuda.isect_tiles_tile_based_culling import isect_tiles_tile_based_culling, isect_offset_encode_tile_based_culling  # noqa: F401
    from gsplat.hit_pixel_count import hit_pixel_count  # noqa: F401
    from gsplat.rasterize_to_weights import rasterize_to_weights  # noqa: F401
    from gsplat.optimizers import SelectiveAdam
    from gspl_amd import optimizers
    assert SelectiveAdam is optimizers.SelectiveAdam
    with pytest.raises(ImportError):
        from gsplat.relocation import compute_relocation  # noqa: F401  (not built: fails as it would without the package)


def _scene(n=1500, W=176, H=128, seed=5):
    means, scales, quats, opac, shs = [t.double() for t in O.synthetic_scene(n, seed=seed)]
    scales = scales * 5
    cam = O.synthetic_camera(W, H, 170.0, 168.0)
    bg = torch.tensor([0.2, 0.4, 0.1], dtype=torch.float64)
    return (means, scales, quats, opac, shs), cam, bg


class _Cam:
    """Reference-style camera (tensors, as internal/cameras/cameras.py keeps them) in fp64 on the CPU."""

    def __init__(self, cam):
        import math
        t = lambda v, dt=torch.float64: torch.tensor(v, dtype=dt)
        self.world_to_camera, self.full_projection = cam["world_to_camera"].double(), cam["full_projection"].double()
        self.camera_center = cam["camera_center"].double()
        self.fx, self.fy, self.cx, self.cy = t(cam["fx"]), t(cam["fy"]), t(cam["cx"]), t(cam["cy"])
        self.width, self.height = t(cam["width"], torch.int32), t(cam["height"], torch.int32)
        self.fov_x, self.fov_y = t(2 * math.atan(cam["tanfovx"])), t(2 * math.atan(cam["tanfovy"]))
        self.R = self.world_to_camera.T[:3, :3]
        self.camera_type = t(0, torch.int8)


def _oracle_ops(monkeypatch):
    """gspl_amd.ops entry points -> oracle stages (this test only; the shims bind at call time)."""
    from gspl_amd import ops

    def project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width, block_width,
                          clip_thresh=0.01, filter_2d_kernel_size=0.3):
        w2c = torch.eye(4, dtype=means3d.dtype)
        w2c[:3, :] = viewmat[:3, :]
        xys, depths, radii, conics, comp, tiles, cov3d, _, _, _ = O.project_gaussians(
            means3d, scales, glob_scale, quats, w2c.T, float(fx), float(fy), float(cx), float(cy), int(img_height), int(img_width),
            eps2d=filter_2d_kernel_size)
        return xys, depths, radii, conics, comp, tiles, cov3d

    def spherical_harmonics(degree, dirs, coeffs, masks=None):
        return O.eval_sh(degree, coeffs, dirs)

    def spherical_harmonics_decomposed(degree, dirs, dc, coeffs, masks=None):
        return O.eval_sh(degree, torch.cat([dc, coeffs], dim=1), dirs)

    def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width, block_width,
                            background=None, return_alpha=False, **kw):
        _, _, flat, offs = O.isect_tiles(O.MODE_GSPLAT, xys.detach(), radii, depths.detach(), img_width, img_height)
        out, alpha = O.composite_c(O.MODE_GSPLAT, xys, conics, colors, opacity.reshape(-1), background, img_width, img_height, offs, flat)
        return (out, alpha) if return_alpha else out

    def fully_fused_projection(means, covars, quats, scales, viewmats, Ks, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10,
                               radius_clip=0.0, packed=False, sparse_grad=False, calc_compensations=False, camera_model="pinhole", **kw):
        K = Ks[0]
        xys, depths, radii, conics, comp, _, _, _, _, _ = O.project_gaussians(
            means, scales, 1.0, quats, viewmats[0].T, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), int(height), int(width),
            eps2d=eps2d, camera_model=camera_model)
        return radii[None], xys[None], depths[None], conics[None], (comp[None] if calc_compensations else None)

    def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, packed=False, n_cameras=None, camera_ids=None,
                    gaussian_ids=None, **kw):
        tiles, ids, flat, offs = O.isect_tiles(O.MODE_GSPLAT, means2d.reshape(-1, 2).detach(), radii.reshape(-1), depths.reshape(-1).detach(),
                                               tile_width * tile_size, tile_height * tile_size)
        return torch.as_tensor(np.asarray(tiles))[None], torch.as_tensor(np.asarray(ids)), torch.as_tensor(np.asarray(flat))

    def isect_offset_encode(isect_ids, n_cameras, tile_width, tile_height):
        tile_of = (isect_ids >> 32).to(torch.int64)
        offs = torch.searchsorted(tile_of, torch.arange(tile_width * tile_height, dtype=torch.int64)).to(torch.int32)
        return offs.reshape(1, tile_height, tile_width)

    def rasterize_to_pixels(means2d, conics, colors, opacities, image_width, image_height, tile_size, isect_offsets, flatten_ids,
                            backgrounds=None, absgrad=False, track_hits=False, **kw):
        if track_hits:
            means2d.has_hit_any_pixels = torch.ones(means2d.shape[-2], dtype=torch.bool)      # (not compared here)
        out, alpha = O.composite_c(O.MODE_GSPLAT, means2d.reshape(-1, 2), conics.reshape(-1, 3), colors.reshape(-1, colors.shape[-1]),
                                   opacities.reshape(-1), backgrounds.reshape(-1), image_width, image_height,
                                   isect_offsets.reshape(-1).numpy(), flatten_ids.numpy())
        return out[None], alpha[None, ..., None]

    for name, fn in dict(project_gaussians=project_gaussians, spherical_harmonics=spherical_harmonics,
                         spherical_harmonics_decomposed=spherical_harmonics_decomposed, rasterize_gaussians=rasterize_gaussians,
                         fully_fused_projection=fully_fused_projection, isect_tiles=isect_tiles, isect_offset_encode=isect_offset_encode,
                         rasterize_to_pixels=rasterize_to_pixels).items():
        monkeypatch.setattr(ops, name, fn)


@needs_reference
def test_reference_gsplat_renderers_run_unedited_on_the_shimmed_packages(monkeypatch):
    import gspl_amd  # noqa: F401
    from gspl_amd import compat
    import gsplat as _g
    compat.install()
    if "gspl_amd" not in (_g.__doc__ or ""):
        pytest.skip("a real gsplat package is installed")
    _stubs()
    from fakes import FakeGaussianModel
    from internal.renderers.gsplat_renderer import GSPlatRenderer              # the reference's own classes, unedited
    from internal.renderers.gsplat_v1_renderer import GSplatV1Renderer
    _oracle_ops(monkeypatch)
    params, cam, bg = _scene()
    def oracle(p):
        return O.render_gsplat(*p, 3, cam["world_to_camera"].double(), cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["width"], cam["height"],
                               bg, cam["camera_center"].double())

    means, scales, quats, opac, shs = params
    ref_as_given = oracle(params)
    ref_normalised = oracle((means, scales, quats / quats.norm(dim=-1, keepdim=True), opac, shs))      # gsplat_renderer.py:68
    for make, ref in ((lambda: GSPlatRenderer(), ref_normalised), (lambda: GSplatV1Renderer().instantiate(), ref_as_given),
                      (lambda: GSplatV1Renderer(separate_sh=True).instantiate(), ref_as_given)):
        model = FakeGaussianModel(*[p.clone() for p in params])
        renderer = make()
        out = renderer(_Cam(cam), model, bg)
        assert out["render"].shape == (3, cam["height"], cam["width"])
        assert float((out["render"].detach() - ref["render"]).abs().max()) <= 1e-9, type(renderer).__name__
        assert torch.equal(out["visibility_filter"].reshape(-1), ref["mask"]) and torch.equal(out["radii"].reshape(-1), ref["radii"])
        out["render"].sum().backward()                                          # gradients flow through the shimmed calls
        assert model.means.grad is not None and float(model.means.grad.abs().sum()) > 0


@needs_reference
def test_reference_vanilla_renderer_runs_unedited_on_the_shimmed_package(monkeypatch):
    import gspl_amd  # noqa: F401
    from gspl_amd import compat
    compat.install()
    import diff_gaussian_rasterization as dgr
    if "gspl_amd" not in (dgr.__doc__ or ""):
        pytest.skip("a real diff_gaussian_rasterization package is installed")
    _stubs()
    from fakes import FakeGaussianModel
    import internal.renderers.vanilla_renderer as vr                             # imports GaussianRasterizer from the shim
    assert vr.GaussianRasterizer is dgr.GaussianRasterizer and vr.GaussianRasterizationSettings is dgr.GaussianRasterizationSettings

    class OracleRasterizer:          # stands in for the HIP rasterizer behind the shim (this test runs without a GPU)
        def __init__(self, raster_settings):
            self.s = raster_settings

        def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
            s = self.s
            r = O.render_inria(means3D, scales, rotations, opacities, shs, s.sh_degree, s.viewmatrix, s.projmatrix, s.campos,
                               s.tanfovx, s.tanfovy, s.image_width, s.image_height, s.bg)
            return r["render"], r["radii"]

    monkeypatch.setattr(vr, "GaussianRasterizer", OracleRasterizer)
    params, cam, bg = _scene(seed=6)
    model = FakeGaussianModel(*[p.clone() for p in params])
    out = vr.VanillaRenderer()(_Cam(cam), model, bg)
    ref = O.render_inria(*params, 3, cam["world_to_camera"].double(), cam["full_projection"].double(), cam["camera_center"].double(),
                         cam["tanfovx"], cam["tanfovy"], cam["width"], cam["height"], bg)
    assert float((out["render"] - ref["render"]).abs().max()) <= 1e-9
    assert torch.equal(out["radii"], ref["radii"]) and torch.equal(out["visibility_filter"], ref["radii"] > 0)


@pytest.mark.gpu
def test_shimmed_gsplat_module_paths_drive_the_hip_ops():
    """The fork's pipeline written the way gsplat_v1_renderer.py writes it (imports from the `gsplat` package paths), on the GPU:
    same image as HipGSplatV1Renderer bit for bit, and the rasterizer leaves `has_hit_any_pixels` behind unconditionally."""
    import math
    import gspl_amd  # noqa: F401
    from gspl_amd import compat
    compat.install()
    import gsplat
    if "gspl_amd" not in (gsplat.__doc__ or ""):
        pytest.skip("a real gsplat package is installed")
    from gsplat.cuda._wrapper import fully_fused_projection, isect_offset_encode, isect_tiles, spherical_harmonics
    from gsplat.v0_interfaces import rasterize_to_pixels
    from fakes import FakeCamera, FakeGaussianModel
    from gspl_amd.renderers import HipGSplatV1Renderer
    dev = torch.device("cuda:0")
    params, cam, bg = _scene()
    means, scales, quats, opac, shs = [p.float().to(dev) for p in params]
    W, H = cam["width"], cam["height"]
    viewmats = cam["world_to_camera"].float().T[None].to(dev)
    Ks = torch.tensor([[[cam["fx"], 0., cam["cx"]], [0., cam["fy"], cam["cy"]], [0., 0., 1.]]], dtype=torch.float32, device=dev)
    radii, means2d, depths, conics, comp = fully_fused_projection(means, None, quats, scales, viewmats=viewmats, Ks=Ks, width=W, height=H,
                                                                  eps2d=0.3, calc_compensations=True, packed=False)      # anti_aliased: the renderer's default
    tw, th = math.ceil(W / 16.), math.ceil(H / 16.)
    _, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, 16, tw, th, packed=False, n_cameras=1)
    offsets = isect_offset_encode(isect_ids, 1, tw, th)
    dirs = means - cam["camera_center"].float().to(dev)
    colors = torch.clamp_min(spherical_harmonics(3, dirs, shs, radii.squeeze(0) > 0) + 0.5, 0.)
    m2 = means2d.squeeze(0)
    img, alpha = rasterize_to_pixels(means2d=m2, conics=conics, colors=colors.unsqueeze(0), opacities=opac.reshape(1, -1) * comp,
                                     image_width=W, image_height=H, tile_size=16, isect_offsets=offsets, flatten_ids=flatten_ids,
                                     backgrounds=bg.float().to(dev).unsqueeze(0), absgrad=False)
    hit = m2.has_hit_any_pixels
    assert hit.dtype == torch.bool and hit.shape == (means.shape[0],) and 0 < int(hit.sum()) <= int((radii.reshape(-1) > 0).sum())
    out = HipGSplatV1Renderer().instantiate()(FakeCamera(cam, dev), FakeGaussianModel(means, scales, quats, opac, shs), bg.float().to(dev))
    assert torch.equal(img.squeeze(0).permute(2, 0, 1), out["render"])
    assert torch.equal(hit, out["acc_vis"])
