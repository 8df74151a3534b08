"""The unchanged training loop on the other side of the renderer boundary, without Lightning
(reference: internal/gaussian_splatting.py:329-397 `training_step`; internal/density_controllers/vanilla_density_controller.py).

  * CPU, reference tree importable (lightning stubbed): the REAL `VanillaGaussianModel` + `VanillaDensityControllerImpl` of the
    reference drive `HipVanillaRenderer` (its native op replaced in this test by the oracle pipeline) through densify / prune /
    opacity reset; the in-repo restatement of that consumer code (oracle/training_oracle.py) must make the same decisions.
  * GPU: the restated consumer code drives `HipVanillaRenderer` and `HipGSplatRenderer(absgrad)` on the HIP ops for 300 steps
    across N changes (speculative emission capacity, per-camera caches, workspace sizes): the loss must fall, N must change, and
    the final PSNR must match the same loop on the oracle renderer.
"""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as O
from oracle import training_oracle as T

REF_ROOT = os.environ.get("GSPL_REFERENCE_ROOT", "/root/reference")
W_IMG, H_IMG = 160, 112


def _cameras(n=6):
    cams = []
    for i in range(n):
        ang = 2 * math.pi * i / n
        cam = O.synthetic_camera(W_IMG, H_IMG, 150.0, 150.0)
        # orbit: rotate about y, look at the origin from distance 4 (row-vector / transposed storage)
        c, s = math.cos(ang), math.sin(ang)
        R = torch.tensor([[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]])          # world -> camera rotation (standard)
        w2c = torch.eye(4)
        w2c[:3, :3] = R.T
        w2c[3, :3] = torch.tensor([0.0, 0.0, 4.0])
        cam["full_projection"] = w2c @ (torch.linalg.inv(cam["world_to_camera"]) @ cam["full_projection"])
        cam["world_to_camera"] = w2c
        cam["camera_center"] = torch.linalg.inv(w2c)[3, :3]
        cam["idx"] = i
        cams.append(cam)
    return cams


def _gt_and_init(seed=9):
    g = torch.Generator().manual_seed(seed)
    # ground truth: 600 fat, fairly opaque, coloured splats
    n = 600
    gt = dict(means=(torch.rand(n, 3, generator=g) * 2 - 1) * 0.9, scales=torch.exp(torch.randn(n, 3, generator=g) * 0.3 - 2.3),
              quats=torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1), opac=torch.rand(n, 1, generator=g) * 0.5 + 0.45,
              shs=torch.cat([torch.randn(n, 1, 3, generator=g) * 0.8, torch.randn(n, 15, 3, generator=g) * 0.05], dim=1))
    m = 2500
    init = dict(means=(torch.rand(m, 3, generator=g) * 2 - 1) * 1.0, scales=torch.full((m, 3), 0.05),
                quats=torch.nn.functional.normalize(torch.randn(m, 4, generator=g), dim=-1), opac=torch.full((m, 1), 0.1),
                shs=torch.cat([torch.randn(m, 1, 3, generator=g) * 0.3, torch.zeros(m, 15, 3)], dim=1))
    return gt, init


def _targets(gt, cams, bg):
    out = []
    with torch.no_grad():
        for cam in cams:
            r = O.render_inria(gt["means"], gt["scales"], gt["quats"], gt["opac"], gt["shs"], 3, cam["world_to_camera"], cam["full_projection"],
                               cam["camera_center"], cam["tanfovx"], cam["tanfovy"], W_IMG, H_IMG, bg)
            out.append(r["render"].float())
    return out


class OracleVanillaRenderer:
    """`HipVanillaRenderer`'s contract on the oracle pipeline (CPU): render, viewspace_points whose .grad[:, :2] is the
    NDC-scaled screen-space gradient, visibility_filter, radii."""

    def __call__(self, camera, pc, bg_color, **kw):
        means = pc.get_xyz
        vp = torch.zeros_like(means).requires_grad_(True)
        r = O.render_inria(means, pc.get_scaling, pc.get_rotation, pc.get_opacity, pc.get_features, int(pc.active_sh_degree),
                           camera.world_to_camera, camera.full_projection, camera.camera_center,
                           math.tan(float(camera.fov_x) * 0.5), math.tan(float(camera.fov_y) * 0.5), W_IMG, H_IMG, bg_color)
        scale = torch.tensor([0.5 * W_IMG, 0.5 * H_IMG], dtype=means.dtype)

        def deliver(g):
            vp.grad = torch.cat([g * scale, torch.zeros_like(g[:, :1])], dim=1)
        if r["xy"].requires_grad:
            r["xy"].register_hook(deliver)
        return {"render": r["render"], "viewspace_points": vp, "visibility_filter": r["radii"] > 0, "radii": r["radii"]}


CONTROLLER_KW = dict(percent_dense=0.01, densification_interval=40, opacity_reset_interval=150, opacity_reset_value=0.01,
                     densify_from_iter=40, densify_until_iter=260, densify_grad_threshold=0.00012, cull_opacity_threshold=0.005)
EXTENT = 4.4


def _restated_setup(init, dev, absgrad=False, optimizer=None, **override):
    model = T.TrainableGaussians(init["means"].to(dev), init["scales"].to(dev), init["quats"].to(dev), init["opac"].to(dev), init["shs"].to(dev))
    opts = model.make_optimizers(EXTENT) if optimizer is None else model.make_optimizers(EXTENT, **optimizer)
    ctrl = T.DensityControllerOracle(model.n_gaussians, dev, EXTENT, absgrad=absgrad, **{**CONTROLLER_KW, **override})
    return model, opts, ctrl


def _seeded_steps(fn):
    """Densification draws `torch.normal` samples from the global generator of the tensor's device: re-seed it (CPU and
    GPU) every step, and draw on the CPU, so that two loops make the same draws whatever device they run on."""
    def on_step(step, outputs):
        torch.manual_seed(1000 + step)
        if fn is not None:
            fn(step, outputs)
    return on_step


class _cpu_normal:
    """While active, `torch.normal(mean=, std=)` draws on the CPU and moves the result to `std`'s device."""

    def __enter__(self):
        self.orig = torch.normal
        orig = self.orig

        def normal(mean=None, std=None, **kw):
            return orig(mean=mean.cpu(), std=std.cpu(), **kw).to(std.device)
        torch.normal = normal

    def __exit__(self, *exc):
        torch.normal = self.orig


def _run_restated(render, dev, steps, absgrad=False, camera_cls=None, optimizer=None):
    from fakes import FakeCamera
    cams = _cameras()
    gt, init = _gt_and_init()
    bg = torch.zeros(3)
    targets = [t.to(dev) for t in _targets(gt, cams, bg)]
    model, opts, ctrl = _restated_setup(init, dev, absgrad, optimizer=optimizer)
    cam_objs = [FakeCamera(c, dev) for c in cams]
    torch.manual_seed(999)
    with _cpu_normal():
        hist = T.train(model, ctrl, opts, render, cam_objs, targets, steps, bg.to(dev), sh_degree_up_interval=60, on_step=_seeded_steps(None))
    with torch.no_grad():
        final = [render(c, model, bg.to(dev))["render"] for c in cam_objs]
    ps = float(np.mean([T.psnr(f.cpu(), t.cpu()) for f, t in zip(final, targets)]))
    return hist, ps, model


def _check_history(hist, name):
    losses = [h[0] for h in hist]
    ns = [h[1] for h in hist]
    first, last = float(np.mean(losses[:12])), float(np.mean(losses[-12:]))
    changes = sum(1 for a, b in zip(ns, ns[1:]) if a != b)
    print(f"{name}: loss {first:.4f} -> {last:.4f}, N {ns[0]} -> {ns[-1]} ({changes} changes, max {max(ns)})")
    assert last < 0.75 * first, (name, first, last)
    assert changes >= 2 and max(ns) > ns[0], (name, ns[0], ns[-1], changes)
    assert all(math.isfinite(l) for l in losses)


def test_restated_loop_trains_on_the_oracle_renderer_cpu():
    hist, ps, _ = _run_restated(OracleVanillaRenderer(), torch.device("cpu"), 140)
    _check_history(hist, "oracle renderer (CPU)")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ROOT, "internal", "density_controllers", "vanilla_density_controller.py")),
                    reason="reference tree not present")
def test_reference_consumer_code_drives_the_plugin_and_matches_the_restatement_cpu():
    """Real reference model + density controller (lightning stubbed) drive HipVanillaRenderer whose native op is replaced, in
    this test, by the oracle pipeline; the restated consumer code makes the same decisions on the same inputs."""
    if "lightning" not in sys.modules:
        L = types.ModuleType("lightning")
        L.LightningModule = type("LightningModule", (), {})
        sys.modules["lightning"] = L
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from internal.density_controllers.vanilla_density_controller import VanillaDensityController
    from internal.models.vanilla_gaussian import VanillaGaussian
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    from gspl_amd.renderers import HipVanillaRenderer
    from fakes import FakeCamera

    oracle_render = OracleVanillaRenderer()
    seen_raw = []

    class OracleRasterizer:          # stands in for ops.GaussianRasterizer (the one native call of the plugin)
        def __init__(self, raster_settings):
            self.s = raster_settings

        def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                     shs_rest=None, raw_parameters=False):
            s = self.s
            seen_raw.append(bool(raw_parameters))
            if raw_parameters:            # the real VanillaGaussianModel qualifies (renderer.model_raw_parameters): the plugin hands over
                # its raw parameters and the rasterizer owns the three activations
                opacities, scales, rotations = torch.sigmoid(opacities), torch.exp(scales), torch.nn.functional.normalize(rotations)
            if shs_rest is not None:      # the plugin hands over the model's two SH parameters as they are stored
                shs = torch.cat((shs, shs_rest), dim=1)
            r = O.render_inria(means3D, scales, rotations, opacities, shs, s.sh_degree, s.viewmatrix, s.projmatrix, s.campos,
                               s.tanfovx, s.tanfovy, s.image_width, s.image_height, s.bg)
            sc = torch.tensor([0.5 * s.image_width, 0.5 * s.image_height])
            if r["xy"].requires_grad:
                r["xy"].register_hook(lambda g: setattr(means2D, "grad", torch.cat([g * sc, torch.zeros_like(g[:, :1])], dim=1)))
            return r["render"], r["radii"]

    cams = _cameras()
    gt, init = _gt_and_init()
    bg = torch.zeros(3)
    targets = _targets(gt, cams, bg)
    cam_objs = [FakeCamera(c, "cpu") for c in cams]
    steps = 130

    # ---- the reference's own consumer code
    model = VanillaGaussian(sh_degree=3).instantiate()
    model.setup_from_tensors({"means": init["means"], "shs_dc": init["shs"][:, :1], "shs_rest": init["shs"][:, 1:],
                              "opacities": T.inverse_sigmoid(init["opac"]), "scales": torch.log(init["scales"]), "rotations": init["quats"]},
                             active_sh_degree=0)
    g = model.gaussians
    opts = [torch.optim.Adam([{"params": [g["means"]], "name": "means"}], lr=0.00016 * EXTENT, eps=1e-15),
            torch.optim.Adam([{"params": [g["shs_dc"]], "lr": 0.0025, "name": "shs_dc"}, {"params": [g["shs_rest"]], "lr": 0.0025 / 20.0, "name": "shs_rest"},
                              {"params": [g["scales"]], "lr": 0.005, "name": "scales"}, {"params": [g["rotations"]], "lr": 0.001, "name": "rotations"},
                              {"params": [g["opacities"]], "lr": 0.05, "name": "opacities"}], lr=0.0, eps=1e-15)]
    ctrl = VanillaDensityController(**{**CONTROLLER_KW, "opacity_reset_interval": 60}).instantiate()
    dp = types.SimpleNamespace(camera_extent=EXTENT)
    module = types.SimpleNamespace(trainer=types.SimpleNamespace(datamodule=types.SimpleNamespace(dataparser_outputs=dp, prune_extent=EXTENT)),
                                   gaussian_model=model, device=torch.device("cpu"), background_color=bg)
    ctrl.setup("fit", module)
    plugin = HipVanillaRenderer()
    saved = ops.GaussianRasterizer
    ops.GaussianRasterizer = OracleRasterizer
    try:
        torch.manual_seed(999)
        hist_ref = T.train(model, ctrl, opts, lambda c, m, b: plugin(c, m, b), cam_objs, targets, steps, bg, sh_degree_up_interval=60,
                           on_step=_seeded_steps(None), controller_is_reference=True, pl_module=module)
    finally:
        ops.GaussianRasterizer = saved
    assert seen_raw and all(seen_raw)          # the reference's own model was recognised in every step

    # ---- the restatement on the same inputs
    model2, opts2, ctrl2 = _restated_setup(init, torch.device("cpu"), opacity_reset_interval=60)
    torch.manual_seed(999)
    hist2 = T.train(model2, ctrl2, opts2, oracle_render, cam_objs, targets, steps, bg, sh_degree_up_interval=60, on_step=_seeded_steps(None))
    assert [h[1] for h in hist_ref] == [h[1] for h in hist2], "N per step differs between the reference's controller and the restatement"
    ns = [h[1] for h in hist_ref]
    assert sum(1 for a, b in zip(ns, ns[1:]) if a != b) >= 2
    np.testing.assert_allclose([h[0] for h in hist_ref], [h[0] for h in hist2], rtol=1e-5, atol=1e-7)
    for name in model2.get_property_names():
        np.testing.assert_allclose(model.get_property(name).detach().numpy(), model2.get_property(name).detach().numpy(), rtol=1e-4, atol=1e-6,
                                   err_msg=name)
    np.testing.assert_allclose(ctrl.max_radii2D.numpy(), ctrl2.max_radii2D.numpy())
    np.testing.assert_allclose(ctrl.xyz_gradient_accum.numpy(), ctrl2.xyz_gradient_accum.numpy(), rtol=1e-4, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["vanilla", "vanilla-fused-adam-deferred", "gsplat-absgrad"])
def test_training_loop_survives_density_changes_on_the_hip_renderers(which):
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers import HipGSplatRenderer, HipVanillaRenderer
    dev = torch.device("cuda:0")
    steps = 290          # one opacity reset (step 150), densification until step 260
    if which.startswith("vanilla"):
        plugin = HipVanillaRenderer()
        optimizer = None
        if which == "vanilla-fused-adam-deferred":
            # the package's fused Adam with the shs_rest update on the colour stream, through densify / prune / opacity reset: the
            # controller replaces the parameters and swaps the optimizer state between a step and the next render
            from gspl_amd.optimizers import FusedAdam
            optimizer = dict(cls=FusedAdam, deferred=("shs_rest",))
        hist, ps, model = _run_restated(lambda c, m, b: plugin(c, m, b), dev, steps, optimizer=optimizer)
        hist_o, ps_o, _ = _run_restated(OracleVanillaRenderer(), torch.device("cpu"), steps)
        _check_history(hist, "HipVanillaRenderer")
        print(f"PSNR HIP {ps:.3f} dB, oracle renderer {ps_o:.3f} dB; final N {hist[-1][1]} vs {hist_o[-1][1]}")
        # Two 290-step trajectories with Adam (eps 1e-15: a gradient of 1e-12 with a flipped sign is a full-size step) and
        # thresholded densification decisions diverge from fp32-vs-fp64 rounding alone; measured 0.10 dB at 38.5 dB
        assert abs(ps - ps_o) <= 0.25, (ps, ps_o)
        assert abs(hist[-1][1] - hist_o[-1][1]) <= 0.02 * hist_o[-1][1]
    else:
        plugin = HipGSplatRenderer(absgrad=True)
        hist, ps, model = _run_restated(lambda c, m, b: plugin(c, m, b), dev, steps, absgrad=True)
        _check_history(hist, "HipGSplatRenderer(absgrad)")
        assert ps > 14.0, ps
