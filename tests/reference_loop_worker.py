"""Worker of tests/test_reference_lightning_loop.py (its own process: the stand-in `lightning` package must not leak into the
other tests' imports of the reference tree).

Runs the reference's UNCHANGED LightningModule — `internal/gaussian_splatting.py`: `setup("fit")` (its own `setup_from_pcd`),
`configure_optimizers` (its own optimizers and schedulers), `on_train_start`, then `on_train_batch_start` / `training_step` /
`on_train_batch_end` per batch (:329-397, :441-455) — with the reference's own `Cameras`, `VanillaGaussian` model,
`VanillaMetrics` loss, `VanillaDensityController` and `VanillaOptStrategy`, and with THIS repository's renderer plugin selected the
way `--model.renderer gspl_amd.renderers.<Name>` does it (the constructor argument `renderer=`).

There is no GPU in the container that holds the reference tree and no reference tree on the GPU box, so the plugin's native op —
`ops.GaussianRasterizer`, the one C-ABI call of `HipVanillaRenderer` — is replaced here by the fp64-capable oracle pipeline
(`oracle.render_inria`), and `simple_knn._C.distCUDA2` by the oracle's k-d-tree version; everything else is the code a training run
executes.  Prints one JSON line: per-step loss and Gaussian count, what the density controller accumulated, final PSNR.
A second variant swaps nothing at all: the reference's own `GSPlatRenderer` (internal/renderers/gsplat_renderer.py, unedited) on
the `gsplat` stand-in package of `gspl_amd.compat`, whose ops are routed to the oracle stages the same way.
A third one selects the Gaussian-sharded multi-GPU plugin (`HipGSplatDistributedRenderer`, world size 1: `training_setup`, the
per-camera `projection_results_list` contract) together with the reference's `DistributedVanillaDensityController`.
With `<rank> <world> <port>` appended, `hip-distributed` runs as one rank of a gloo group: `training_setup` shards the model and its
optimizers, every step exchanges the visible splats' records with the peers (camera (step * world + rank) per rank, the peers'
cameras found through `trainer.train_dataloader.dataset.image_cameras` as the reference does), densification stays rank-local, and a
forced `redistribute` moves rows and their Adam moments between the ranks in the middle of the run.
usage: python reference_loop_worker.py <reference root> <steps> [hip-vanilla | hip-gsplat-v1 | reference-gsplat-on-shims | hip-distributed]
                                       [<rank> <world> <port>]
"""
import json
import math
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_ROOT, STEPS = sys.argv[1], int(sys.argv[2])
VARIANT = sys.argv[3] if len(sys.argv) > 3 else "hip-vanilla"
RANK, WORLD, PORT = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (0, 1, 0)
for p in (REF_ROOT, HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import lightning_standin  # noqa: E402

lightning_standin.install()

import gspl_amd  # noqa: E402,F401
from gspl_amd import compat, ops  # noqa: E402
from gspl_amd.renderers import HipVanillaRenderer  # noqa: E402
from oracle import gsplat_oracle as O  # noqa: E402
from oracle import knn_oracle  # noqa: E402
from oracle import training_oracle as T  # noqa: E402

compat.install()          # diff_gaussian_rasterization / gsplat / simple_knn / fused_ssim stand-ins, as a run with the plugins has them

from internal.gaussian_splatting import GaussianSplatting  # noqa: E402  (the reference's LightningModule, unchanged)
from internal.cameras.cameras import Cameras  # noqa: E402
from internal.configs.light_gaussian import LightGaussian  # noqa: E402
from internal.density_controllers.vanilla_density_controller import VanillaDensityController  # noqa: E402
from internal.metrics.vanilla_metrics import VanillaMetrics  # noqa: E402
from internal.models.vanilla_gaussian import VanillaGaussian  # noqa: E402
from internal.optimizers import Adam  # noqa: E402
from internal.schedulers import ExponentialDecayScheduler  # noqa: E402
from internal.renderers.renderer import Renderer as ReferenceRenderer  # noqa: E402

W_IMG, H_IMG, FOCAL = 160, 112, 150.0
EXTENT = 4.4


class OracleRasterizer:
    """Stands in for `ops.GaussianRasterizer` (csrc/fused.hip behind it): same arguments, same returns, `.grad` of the
    screen-space carrier in the Inria (NDC-scaled) units."""

    def __init__(self, raster_settings):
        self.s = raster_settings

    def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 shs_rest=None, raw_parameters=False):
        s = self.s
        if raw_parameters:      # the reference's VanillaGaussianModel qualifies (renderer.model_raw_parameters): the rasterizer owns the activations
            opacities, scales, rotations = torch.sigmoid(opacities), torch.exp(scales), torch.nn.functional.normalize(rotations)
        if shs_rest is not None:
            shs = torch.cat((shs, shs_rest), dim=1)
        r = O.render_inria(means3D, scales, rotations, opacities, shs, s.sh_degree, s.viewmatrix, s.projmatrix, s.campos,
                           s.tanfovx, s.tanfovy, s.image_width, s.image_height, s.bg)
        sc = torch.tensor([0.5 * s.image_width, 0.5 * s.image_height])
        if r["xy"].requires_grad:
            r["xy"].register_hook(lambda g: setattr(means2D, "grad", torch.cat([g * sc, torch.zeros_like(g[:, :1])], dim=1)))
        return r["render"], r["radii"]


def orbit_cameras(n=6):
    """The reference's own `Cameras` container: n views on a circle of radius 4 about the y axis, looking at the origin."""
    Rs, Ts = [], []
    for i in range(n):
        a = 2 * math.pi * i / n
        c, s = math.cos(a), math.sin(a)
        Rs.append(torch.tensor([[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]]))
        Ts.append(torch.tensor([0.0, 0.0, 4.0]))
    f = lambda v, dt=torch.float32: torch.full((n,), v, dtype=dt)
    return Cameras(R=torch.stack(Rs), T=torch.stack(Ts), fx=f(FOCAL), fy=f(FOCAL), cx=f(W_IMG / 2), cy=f(H_IMG / 2),
                   width=f(W_IMG, torch.int32), height=f(H_IMG, torch.int32), appearance_id=f(0, torch.int32),
                   normalized_appearance_id=f(0.0), distortion_params=None, camera_type=f(0, torch.int32))


def main():
    assert gspl_amd.renderers.renderer.INSIDE_REFERENCE, "the plugins must subclass the reference's own Renderer here"
    density_cls = VanillaDensityController
    if VARIANT == "hip-vanilla":
        plugin = HipVanillaRenderer()
    elif VARIANT == "hip-gsplat-v1":
        # configs/gsplat_v1.yaml with the plugin in place of GSplatV1Renderer; its ops -> oracle stages
        from gspl_amd.renderers import HipGSplatV1Renderer
        import test_distributed_renderer
        test_distributed_renderer._install_oracle_ops()          # projection, list-only binning, compositing -> oracle stages
        composite = ops.rasterize_to_pixels

        def rasterize_to_pixels(means2d, *args, track_hits=False, **kwargs):
            if track_hits:       # the fork's rasterizer leaves `has_hit_any_pixels` on the screen-space tensor (read as `acc_vis`)
                means2d.has_hit_any_pixels = torch.ones(means2d.shape[-2], dtype=torch.bool)
            return composite(means2d, *args, **kwargs)

        def sh_view_colors(degree, means, center, dc, rest, masks=None, detach_means=True):
            rgb = O.sh_colors(degree, dc if rest is None else torch.cat([dc, rest], dim=1), means, center, detach_dirs=True)
            return rgb if masks is None else torch.where(masks[:, None], rgb, torch.zeros((), dtype=rgb.dtype))
        ops.rasterize_to_pixels, ops.sh_view_colors = rasterize_to_pixels, sh_view_colors
        plugin = HipGSplatV1Renderer().instantiate()
    elif VARIANT == "hip-distributed":
        # configs/distributed.yaml: the sharded renderer + its density controller (one rank here; ops of its host path -> oracle stages)
        from gspl_amd.renderers import HipGSplatDistributedRenderer
        from internal.density_controllers.distributed_vanilla_density_controller import DistributedVanillaDensityController
        import test_distributed_renderer
        test_distributed_renderer._install_oracle_ops()
        if WORLD > 1:
            import torch.distributed as dist
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(PORT)
            dist.init_process_group("gloo", rank=RANK, world_size=WORLD)
        # a redistribution in the middle of the run (the reference's default interval is 1000 steps; any imbalance triggers it here)
        plugin = HipGSplatDistributedRenderer(redistribute_interval=90 if WORLD > 1 else 1000, redistribute_threshold=1.0 + 1e-9).instantiate()
        density_cls = DistributedVanillaDensityController
    else:
        # the reference's own class, importable here only because `gsplat` resolves to the stand-in package
        from internal.renderers.gsplat_renderer import GSPlatRenderer
        import gsplat
        assert "gspl_amd" in (gsplat.__doc__ or "")
        import test_package_shims

        class _Setter:
            def setattr(self, obj, name, value):
                setattr(obj, name, value)
        test_package_shims._oracle_ops(_Setter())          # gspl_amd.ops.<gsplat entry points> -> oracle stages (no GPU here)
        plugin = GSPlatRenderer()
    assert isinstance(plugin, ReferenceRenderer)          # what gaussian_splatting.py:75-77 relies on

    g = torch.Generator().manual_seed(9)
    n_gt = 600
    gt = dict(means=(torch.rand(n_gt, 3, generator=g) * 2 - 1) * 0.9, scales=torch.exp(torch.randn(n_gt, 3, generator=g) * 0.3 - 2.3),
              quats=torch.nn.functional.normalize(torch.randn(n_gt, 4, generator=g), dim=-1), opac=torch.rand(n_gt, 1, generator=g) * 0.5 + 0.45,
              shs=torch.cat([torch.randn(n_gt, 1, 3, generator=g) * 0.8, torch.randn(n_gt, 15, 3, generator=g) * 0.05], dim=1))
    cameras = orbit_cameras()
    bg = torch.zeros(3)
    targets = []
    with torch.no_grad():
        for cam in cameras:
            r = O.render_inria(gt["means"], gt["scales"], gt["quats"], gt["opac"], gt["shs"], 3, cam.world_to_camera, cam.full_projection,
                               cam.camera_center, math.tan(float(cam.fov_x) / 2), math.tan(float(cam.fov_y) / 2), W_IMG, H_IMG, bg)
            targets.append(r["render"].float().clamp(0, 1))
    # the "SfM point cloud" the run starts from: the ground-truth centres, jittered, with their base colours
    n0 = 2000
    pick = torch.randint(0, n_gt, (n0,), generator=g)
    xyz = (gt["means"][pick] + 0.05 * torch.randn(n0, 3, generator=g)).numpy()
    rgb = ((gt["shs"][pick, 0] * 0.28209479177387814 + 0.5).clamp(0, 1) * 255).numpy()

    # the native helpers on the CPU: oracle pipeline for the rasterizer op, k-d tree for distCUDA2
    ops.GaussianRasterizer = OracleRasterizer
    sys.modules["simple_knn._C"].distCUDA2 = lambda pts: torch.from_numpy(knn_oracle.mean_dist2_kdtree(pts.detach().cpu().numpy().astype(np.float64))).float()
    torch.Tensor.cuda = lambda self, *a, **k: self            # `setup_from_pcd` moves the points to "cuda" for distCUDA2 (vanilla_gaussian.py:124)

    out_dir = tempfile.mkdtemp(prefix="gspl_ref_loop_")
    density = density_cls(percent_dense=0.01, densification_interval=40, opacity_reset_interval=150, densify_from_iter=40,
                                       densify_until_iter=260, densify_grad_threshold=0.00012, cull_opacity_threshold=0.005)
    gaussian = VanillaGaussian(sh_degree=3)
    gaussian.optimization.sh_degree_up_interval = 60
    # what the reference's CLI (jsonargparse) makes of the `{"class_path": ...}` defaults of OptimizationConfig (vanilla_gaussian.py:30-51)
    gaussian.optimization.optimizer = Adam()
    gaussian.optimization.means_lr_scheduler = ExponentialDecayScheduler(lr_final=0.0000016, max_steps=STEPS)
    module = GaussianSplatting(light_gaussian=LightGaussian(), save_iterations=[], gaussian=gaussian, renderer=plugin,
                               metric=VanillaMetrics(), density=density, output_path=out_dir)
    assert module.renderer is plugin and module.automatic_optimization is False

    ns = lambda **kw: type("NS", (), kw)()
    datamodule = ns(point_cloud=ns(xyz=xyz, rgb=rgb), prune_extent=EXTENT,
                    dataparser_outputs=ns(camera_extent=EXTENT, train_set=ns(cameras=cameras, image_names=[f"{i:03d}" for i in range(len(cameras))]),
                                          val_set=ns(cameras=cameras)),
                    set_device=lambda device: None)
    trainer = lightning_standin.Trainer(datamodule, max_steps=STEPS)
    trainer.global_rank, trainer.world_size = RANK, WORLD
    loader = ns(dataset=ns(image_cameras=[cameras[i] for i in range(len(cameras))]))
    trainer.train_dataloader, trainer.val_dataloaders = loader, loader
    trainer.fit_setup(module)
    from gspl_amd import distributed as gdist
    lo, hi = gdist.shard_bounds(n0, WORLD, RANK)
    assert module.gaussian_model.get_xyz.shape[0] == hi - lo and len(trainer.raw_optimizers) >= 2      # (world 1: the whole model)
    n_redistributions = [0]
    if WORLD > 1:
        moved = plugin.random_redistribute
        plugin.random_redistribute = lambda m, destination=None: (n_redistributions.__setitem__(0, n_redistributions[0] + 1), moved(m, destination))[1]

    losses, counts, accum_max, radii_max, sh_degrees = [], [], 0.0, 0.0, []
    for i in range(STEPS):
        torch.manual_seed(1000 + i)                           # the split samples of a densification
        k = (i * WORLD + RANK) % len(cameras)
        batch = (cameras[k], (f"{k:03d}", targets[k], None), None)
        trainer.train_batch(module, batch, i)
        assert trainer.global_step == i + 1, (trainer.global_step, i)       # one count per batch, whatever the number of optimizers
        losses.append(module.logged["train/loss"])
        counts.append(int(module.gaussian_model.get_xyz.shape[0]))
        sh_degrees.append(int(module.gaussian_model.active_sh_degree))
        accum_max = max(accum_max, float(module.density_controller.xyz_gradient_accum.max()))
        radii_max = max(radii_max, float(module.density_controller.max_radii2D.max()))
    module.eval()
    with torch.no_grad():
        mine = [(j * WORLD + RANK) % len(cameras) for j in range(max(len(cameras) // WORLD, 1))]     # every rank renders ITS cameras, in step
        finals = [module(cameras[k])["render"] for k in mine]
    psnr = float(np.mean([T.psnr(f, targets[k]) for f, k in zip(finals, mine)]))
    lrs = [m for _, m in trainer.logger.metrics]
    print(json.dumps({"losses": losses, "counts": counts, "sh_degrees": sh_degrees, "accum_max": accum_max, "radii_max": radii_max, "psnr": psnr,
                      "logged_lr_rows": len(lrs), "means_lr_first": lrs[0].get("lr/0_means") if lrs else None,
                      "means_lr_last": float(trainer.raw_optimizers[0].param_groups[0]["lr"]),
                      "inside_reference": bool(gspl_amd.renderers.renderer.INSIDE_REFERENCE), "renderer": type(plugin).__module__ + "." + type(plugin).__name__,
                      "rank": RANK, "world": WORLD, "redistributions": n_redistributions[0],
                      "last_exchange": getattr(plugin, "last_exchange", None)}))
    if WORLD > 1:
        import torch.distributed as dist
        dist.barrier()
        # leave without the collective tear-down and without interpreter finalisation (conftest.leave_process_group: a gloo thread torn
        # down while the peers close their ends was seen to abort one rank of eight after everything had passed)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
