"""The Gaussian-sharded renderer end to end with two ranks (reference: internal/renderers/gsplat_distributed_renderer.py).

Two processes shard a seeded scene with `shard_bounds`, each renders its own camera through
`HipGSplatDistributedRendererImpl.forward` (batched projection + batched SH, packed all-to-all, list-only binning,
compositing) and back-propagates its own image loss.  Checked against the one-process result on the full model:
the render of each rank's camera, the gradient of EVERY parameter row of the shard (= gradient of the sum of both losses),
and the per-camera screen-space gradients the distributed density controller reads.  Then `training_setup` (sharding with
optimizer surgery) and `random_redistribute` (Adam moments must follow their parameters).

  * `-m gpu`     : both processes share cuda:0 (gloo, payload staged through the host — RCCL refuses two ranks on one
                   device); every op is the HIP op; checked against the fp64 oracle of the full model (`oracle.render_gsplat`)
                   AND against the one-process `HipGSplatV1Renderer`, both with the tiered tolerances of hip_helpers.
  * `-m "not gpu"`: CPU processes; the HIP ops are replaced IN THIS TEST by the fp64 oracle stages, so the renderer's
                   host logic, the exchange and its autograd route run without a GPU; reference = `oracle.render_gsplat`.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
N_SPLATS, W_IMG, H_IMG = 3000, 208, 144


def _cameras(world=2):
    from oracle import gsplat_oracle as O
    cams = []
    poses = (((0.0, 0.0, 4.0), 190.0), ((0.35, -0.2, 3.4), 205.0), ((-0.3, 0.25, 4.6), 180.0), ((0.1, 0.3, 3.8), 198.0),
             ((-0.45, -0.3, 4.2), 186.0), ((0.5, 0.15, 5.0), 214.0), ((0.0, -0.4, 3.2), 176.0), ((-0.2, 0.1, 5.6), 222.0))
    assert world <= len(poses)
    for i, (t, f) in enumerate(poses[:world]):
        cam = O.synthetic_camera(W_IMG, H_IMG, f, f + 3.0)
        w2c = cam["world_to_camera"].clone()
        w2c[3, :3] = torch.tensor(t)
        cam["world_to_camera"] = w2c
        cam["camera_center"] = torch.linalg.inv(w2c)[3, :3]
        cam["idx"] = i
        cams.append(cam)
    return cams


def _scene(dtype):
    from oracle import gsplat_oracle as O
    means, scales, quats, opac, shs = O.synthetic_scene(N_SPLATS, seed=77)
    scales = scales * 5
    return [t.to(dtype) for t in (means, scales, quats, opac, shs)]


def _weights(dtype, world=2):
    g = torch.Generator().manual_seed(3)
    return [torch.randn(3, H_IMG, W_IMG, generator=g).to(dtype) for _ in range(world)]


def _install_oracle_ops():
    """CPU variant: route the op entry points the sharded renderer uses to the oracle stages (test only)."""
    from oracle import gsplat_oracle as O
    from gspl_amd import ops

    def fully_fused_projection(means, covars, quats, scales, viewmats, Ks, width, height, eps2d=0.3, calc_compensations=False,
                               packed=False, **kw):
        outs = []
        for c in range(viewmats.shape[0]):
            K = Ks[c]
            xys, depths, radii, conics, comp, _, _, _, _, _ = O.project_gaussians(
                means, scales, 1.0, quats, viewmats[c].T, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                int(height), int(width), eps2d=eps2d)
            outs.append((radii, xys, depths, conics, comp))
        return tuple(torch.stack([o[i] for o in outs]) for i in range(5))

    def sh_view_colors_batched(degree, means, centers, dc, rest, radii=None):
        coeffs = torch.cat([dc, rest], dim=1)
        cols = []
        for c in range(centers.shape[0]):
            rgb = O.sh_colors(degree, coeffs, means, centers[c], detach_dirs=True)
            if radii is not None:
                rgb = torch.where((radii[c] > 0)[:, None], rgb, torch.zeros((), dtype=rgb.dtype))
            cols.append(rgb)
        return torch.stack(cols)

    def bin_gaussians(xys, depths, radii, img_height, img_width, block_width=16, mode=0, conics=None, opacities=None, lazy=False):
        _, _, flat, offs = O.isect_tiles(O.MODE_GSPLAT, xys.detach(), radii, depths.detach(), img_width, img_height)
        return torch.from_numpy(np.asarray(flat)), torch.from_numpy(np.asarray(offs))

    def rasterize_to_pixels(means2d, conics, colors, opacities, image_width, image_height, tile_size, isect_offsets, flatten_ids,
                            backgrounds=None, absgrad=False, channels_first=False, **kw):
        out, alpha = O.composite_c(O.MODE_GSPLAT, means2d.reshape(-1, 2), conics.reshape(-1, 3), colors.reshape(-1, colors.shape[-1]),
                                   opacities.reshape(-1), backgrounds.reshape(-1), image_width, image_height,
                                   isect_offsets.reshape(-1).numpy(), flatten_ids.numpy())
        if channels_first:
            out = out.permute(2, 0, 1)
        return out[None], alpha[None, ..., None]

    ops.fully_fused_projection = fully_fused_projection
    ops.sh_view_colors_batched = sh_view_colors_batched
    ops.bin_gaussians = bin_gaussians
    ops.rasterize_to_pixels = rasterize_to_pixels


def _reference_full_model(on_gpu, params, cams, weights, bg, dev, return_renders=False):
    """One-process result: every camera rendered from the FULL model, loss = sum over cameras of <render, weight>."""
    from fakes import FakeCamera, FakeGaussianModel
    if on_gpu:
        from gspl_amd.renderers import HipGSplatV1Renderer
        model = FakeGaussianModel(*[p.clone().to(dev) for p in params])
        renderer = HipGSplatV1Renderer(anti_aliased=True).instantiate()
        renders, xy_grads = [], []
        loss = 0
        outs = []
        for cam, w in zip(cams, weights):
            out = renderer(FakeCamera(cam, dev), model, bg.to(dev))
            out["viewspace_points"].retain_grad()
            outs.append(out)
            loss = loss + (out["render"] * w.to(dev)).sum()
        loss.backward()
        grads = [model.means.grad, model.scales_.grad, model.rotations_.grad, model.opacities_.grad, model.shs_dc.grad, model.shs_rest.grad]
        return [o["render"].detach().cpu() for o in outs], [g.cpu() for g in grads], [o["viewspace_points"].grad.cpu() for o in outs]
    from oracle import gsplat_oracle as O
    leaves = [p.clone().requires_grad_(True) for p in params]
    m, s, q, o, c = leaves
    loss = 0
    rs = []
    for cam, w in zip(cams, weights):
        r = O.render_gsplat(m, s, q, o, c, 3, cam["world_to_camera"].to(m.dtype), cam["fx"], cam["fy"], cam["cx"], cam["cy"], W_IMG, H_IMG,
                            bg, cam["camera_center"].to(m.dtype))
        rs.append(r)
        loss = loss + (r["render"] * w).sum()
    loss.backward()
    grads = [m.grad, s.grad, q.grad, o.grad, c.grad[:, :1], c.grad[:, 1:]]
    if return_renders:
        return [r["render"].detach() for r in rs], grads, [r["xys"].grad for r in rs], rs
    return [r["render"].detach() for r in rs], grads, [r["xys"].grad for r in rs]


class _Trainer:
    def __init__(self, world, rank):
        self.world_size, self.global_rank = world, rank
        self.profiler = None


class _Module:
    """What the renderer touches of the LightningModule (gaussian_splatting.py): trainer, gaussian_model,
    gaussian_optimizers, density_updated_by_renderer, device."""

    def __init__(self, model, optimizers, world, rank, dev):
        self.trainer, self.gaussian_model, self.gaussian_optimizers, self.device = _Trainer(world, rank), model, optimizers, dev
        self.density_changes = 0

    def density_updated_by_renderer(self):
        self.density_changes += 1


def _close(got, ref, rel, name, rows=None):
    """CPU variant (fp64 oracle ops on both sides): every element within `rel`.  GPU variant (`rows` = the shard's rows a fragile
    decision of the fp64 oracle reaches, hip_helpers.fragile_rows): ATTRIBUTED — every element beyond the tolerance belongs to such a
    row (a splat whose alpha meets the 1/255 threshold within the fp32-vs-fp64 difference gains or loses one pixel's contribution),
    and even there nothing is beyond 0.05 (hip_helpers.assert_close_attributed; a quota of a 375-row shard would be zero elements)."""
    from hip_helpers import assert_close_scaled, assert_close_attributed
    g, r = got.detach().cpu().double().numpy(), ref.detach().cpu().double().numpy()
    if rows is not None:
        assert_close_attributed(g, r, rel, name, rows, frac_ok=0.99, rel_all=0.05, rel_firm=5 * rel, frac_firm=2e-5, quiet=True)
    else:
        assert_close_scaled(g, r, rel, name, frac_ok=1.0)


def _worker(rank, world, port, tmpdir, on_gpu, exchange="counted"):
    if world > 2:          # the ranks share the host's cores (and the OpenMP oracle would start a team per process)
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 2) // world))
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
    parts = exchange.split("-")                           # "padded-staged": the stage-by-stage formulation of the step;
    exchange, form = parts[0], ("staged" if "staged" in parts else "")      # "padded-peer": direct peer writes instead of the collective
    transport = "peer" if "peer" in parts else "collective"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (HERE, os.path.dirname(HERE)):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gspl_amd  # noqa: F401
        from gspl_amd import distributed as D
        from gspl_amd.renderers import HipGSplatDistributedRenderer
        from fakes import FakeCamera, FakePropertyModel
        dev = torch.device("cuda:0") if on_gpu else torch.device("cpu")
        dtype = torch.float32 if on_gpu else torch.float64
        if not on_gpu:
            _install_oracle_ops()
        params = _scene(dtype)
        cams, weights = _cameras(world), _weights(dtype, world)
        bg = torch.tensor([0.2, 0.1, 0.4], dtype=dtype)
        N = params[0].shape[0]
        lo, hi = D.shard_bounds(N, world, rank)

        # ---- training_setup: the full model is cut down to this rank's rows, optimizer parameters replaced, moments reset
        ids = torch.arange(N, dtype=torch.float32)
        model = FakePropertyModel(*[p.clone().to(dev) for p in params], extra={"ids": ids.to(dev)})
        opt = model.named_optimizer()
        for g in opt.param_groups:      # give the optimizer a state to reset
            p = g["params"][0]
            opt.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.ones_like(p), "exp_avg_sq": torch.ones_like(p)}
        module = _Module(model, [opt], world, rank, dev)
        renderer = HipGSplatDistributedRenderer(exchange=exchange, fused_step=(form != "staged"), auto_padded_with_peers=True,
                                                exchange_transport=transport).instantiate()
        assert renderer.training_setup(module) == (None, None)
        assert module.density_changes == 1 and renderer.world_size == world and renderer.global_rank == rank
        assert model.n_gaussians == hi - lo and torch.equal(model.get_property("ids").cpu(), ids[lo:hi])
        for g in opt.param_groups:
            p = g["params"][0]
            assert p is model.get_property(g["name"]) and p.shape[0] == hi - lo and p.requires_grad
            st = opt.state[p]
            assert st["exp_avg"].shape == p.shape and float(st["exp_avg"].abs().sum()) == 0 and float(st["exp_avg_sq"].abs().sum()) == 0
        assert torch.equal(model.get_xyz.detach().cpu(), params[0][lo:hi])

        # ---- forward / backward against the one-process result
        camset = [FakeCamera(c, dev) for c in cams]
        renderer.camera_lookup = lambda idx, training: camset[idx]
        renderer.train()
        out = renderer(camset[rank], model, bg.to(dev))
        assert set(out) == {"render", "hard_inverse_depth", "cameras", "projection_results_list", "visible_mask_list", "xys_grad_scale_required"}
        assert out["render"].shape == (3, H_IMG, W_IMG) and len(out["cameras"]) == world and out["xys_grad_scale_required"] is True
        assert [int(c.idx) for c in out["cameras"]] == list(range(world))
        # "auto" starts with the counted exchange (no rank has a visible share to vote with yet) and moves to the padded one when
        # every rank saw at least half of its (camera, splat) pairs; the other two settings are fixed
        assert renderer.last_exchange == ("counted" if exchange == "auto" else exchange)      # (auto: nobody has voted yet)
        if transport == "peer":
            # the records of this step went through the peers' IPC-mapped buffers, not through a collective: the image must be the
            # one the collective route composes from the same records, bit for bit (the forward pass is deterministic)
            assert renderer._peer is not None and renderer._peer.step == 1 and renderer._peer.world == world
            via_collective = HipGSplatDistributedRenderer(exchange=exchange, fused_step=True).instantiate()
            via_collective.world_size, via_collective.global_rank = world, rank
            via_collective.camera_lookup = renderer.camera_lookup
            via_collective.train()
            other = via_collective(camset[rank], model, bg.to(dev))["render"]
            assert via_collective._peer is None and torch.equal(other, out["render"])
        assert [r[1] for r in renderer._peer_rows] == [D.shard_bounds(N, world, r)[1] - D.shard_bounds(N, world, r)[0] for r in range(world)]
        for r in out["projection_results_list"]:          # what DistributedVanillaDensityControllerImpl.before_backward does
            r[1].retain_grad()
        (out["render"] * weights[rank].to(dev)).sum().backward()
        ref_renders, ref_grads, ref_xy = _reference_full_model(on_gpu, params, cams, weights, bg, dev)
        names = ("means", "scales", "rotations", "opacities", "shs_dc", "shs_rest")
        if on_gpu:
            # (1) against the fp64 ORACLE of the one-process pipeline (oracle.render_gsplat on the full model, as the CPU variant);
            # (2) against the one-process HIP renderer (same kernels, unsharded): sharding must not change anything beyond the
            # order of the fp32 atomics.
            from hip_helpers import assert_pixels_attributed, fragile_rows
            from oracle import gsplat_oracle as O
            p64 = [p.double() for p in params]
            orc_renders, orc_grads, orc_xy, orc_rs = _reference_full_model(False, p64, cams, [w.double() for w in weights], bg.double(), torch.device("cpu"),
                                                                          return_renders=True)
            # rows a fragile decision reaches, per camera (the parameter gradients sum over the cameras: their union); plus this shard's
            # rows whose integer extent the two sides disagree on
            per_cam = [fragile_rows(O.MODE_GSPLAT, r_, W_IMG, H_IMG, bg.double()) for r_ in orc_rs]
            cam_rows = []
            for i, (rows_i, _) in enumerate(per_cam):
                local = out["projection_results_list"][i][0].cpu().numpy().reshape(-1) != orc_rs[i]["radii"].numpy().reshape(-1)[lo:hi]
                cam_rows.append(rows_i[lo:hi] | local)
            all_rows = np.logical_or.reduce(cam_rows)
            assert_pixels_attributed(out["render"].detach().cpu().double().numpy(), orc_renders[rank].numpy(), per_cam[rank][1], tol=1e-5,
                                     name="render vs fp64 oracle")
            assert float((out["render"].detach().cpu() - ref_renders[rank]).abs().max()) <= 2e-5
            for name, ref, orc in zip(names, ref_grads, orc_grads):
                got = model.get_property(name).grad
                assert got is not None and got.shape[0] == hi - lo, name
                _close(got, orc[lo:hi], 2e-4, name + " vs fp64 oracle", rows=all_rows)
                _close(got, ref[lo:hi], 2e-4, name + " vs one-process HIP", rows=all_rows)
            for i, r in enumerate(out["projection_results_list"]):
                assert torch.equal(out["visible_mask_list"][i], r[0] > 0)
                _close(r[1].grad, orc_xy[i].reshape(N, 2)[lo:hi], 2e-4, f"xys grad of camera {i} vs fp64 oracle", rows=cam_rows[i])
                _close(r[1].grad, ref_xy[i].reshape(N, 2)[lo:hi], 2e-4, f"xys grad of camera {i} vs one-process HIP", rows=cam_rows[i])
        else:
            diff = (out["render"].detach().cpu() - ref_renders[rank]).abs()
            assert float(diff.max()) <= 1e-9, float(diff.max())
            for name, ref in zip(names, ref_grads):
                got = model.get_property(name).grad
                assert got is not None and got.shape[0] == hi - lo, name
                _close(got, ref[lo:hi], 1e-8, name)
            for i, r in enumerate(out["projection_results_list"]):
                assert torch.equal(out["visible_mask_list"][i], r[0] > 0)
                _close(r[1].grad, ref_xy[i].reshape(N, 2)[lo:hi], 1e-8, f"xys grad of camera {i}")
        # ---- the reference's own DistributedVanillaDensityControllerImpl consumes these outputs (lightning stubbed; only where the
        # reference tree exists — not on the GPU box): its statistics buffers equal the sums computed by hand
        ref_root = os.environ.get("GSPL_REFERENCE_ROOT", "/root/reference")
        if os.path.exists(os.path.join(ref_root, "internal", "density_controllers", "distributed_vanilla_density_controller.py")):
            import types
            if "lightning" not in sys.modules:
                Lm = types.ModuleType("lightning")
                Lm.LightningModule = type("LightningModule", (), {})
                sys.modules["lightning"] = Lm
            if ref_root not in sys.path:
                sys.path.insert(0, ref_root)
            from internal.density_controllers.distributed_vanilla_density_controller import DistributedVanillaDensityController
            ctrl = DistributedVanillaDensityController().instantiate()
            n_local = hi - lo
            ctrl._init_state(n_local, dev)
            ctrl.max_radii2D, ctrl.xyz_gradient_accum, ctrl.denom = (b.to(dtype) for b in (ctrl.max_radii2D, ctrl.xyz_gradient_accum, ctrl.denom))
            with torch.no_grad():
                ctrl.update_states(out)
            exp_accum = torch.zeros(n_local, 1, dtype=dtype, device=dev)
            exp_denom = torch.zeros(n_local, 1, dtype=dtype, device=dev)
            exp_radii = torch.zeros(n_local, dtype=dtype, device=dev)
            scale = 0.5 * torch.tensor([[W_IMG, H_IMG]], dtype=dtype, device=dev)
            for i, r in enumerate(out["projection_results_list"]):
                vis = out["visible_mask_list"][i]
                exp_accum[vis] += torch.norm(r[1].grad[vis, :2] * scale, dim=-1, keepdim=True)
                exp_denom[vis] += 1
                exp_radii[vis] = torch.max(exp_radii[vis], r[0][vis].to(dtype))
            assert torch.equal(ctrl.denom, exp_denom) and torch.equal(ctrl.max_radii2D, exp_radii)
            assert torch.allclose(ctrl.xyz_gradient_accum, exp_accum, rtol=1e-6, atol=0) and float(ctrl.denom.sum()) > 0
        # hard inverse depth render type is produced and finite
        with torch.no_grad():
            d = renderer(camset[rank], model, bg.to(dev), render_types=["rgb", "hard_inverse_depth"])
        assert d["hard_inverse_depth"].shape == (1, H_IMG, W_IMG) and bool(torch.isfinite(d["hard_inverse_depth"]).all())
        if exchange == "auto":                # every rank has voted by now: > 50 % of the pairs are visible in this scene
            for _ in range(3):                # (on the GPU the visible count arrives through pinned memory, a step or two later)
                with torch.no_grad():
                    again = renderer(camset[rank], model, bg.to(dev))["render"]
                torch.cuda.synchronize() if on_gpu else None
            assert renderer.last_exchange == "padded" and min(r[2] for r in renderer._peer_rows) >= 500
            assert float((again.cpu() - ref_renders[rank]).abs().max()) <= (2e-5 if on_gpu else 1e-9)

        # ---- random_redistribute: every row (parameter AND Adam moments AND non-optimised property) follows its id
        with torch.no_grad():
            for g in opt.param_groups:
                p = g["params"][0]
                row = model.get_property("ids").reshape(-1, *([1] * (p.dim() - 1)))
                opt.state[p]["exp_avg"] = torch.zeros_like(p) + row * 0.5
                opt.state[p]["exp_avg_sq"] = torch.zeros_like(p) + row * 0.25
        before = module.density_changes
        gen = torch.Generator().manual_seed(1234 + rank)
        destination = torch.randint(0, world, (model.n_gaussians,), generator=gen).to(dev)
        renderer.random_redistribute(module, destination=destination)
        assert module.density_changes == before + 1
        new_ids = model.get_property("ids").cpu()
        counts = D.gather_ints(model.n_gaussians, dev)
        assert sum(counts) == N
        exp_ids = []
        for src in range(world):
            slo, shi = D.shard_bounds(N, world, src)
            dsrc = torch.randint(0, world, (shi - slo,), generator=torch.Generator().manual_seed(1234 + src))
            exp_ids.append(ids[slo:shi][dsrc == rank])
        assert torch.equal(new_ids, torch.cat(exp_ids))
        full = {"means": params[0], "scales": params[1], "rotations": params[2], "opacities": params[3],
                "shs_dc": params[4][:, :1], "shs_rest": params[4][:, 1:]}
        for g in opt.param_groups:
            p = g["params"][0]
            assert p is model.get_property(g["name"]) and p.requires_grad and p.shape[0] == new_ids.shape[0]
            assert torch.equal(p.detach().cpu(), full[g["name"]][new_ids.long()])
            st = opt.state[p]
            row = new_ids.reshape(-1, *([1] * (p.dim() - 1))).to(p.dtype)
            assert torch.equal(st["exp_avg"].cpu(), torch.zeros(p.shape, dtype=p.dtype) + row * 0.5)
            assert torch.equal(st["exp_avg_sq"].cpu(), torch.zeros(p.shape, dtype=p.dtype) + row * 0.25)
            assert float(st["step"]) == 3.0
        # the rebalanced shards still render the same image
        with torch.no_grad():
            again = renderer(camset[rank], model, bg.to(dev))["render"]
        # (the rows are permuted now: splats whose depths agree to the 32 bits of the sort key change their order — ties are broken
        # by row index, as in the reference — which moves single pixels by ~1e-7 even in the fp64 CPU variant)
        assert float((again.cpu() - ref_renders[rank]).abs().max()) <= (2e-5 if on_gpu else 2e-6), float((again.cpu() - ref_renders[rank]).abs().max())

        # ---- after_training_step honours interval / until / threshold
        renderer.config.redistribute_interval, renderer.config.redistribute_until = 10, 100
        calls = []
        renderer.random_redistribute = lambda m, destination=None: calls.append(1)
        renderer.after_training_step(5, module)          # not on the interval
        renderer.after_training_step(200, module)        # past `until`
        renderer.config.redistribute_threshold = 100.0   # balanced enough
        renderer.after_training_step(10, module)
        assert calls == []
        renderer.config.redistribute_threshold = 1.0 + 1e-6
        renderer.after_training_step(20, module)
        assert calls == ([1] if min(counts) * renderer.config.redistribute_threshold < max(counts) else [])
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    except BaseException:
        dist.destroy_process_group()
        raise
    from conftest import leave_process_group
    leave_process_group(dist)


def _run(tmp_path, on_gpu, world=2, exchange="counted"):
    from conftest import free_port
    port = free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), on_gpu, exchange), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("exchange", ["auto", "padded"])
def test_world2_sharded_renderer_cpu_oracle_ops(tmp_path, exchange):
    """auto: the graded step uses the counted exchange (the reference's scheme), later steps the padded one; padded: the graded
    step itself sends one record per (camera, local splat) — same render, same gradients."""
    _run(tmp_path, False, exchange=exchange)


def test_world3_sharded_renderer_cpu_oracle_ops(tmp_path):
    """Three ranks: uneven shards (3000 = 1000 + 1000 + 1000 here, but the random redistribution leaves uneven ones), three
    cameras per projection batch, three-way all-to-all."""
    _run(tmp_path, False, world=3, exchange="padded")


@pytest.mark.parametrize("exchange", ["auto", "padded"])
def test_world8_sharded_renderer_cpu_oracle_ops(tmp_path, exchange):
    """World size 8 — the size of BASELINE configs[3] / [4] and of `configs/distributed.yaml` on an 8-GPU node
    (gsplat_distributed_renderer.py:141-202,252-311,435-510): eight gloo ranks, 375-row shards, eight cameras per projection batch, an
    eight-way all-to-all in both directions, a redistribution with Adam rows over eight destinations.  Same bar as W = 2: every
    rank's image and every gradient row of its shard equal the one-process result on the full model."""
    _run(tmp_path, False, world=8, exchange=exchange)


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["auto", "padded-peer", "counted-peer"])
def test_world8_sharded_renderer_shared_gpu(tmp_path, exchange):
    """Eight processes on cuda:0: the 8 x 8 flag matrix of csrc/peer.hip, eight IPC-mapped receive buffers per
    process, the 8-row mailboxes and count matrix, the 8-camera batched projection / SH launch — at the size they exist for.  Same
    bar as the two-process test: the peer transports' image bit-equal to the collective route's, gradients home to their owners
    (against the fp64 oracle and the one-process HIP renderer), one forced redistribution with Adam rows."""
    _run(tmp_path, True, world=8, exchange=exchange)


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["auto", "padded", "auto-staged", "padded-peer", "counted-peer"])
def test_world2_sharded_renderer_shared_gpu(tmp_path, exchange):
    """On the GPU the step runs as three autograd nodes (ops.sharded_front / sharded_exchange / sharded_back); "auto-staged" keeps
    the stage-by-stage formulation (what a subclass overriding `get_rgbs` and the extra render types take) under the same checks;
    "padded-peer": the two processes (sharing the GPU) write their records straight into each other's IPC-mapped receive buffers."""
    _run(tmp_path, True, exchange=exchange)


def test_exchange_format_is_a_function_of_the_gathered_rows_only():
    """Every rank must pick the same format from the same gathered rows (a mismatch would pair a padded send with a counted
    receive): fixed settings ignore the votes; "auto" is padded only when EVERY rank has voted at least the threshold, and a
    rank that has not seen a step yet (-1) keeps everybody on the counted format."""
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers import HipGSplatDistributedRenderer
    rows = lambda *votes: [[i, 1000, v] for i, v in enumerate(votes)]
    r = HipGSplatDistributedRenderer(exchange="auto", padded_min_visible=0.5, auto_padded_with_peers=True).instantiate()
    r._world = lambda: 2
    for votes, expect in (((-1, -1), "counted"), ((900, -1), "counted"), ((900, 499), "counted"), ((500, 500), "padded"),
                          ((1000, 730, 651), "padded"), ((1000, 730, 0), "counted")):
        r._peer_rows = rows(*votes)
        assert r._exchange_format() == expect, votes
    for fixed in ("counted", "padded"):
        r = HipGSplatDistributedRenderer(exchange=fixed).instantiate()
        for votes in ((-1, -1), (1000, 1000), (0, 0)):
            r._peer_rows = rows(*votes)
            assert r._exchange_format() == fixed
    # with peers the default "auto" stays on the reference's counted scheme whatever the votes (the padded bytes on a real
    # interconnect are unmeasured: ADVICE r3); with one rank it follows the vote
    r = HipGSplatDistributedRenderer(exchange="auto", padded_min_visible=0.5).instantiate()
    r._peer_rows = rows(1000, 1000)
    r._world = lambda: 2
    assert r._exchange_format() == "counted"
    r._world = lambda: 1
    r._peer_rows = rows(1000)
    assert r._exchange_format() == "padded"
    with pytest.raises(ValueError):
        HipGSplatDistributedRenderer(exchange="compressed").instantiate()
    assert HipGSplatDistributedRenderer().exchange == "auto"             # counted (the reference's scheme) until every rank has voted


def test_camera_batch_cache_is_keyed_on_tensor_identity_and_version():
    """the stacked view matrices of a camera set must never be served for other cameras.  The cache entry keeps its
    source tensors (so their addresses cannot be handed out again while it lives) and is valid only for the same tensor objects
    at the same version: a new camera object misses, an in-place pose edit misses."""
    import gspl_amd  # noqa: F401
    from gspl_amd.renderers import HipGSplatDistributedRenderer
    from fakes import FakeCamera
    r = HipGSplatDistributedRenderer().instantiate()
    dev = torch.device("cpu")
    cams = [FakeCamera(c, dev) for c in _cameras(2)]
    a = r._camera_batch(cams, dev)
    assert r._camera_batch(cams, dev) is a                               # same objects, same versions: served from the cache
    assert torch.equal(a[0][1], cams[1].world_to_camera.T) and torch.equal(a[2][0], cams[0].camera_center)
    fresh = [FakeCamera(c, dev) for c in _cameras(2)]                    # equal values, different tensor objects
    fresh[1].world_to_camera = fresh[1].world_to_camera.clone()
    fresh[1].world_to_camera[3, 0] += 0.25
    b = r._camera_batch(fresh, dev)
    assert b is not a and float(b[0][1][0, 3]) == pytest.approx(float(a[0][1][0, 3]) + 0.25)
    cams[0].world_to_camera[3, 2] += 1.0                                 # in-place edit: version counter moves
    c = r._camera_batch(cams, dev)
    assert c is not a and float(c[0][0][2, 3]) == pytest.approx(float(a[0][0][2, 3]) + 1.0)
    held = [e[0] for e in r._camera_batches.values()]
    assert any(any(v is cams[0].world_to_camera for v in src) for src in held)      # entries hold their sources
