"""`bench.py --loop reference-shaped` (bench_loop.py): the bench-side restatement of the reference's consumer code must make the
decisions of the test-side one (oracle/training_oracle.py, itself pinned to the reference's real `VanillaGaussianModel` +
`VanillaDensityControllerImpl` on CPU by tests/test_training_loop.py) when both drive the same renderer plugin."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _setup(dev, n=20000, W=320, H=208):
    import gspl_amd  # noqa: F401
    from gspl_amd import synthetic
    from gspl_amd.renderers import HipVanillaRenderer
    import bench_loop as BL
    clean = synthetic.scene(n, seed=42)
    clean = (clean[0], clean[1] * 3.0, clean[2], clean[3], clean[4])
    cam_dicts = synthetic.camera_set(W, H, 300.0, count=4)
    cams = [synthetic.CameraObject(c, dev, idx=i) for i, c in enumerate(cam_dicts)]
    renderer = HipVanillaRenderer()
    bg = torch.zeros(3, device=dev)
    truth = synthetic.ModelObject(*[t.to(dev) for t in clean], active_sh_degree=3)
    with torch.no_grad():
        targets = [renderer(c, truth, bg)["render"].clone() for c in cams]
    return BL, renderer, cams, targets, bg, BL.perturbed(clean)


@pytest.mark.gpu
def test_bench_loop_matches_the_oracle_loop_n_trajectory():
    from gspl_amd.optimizers import FusedAdam
    from oracle import training_oracle as T
    dev = torch.device("cuda:0")
    BL, renderer, cams, targets, bg, start = _setup(dev)
    steps = 260
    cfg = dict(densify_from_iter=40, densification_interval=40, opacity_reset_interval=120)
    loss = lambda img, gt: (img - gt).abs().mean()

    torch.manual_seed(11)
    model = BL.RawGaussians(*[t.to(dev) for t in start], active_sh_degree=1)
    ctl = BL.DensityController(model.n_gaussians, dev, 2.6, **cfg)
    got = BL.run(renderer, model, ctl, model.make_optimizers(1.0, FusedAdam), cams, targets, steps, bg, loss, sh_degree_up_interval=100)

    torch.manual_seed(11)
    ref_model = T.TrainableGaussians(*[t.to(dev) for t in start], active_sh_degree=1)
    ref_ctl = T.DensityControllerOracle(ref_model.n_gaussians, dev, 2.6, **cfg)
    hist = T.train(ref_model, ref_ctl, ref_model.make_optimizers(1.0, cls=FusedAdam), lambda c, m, b: renderer(c, m, b), cams, targets, steps, bg,
                   sh_degree_up_interval=100)
    ref_n = [n for _, n in hist]

    changes = sum(1 for a, b in zip(got["n"], got["n"][1:]) if a != b)
    assert changes >= 4, f"N changed {changes} times only: {got['n'][::20]}"
    assert model.active_sh_degree == ref_model.active_sh_degree == 3
    # The two loops make the same decisions from the same numbers; the numbers themselves carry the run-to-run spread of the
    # backward's atomics (1e-7 relative on a gradient), which can move a splat across the densification threshold: the trajectories
    # agree to a fraction of a per cent, not necessarily to the splat.
    for i, (a, b) in enumerate(zip(got["n"], ref_n), start=1):
        assert abs(a - b) <= max(2, 0.005 * b), f"step {i}: N = {a} (bench loop) vs {b} (oracle loop)"
    assert got["loss"][-1] < got["loss"][0]
    assert any(e.get("opacity_reset") for e in ctl.events)


def test_bench_loop_controller_decisions_equal_the_oracle_controller_cpu():
    """Same state, same gradients statistics -> the same clone / split / prune / reset decisions, row for row (CPU, no renderer)."""
    import bench_loop as BL
    from oracle import training_oracle as T
    from oracle import gsplat_oracle as O
    means, scales, quats, opac, shs = O.synthetic_scene(3000, seed=4)
    scales = scales * 4
    dev = torch.device("cpu")

    def build(kind):
        if kind == "bench":
            m = BL.RawGaussians(means, scales, quats, opac, shs)
            c = BL.DensityController(m.n_gaussians, dev, 2.6, densify_from_iter=0, densification_interval=1, opacity_reset_interval=2, fused_stats=False)
            opts = m.make_optimizers(1.0, torch.optim.Adam)
        else:
            m = T.TrainableGaussians(means, scales, quats, opac, shs)
            c = T.DensityControllerOracle(m.n_gaussians, dev, 2.6, densify_from_iter=0, densification_interval=1, opacity_reset_interval=2)
            opts = m.make_optimizers(1.0)
        return m, c, opts

    results = []
    for kind in ("bench", "oracle"):
        m, c, opts = build(kind)
        torch.manual_seed(5)
        for step in (1, 2, 3):
            N = m.n_gaussians
            g = torch.Generator().manual_seed(100 + step)
            vp = torch.zeros(N, 3, requires_grad=True)
            vp.grad = torch.randn(N, 3, generator=g) * 3e-4
            radii = torch.randint(0, 40, (N,), generator=g, dtype=torch.int32)
            outputs = {"viewspace_points": vp, "visibility_filter": radii > 0, "radii": radii}
            for o in opts:      # give the moments something to carry through the surgery
                for grp in o.param_groups:
                    p = grp["params"][0]
                    p.grad = torch.ones_like(p) * 1e-3
                o.step()
                o.zero_grad(set_to_none=True)
            c.after_backward(outputs, m, opts, step)
        results.append({k: v.detach().clone() for k, v in m.properties.items()} | {"accum": c.xyz_gradient_accum.clone(), "maxr": c.max_radii2D.clone()})
    a, b = results
    assert a["means"].shape[0] != 3000
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k


@pytest.mark.gpu
def test_bench_loop_with_the_optimizers_inside_the_backward():
    """The same loop with both optimizers built with fuse_into_backward=True (tests/test_fused_backward_adam.py has the bit-level
    equality on a static model): raw parameters of TWO optimizers claimed together by the rasterizer's backward, N changing through
    densification / pruning (the surgery's new Parameters and moments are found by address on the next backward), an opacity reset, SH
    degree raises.  No parameter ever carries a gradient; every step counter advances once per step; the run trains as the two-kernel
    path does — not splat for splat: on a densification step the reference drops the step's gradients (the surgery replaces the
    Parameters before `step()`), here they have already been applied."""
    from gspl_amd.optimizers import FusedAdam
    dev = torch.device("cuda:0")
    BL, renderer, cams, targets, bg, start = _setup(dev)
    steps = 260
    cfg = dict(densify_from_iter=40, densification_interval=40, opacity_reset_interval=120)
    loss = lambda img, gt: (img - gt).abs().mean()

    def run(fuse):
        torch.manual_seed(11)
        model = BL.RawGaussians(*[t.to(dev) for t in start], active_sh_degree=1)
        ctl = BL.DensityController(model.n_gaussians, dev, 2.6, **cfg)
        opts = model.make_optimizers(1.0, FusedAdam, **({"fuse_into_backward": True} if fuse else {}))
        seen = []

        def on_step(step, outputs):
            if fuse:
                seen.append(all(p.grad is None for o in opts for g in o.param_groups for p in g["params"]))
        res = BL.run(renderer, model, ctl, opts, cams, targets, steps, bg, loss, sh_degree_up_interval=100, on_step=on_step)
        counters = {int(o.state[p]["step"]) for o in opts for g in o.param_groups for p in g["params"] if p in o.state}
        return res, model, ctl, seen, counters

    plain, _, _, _, _ = run(False)
    fused, model, ctl, seen, counters = run(True)
    assert seen and all(seen), "a parameter carried a gradient although the backward applies the updates"
    assert sum(1 for a, b in zip(fused["n"], fused["n"][1:]) if a != b) >= 4 and any(e.get("opacity_reset") for e in ctl.events)
    assert model.active_sh_degree == 3
    assert fused["loss"][-1] < fused["loss"][0] and abs(fused["loss"][-1] - plain["loss"][-1]) <= 0.15 * plain["loss"][-1]
    for i, (a, b) in enumerate(zip(fused["n"], plain["n"]), start=1):
        assert abs(a - b) <= max(50, 0.03 * b), f"step {i}: N = {a} (optimizers inside the backward) vs {b} (two-kernel path)"
    assert max(counters) <= steps      # (moments of split / cloned rows keep their parameter's counter; nothing advanced twice)
