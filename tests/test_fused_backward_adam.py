"""`FusedAdam(fuse_into_backward=True)` / `gspl_rasterize_inria_bwd_adam`: the Adam update applied by the kernels that end the fused Inria
backward (SH backward, preprocess backward) instead of by a separate optimizer launch over gradients written to HBM.

Checked against the two-kernel path it replaces (the backward writes the gradients, `FusedAdam.step()` = `selective_adam_kernel` reads
them back — what internal/optimizers.py:14-22 / internal/models/vanilla_gaussian.py:266-300 amount to behind
gaussian_splatting.py:380-397), which tests/test_adam.py pins to the oracle and to torch.optim.Adam:
  * deterministic compositing backward (`ops.set_deterministic`): parameters AND both moments BIT-EQUAL after 40 steps over several
    cameras, activated leaves and the reference model's raw parameters (activations inside the preprocess kernels) alike;
  * regular mode: equal within the spread of the compositing backward's fp32 atomics;
  * the contract: no `.grad` on an updated parameter, `step()` applies nothing twice, a second backward before `step()` raises, a
    parameter that is not the optimizer's (or has a gradient waiting) sends the whole backward down the two-kernel path.
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ("means", "scales", "rotations", "opacities", "shs_dc", "shs_rest")
LRS = (1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 2.5e-3 / 20.0)


def _scene(raw: bool, n=20_000, sh_degree=3):
    from gspl_amd import synthetic
    means, scales, quats, opac, shs = synthetic.scene(n, seed=7, sh_degree=sh_degree)
    scales = scales * 4
    if raw:      # the reference model stores log-scales, logits and unnormalised quaternions (vanilla_gaussian.py:345-358)
        scales, opac = torch.log(scales), torch.logit(opac.clamp(1e-4, 1 - 1e-4))
        quats = quats * (0.5 + torch.rand(n, 1, generator=torch.Generator().manual_seed(3)))
    ts = [means, scales, quats, opac, shs[:, :1].contiguous(), shs[:, 1:].contiguous()]
    return [torch.nn.Parameter(t.to(DEV).contiguous()) for t in ts if t.shape[1] > 0]


def _render(params, cam, raw, sh_degree=3):
    from gspl_amd import ops
    W, H = cam["width"], cam["height"]
    m, s, q, o = params[:4]
    dc = params[4]
    rest = params[5] if len(params) > 5 else None
    settings = ops.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.tensor([0.1, 0.2, 0.3], device=DEV), scale_modifier=1.0,
        viewmatrix=cam["world_to_camera"].to(DEV), projmatrix=cam["full_projection"].to(DEV), sh_degree=sh_degree, campos=cam["camera_center"].to(DEV))
    screen = torch.zeros_like(m, requires_grad=True)
    render, radii = ops.GaussianRasterizer(settings)(means3D=m, means2D=screen, opacities=o, shs=dc, shs_rest=rest, scales=s, rotations=q,
                                                     raw_parameters=raw)
    return render, screen


def _train(fuse: bool, raw: bool, steps: int, sh_degree=3, n=20_000):
    from gspl_amd import synthetic
    from gspl_amd.optimizers import FusedAdam
    params = _scene(raw, n=n, sh_degree=sh_degree)
    groups = [{"params": [p], "lr": lr, "name": nm} for p, lr, nm in zip(params, LRS, NAMES)]
    opt = FusedAdam(groups, eps=1e-15, fuse_into_backward=fuse)
    cams = synthetic.camera_set(320, 208, 300.0, count=3)
    targets = [torch.rand(3, 208, 320, generator=torch.Generator().manual_seed(10 + i)).to(DEV) for i in range(3)]
    screens, digests = [], []
    for k in range(steps):
        render, screen = _render(params, cams[k % 3], raw, sh_degree)
        (render - targets[k % 3]).abs().mean().backward()
        if fuse:
            assert all(p.grad is None for p in params), "an updated parameter must not carry a gradient"
        else:
            assert all(p.grad is not None for p in params)
        if k == steps - 1:
            screens.append(screen.grad.clone())
        opt.step()
        opt.zero_grad(set_to_none=True)
        if k % 7 == 3:                   # a scheduler steps AFTER the optimizer (gaussian_splatting.py:380-397): the next backward sees the new rate
            opt.param_groups[0]["lr"] *= 0.9
        digests.append([float(screen.grad.double().sum())] + [float(p.detach().double().sum()) for p in params])
    torch.cuda.synchronize()
    state = [(p.detach().clone(), opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone(), opt.state[p]["step"]) for p in params]
    _train.digests = digests          # (per step: sum of the screen-space gradient, then of every parameter AFTER that step's update)
    return state, screens[0]


@pytest.mark.parametrize("raw", [False, True], ids=["activated", "raw-parameters"])
def test_update_inside_the_backward_is_bit_equal_to_the_two_kernel_path(raw):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    was = ops.set_deterministic(True)
    try:
        a, sa = _train(False, raw, 40)
        da = _train.digests
        b, sb = _train(True, raw, 40)
        db = _train.digests
    finally:
        ops.set_deterministic(was)
    first = next((k for k, (x, y) in enumerate(zip(da, db)) if x != y), None)
    assert first is None, (f"the two paths part at step {first}: (screen-space gradient, " + ", ".join(NAMES) + f") sums {da[first]} vs {db[first]}")
    assert torch.equal(sa, sb), "viewspace_points.grad differs"
    for (pa, ma, va, ta), (pb, mb, vb, tb), name in zip(a, b, NAMES):
        assert ta == tb == 40, (name, ta, tb)
        assert torch.equal(ma, mb), f"{name}: exp_avg differs (max {float((ma - mb).abs().max()):.3e})"
        assert torch.equal(va, vb), f"{name}: exp_avg_sq differs"
        assert torch.equal(pa, pb), f"{name}: parameter differs (max {float((pa - pb).abs().max()):.3e})"
        assert bool(torch.isfinite(pb).all())
    # and the parameters did move
    start = _scene(raw)
    assert all(float((s - pb).abs().max()) > 0 for s, (pb, _, _, _) in zip(start, b))


def test_sh_degree_0_model_and_regular_mode():
    """A model without `shs_rest` (SH degree 0, configs of the matrixcity family) and the regular (atomic) compositing backward: equal to
    the two-kernel path within the spread of the fp32 atomics (tests/test_backward_spread.py: 2e-5 on the compositing gradients)."""
    import gspl_amd  # noqa: F401
    a, _ = _train(False, False, 12, sh_degree=0)
    b, _ = _train(True, False, 12, sh_degree=0)
    assert len(a) == 5
    # Two runs of the SAME path differ by this much too: the atomics' order perturbs a gradient in its last bits, and Adam turns a
    # gradient into a step of ~lr whatever its size — an element whose gradient is at noise level may walk lr per step either way.
    # So: every element within the farthest Adam can move it (2 lr per step), the mean difference a small fraction of one step, and the
    # first moments — linear in the gradients — equal to 1e-3 of their scale on average.
    for (pa, ma, va, _), (pb, mb, vb, _), name, lr in zip(a, b, NAMES, LRS):
        d = (pa - pb).abs()
        assert float(d.max()) <= 2 * lr * 12, (name, float(d.max()), lr)
        assert float(d.mean()) <= 0.05 * lr, (name, float(d.mean()), lr)
        assert float((ma - mb).abs().mean()) <= 1e-3 * float(ma.abs().mean()) + 1e-12, name      # (the mean: single elements follow their parameter's walk)


def test_contract_second_backward_raises_and_foreign_parameters_fall_back():
    import gspl_amd  # noqa: F401
    from gspl_amd import synthetic
    from gspl_amd.optimizers import FusedAdam
    cam = synthetic.camera(320, 208, 300.0)
    params = _scene(False, n=5000)
    opt = FusedAdam([{"params": [p], "lr": lr, "name": nm} for p, lr, nm in zip(params, LRS, NAMES)], eps=1e-15, fuse_into_backward=True)
    before = [p.detach().clone() for p in params]
    render, _ = _render(params, cam, False)
    render.mean().backward()
    assert all(p.grad is None for p in params) and any(not torch.equal(b, p.detach()) for b, p in zip(before, params))
    after_bwd = [p.detach().clone() for p in params]
    render, _ = _render(params, cam, False)
    with pytest.raises(RuntimeError, match="second backward"):
        render.mean().backward()
    opt.step()                                   # nothing left to apply: the parameters stay where the backward left them
    assert all(torch.equal(a, p.detach()) for a, p in zip(after_bwd, params))
    assert all(opt.state[p]["step"] == 1 for p in params)
    # a parameter of ANOTHER (plain) optimizer among the inputs: the backward writes every gradient, step() applies them
    params2 = _scene(False, n=5000)
    opt_a = FusedAdam([{"params": [p], "lr": lr, "name": nm} for p, lr, nm in zip(params2[:1], LRS, NAMES)], eps=1e-15)
    opt_b = FusedAdam([{"params": [p], "lr": lr, "name": nm} for p, lr, nm in zip(params2[1:], LRS[1:], NAMES[1:])], eps=1e-15, fuse_into_backward=True)
    snap = [p.detach().clone() for p in params2]
    render, _ = _render(params2, cam, False)
    render.mean().backward()
    assert all(p.grad is not None for p in params2) and all(torch.equal(s, p.detach()) for s, p in zip(snap, params2))
    opt_a.step(); opt_b.step()
    assert all(not torch.equal(s, p.detach()) for s, p in zip(snap, params2))
    # switched off: the two-kernel path again
    opt.zero_grad(set_to_none=True)
    opt.fuse_into_backward = False
    render, _ = _render(params, cam, False)
    render.mean().backward()
    assert all(p.grad is not None for p in params)
