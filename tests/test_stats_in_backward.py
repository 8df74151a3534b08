"""The density controller's statistics applied by the frame's backward (`density.request_stats_in_backward`,
`gspl_inria_state.stats_*`, ABI 34; VERDICT r4 #2 "densify_stats folded into inria_preprocess_bwd").

What it replaces is `update_densification_stats(viewspace_points.grad, None, radii, ...)` after the backward — the fused form of
`VanillaDensityControllerImpl.update_states` (internal/density_controllers/vanilla_density_controller.py:101-123), which
tests/test_density.py pins to the oracle.  The preprocess-backward kernel holds the row's screen-space gradient in registers and
applies the same three lines with the same arithmetic, so the check is BIT-equality against that kernel run on the gradient the same
backward returned, from identical buffers:
  * plain backward, raw-parameter backward, and the backward that also applies Adam;
  * once per request (a second backward through the same render is refused), never for another frame's radii, never when switched off;
  * the mixin: `before_backward` hands the buffers over, `update_states` launches nothing — and falls back to its kernel whenever
    the request was not (or could not be) taken: a renderer without the mark, a gradient scale, absgrad, past `densify_until_iter`.
"""
import pytest
import torch

from test_fused_backward_adam import DEV, LRS, NAMES, _scene

CAM = dict(width=320, height=208, fx=300.0)


def _cam(i=0):
    from gspl_amd import synthetic
    return synthetic.camera_set(CAM["width"], CAM["height"], CAM["fx"], count=3)[i]


def _buffers(n, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(n, generator=g) * s).to(DEV) for s in (0.01, 5.0, 30.0)]      # accum, denom, max_radii: not zero


def _frame(params, raw, cam=None):
    """One render; returns (loss, screen, radii)."""
    from gspl_amd import ops
    cam = cam or _cam()
    W, H = cam["width"], cam["height"]
    m, s, q, o, dc, rest = params
    settings = ops.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.tensor([0.1, 0.2, 0.3], device=DEV), scale_modifier=1.0,
        viewmatrix=cam["world_to_camera"].to(DEV), projmatrix=cam["full_projection"].to(DEV), sh_degree=3, campos=cam["camera_center"].to(DEV))
    screen = torch.zeros_like(m, requires_grad=True)
    render, radii = ops.GaussianRasterizer(settings)(means3D=m, means2D=screen, opacities=o, shs=dc, shs_rest=rest, scales=s, rotations=q,
                                                     raw_parameters=raw)
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(4)).to(DEV)
    return (render - target).abs().mean(), screen, radii


def _expected(screen_grad, radii, before, vis=None):
    from gspl_amd.density import update_densification_stats
    accum, denom, max_radii = [t.clone() for t in before]
    update_densification_stats(screen_grad, vis, radii, accum, denom, max_radii, scale=None)
    return accum, denom, max_radii


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["activated", "raw-parameters", "with-adam-inside"])
def test_statistics_applied_by_the_backward_equal_the_kernel_after_it(kind):
    import gspl_amd  # noqa: F401
    from gspl_amd.density import request_stats_in_backward
    from gspl_amd.optimizers import FusedAdam
    raw = kind == "raw-parameters"
    params = _scene(raw)
    n = params[0].shape[0]
    opt = None
    if kind == "with-adam-inside":
        opt = FusedAdam([{"params": [p], "lr": lr, "name": nm} for p, lr, nm in zip(params, LRS, NAMES)], eps=1e-15, fuse_into_backward=True)
    bufs = _buffers(n)
    before = [t.clone() for t in bufs]
    loss, screen, radii = _frame(params, raw)
    req = request_stats_in_backward(radii, *bufs)
    assert req is not None and not req.applied
    loss.backward()
    assert req.applied
    if opt is not None:
        assert all(p.grad is None for p in params), "the optimizer's update must have run inside this backward too"
    exp = _expected(screen.grad, radii, before)
    vis = radii > 0
    assert 0 < int(vis.sum()) < n
    for got, want, name in zip(bufs, exp, ("xyz_gradient_accum", "denom", "max_radii2D")):
        assert torch.equal(got, want), name
        assert torch.equal(got[~vis], before[("xyz_gradient_accum", "denom", "max_radii2D").index(name)][~vis]), name + ": an invisible row moved"
    assert float((bufs[0] - before[0]).abs().max()) > 0


@pytest.mark.gpu
def test_a_request_is_taken_once_and_only_by_its_own_frame():
    import gspl_amd  # noqa: F401
    from gspl_amd.density import request_stats_in_backward, withdraw_stats_request
    from gspl_amd.ops._state import STATE
    params = _scene(False)
    n = params[0].shape[0]
    bufs = _buffers(n)
    before = [t.clone() for t in bufs]
    # another frame's backward leaves the request alone
    loss_a, _, radii_a = _frame(params, False, _cam(0))
    loss_b, screen_b, radii_b = _frame(params, False, _cam(1))
    req = request_stats_in_backward(radii_b, *bufs)
    loss_a.backward()
    assert not req.applied and STATE.backward_stats is req
    assert all(torch.equal(a, b) for a, b in zip(bufs, before))
    # its own takes it, once: a second backward through the RETAINED graph runs (round 6) and does not add the statistics again; the
    # graph is released by it, and a third one gets autograd's own refusal
    loss_b.backward(retain_graph=True)
    assert req.applied and STATE.backward_stats is None
    once = [t.clone() for t in bufs]
    exp = _expected(screen_b.grad, radii_b, before)
    assert all(torch.equal(a, b) for a, b in zip(once, exp))
    first = screen_b.grad.clone()
    loss_b.backward()
    assert all(torch.equal(a, b) for a, b in zip(bufs, once))
    assert torch.allclose(screen_b.grad, 2 * first, rtol=1e-4, atol=1e-7)      # (.grad accumulates: the same gradient twice, up to the atomics' order)
    with pytest.raises(RuntimeError):
        loss_b.backward()
    assert all(torch.equal(a, b) for a, b in zip(bufs, once))
    # a withdrawn request is not applied
    loss_c, _, radii_c = _frame(params, False, _cam(2))
    req_c = request_stats_in_backward(radii_c, *bufs)
    withdraw_stats_request(req_c)
    loss_c.backward()
    assert not req_c.applied and all(torch.equal(a, b) for a, b in zip(bufs, once))


@pytest.mark.gpu
def test_requests_are_refused_where_the_backward_cannot_serve_them(monkeypatch):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    from gspl_amd.density import request_stats_in_backward
    from gspl_amd.ops._state import STATE
    params = _scene(False, n=2000)
    n = params[0].shape[0]
    bufs = _buffers(n)
    _, _, radii = _frame(params, False)
    assert request_stats_in_backward(radii.clone(), *bufs) is None                       # not the tensor the fused call returned
    assert request_stats_in_backward(radii, bufs[0], bufs[1], bufs[2][: n - 1]) is None      # a buffer of another size
    assert request_stats_in_backward(radii, bufs[0].double(), bufs[1], bufs[2]) is None
    assert request_stats_in_backward(radii, bufs[0], bufs[1], None) is not None           # max_radii2D is optional
    STATE.backward_stats = None
    monkeypatch.setattr(STATE, "stats_in_backward", False)
    assert request_stats_in_backward(radii, *bufs) is None
    monkeypatch.setattr(STATE, "stats_in_backward", True)
    monkeypatch.setattr(ops, "FUSED_INRIA", False)                                        # the stage-by-stage calls have no such kernel
    _, _, radii_staged = _frame(params, False)
    assert request_stats_in_backward(radii_staged, *bufs) is None
    assert STATE.backward_stats is None


class _Cfg:
    absgrad = False
    densify_until_iter = 100


class _Base:
    """Stands for the reference's controller: `before_backward` keeps the gradient of the screen-space points (:69-76)."""
    def before_backward(self, outputs, batch, gaussian_model, optimizers, global_step, pl_module):
        if global_step < self.config.densify_until_iter:
            outputs["viewspace_points"].retain_grad()


def _controller(n):
    from gspl_amd.density import HipDensityStatsMixin

    class Ctl(HipDensityStatsMixin, _Base):
        config = _Cfg()

    c = Ctl()
    c.xyz_gradient_accum, c.denom, c.max_radii2D = [t.reshape(s) for t, s in zip(_buffers(n), ((n, 1), (n, 1), (n,)))]
    return c


def _count_stat_launches(monkeypatch):
    from gspl_amd import _lib
    calls = []
    real = _lib.call

    def counting(name, *a, **k):
        if name == "gspl_densify_stats":
            calls.append(name)
        return real(name, *a, **k)
    monkeypatch.setattr(_lib, "call", counting)
    import gspl_amd.density as density
    monkeypatch.setattr(density.L, "call", counting, raising=False)
    return calls


@pytest.mark.gpu
def test_mixin_hands_the_buffers_to_the_backward_and_falls_back_otherwise(monkeypatch):
    import gspl_amd  # noqa: F401
    from gspl_amd import synthetic
    from gspl_amd.renderers import HipVanillaRenderer
    launches = _count_stat_launches(monkeypatch)
    means, scales, quats, opac, shs = synthetic.scene(20_000, seed=7)
    model = synthetic.ModelObject(*[t.to(DEV).contiguous().requires_grad_(True) for t in (means, scales * 4, quats, opac, shs)])
    cams = synthetic.camera_set(CAM["width"], CAM["height"], CAM["fx"], count=2)
    cam = synthetic.CameraObject(cams[0], DEV, idx=0)
    renderer = HipVanillaRenderer()
    bg = torch.zeros(3, device=DEV)
    n = means.shape[0]

    def frame(ctl, step, mutate=None):
        """One frame through the hooks; returns how many statistics launches `update_states` needed (0: the backward had done it)."""
        out = renderer(cam, model, bg)
        if mutate:
            mutate(out)
        ctl.before_backward(out, None, model, [], step, None)
        before = [ctl.xyz_gradient_accum.clone(), ctl.denom.clone(), ctl.max_radii2D.clone()]
        out["render"].mean().backward()
        mid = [ctl.xyz_gradient_accum.clone(), ctl.denom.clone(), ctl.max_radii2D.clone()]
        k0 = len(launches)
        with torch.no_grad():
            ctl.update_states(out)
        launched = len(launches) - k0
        after = [ctl.xyz_gradient_accum, ctl.denom, ctl.max_radii2D]
        exp = _expected(out["viewspace_points"].grad, out["radii"], [b.reshape(-1) for b in before], out["visibility_filter"])
        for a, e in zip(after, exp):
            assert torch.equal(a.reshape(-1), e)
        assert float((after[0] - before[0]).abs().max()) > 0
        assert all(torch.equal(m, a) for m, a in zip(mid, after)) == (launched == 0)      # whoever did not launch had nothing left to do
        return launched

    from gspl_amd.ops._state import STATE
    ctl = _controller(n)
    assert frame(ctl, 5) == 0 and frame(ctl, 6) == 0                 # the backward applied them, update_states launched nothing
    # fall-backs: each of these frames is served by the kernel after the backward, with the same result
    def no_mark(out):
        out["visibility_filter"] = out["radii"] > 0
    assert frame(ctl, 7, no_mark) == 1
    def narrower(out):
        out["visibility_filter"] = (out["radii"] > 0) & (torch.arange(n, device=DEV) % 3 != 0)      # another renderer's filter: unmarked
    assert frame(ctl, 8, narrower) == 1
    def scaled(out):
        out["viewspace_points_grad_scale"] = torch.ones(2, device=DEV)      # a renderer that scales the gradient (x 1 here: same sums)
    assert frame(ctl, 9, scaled) == 1
    monkeypatch.setattr(STATE, "stats_in_backward", False)
    assert frame(ctl, 9) == 1
    monkeypatch.setattr(STATE, "stats_in_backward", True)
    assert frame(ctl, 10) == 0
    # past densify_until_iter the reference updates nothing: neither does the backward
    out = renderer(cam, model, bg)
    ctl.before_backward(out, None, model, [], 100, None)
    assert ctl._stats_request is None and STATE.backward_stats is None
    keep = ctl.xyz_gradient_accum.clone()
    out["render"].mean().backward()
    assert torch.equal(ctl.xyz_gradient_accum, keep)


def test_requests_need_the_gpu():
    import gspl_amd  # noqa: F401
    from gspl_amd.density import request_stats_in_backward, withdraw_stats_request
    n = 8
    radii = torch.ones(n, dtype=torch.int32)
    radii._gspl_fused_inria = True
    assert request_stats_in_backward(radii, torch.zeros(n), torch.zeros(n), torch.zeros(n)) is None
    withdraw_stats_request(None)
