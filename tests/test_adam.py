"""Fused / visibility-masked Adam: HIP vs torch.optim.Adam and vs the selective-Adam restatement."""
import pytest
import torch

from oracle import adam_oracle as A

SHAPES = [(3,), (3,), (4,), (1,), (1, 3), (15, 3)]        # means, scales, rotations, opacities, shs_dc, shs_rest


def _model(n, seed, device):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn((n,) + s, generator=g).to(device).requires_grad_(True) for s in SHAPES]


def test_oracle_masks_rows():
    p = torch.ones(4, 3, dtype=torch.float64); g = torch.full((4, 3), 0.5, dtype=torch.float64)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    vis = torch.tensor([True, False, True, False])
    A.selective_adam_step(p, g, m, v, vis, 0.1, 0.9, 0.999, 1e-8)
    assert torch.equal(p[1], torch.ones(3, dtype=torch.float64)) and float(m[1].abs().max()) == 0.0
    # m = 0.05, v = 0.00025, update = 0.1 * 0.05 / (sqrt(0.00025) + 1e-8)
    assert abs(float(p[0, 0]) - (1 - 0.1 * 0.05 / (0.00025 ** 0.5 + 1e-8))) < 1e-12


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam():
    import gspl_amd  # noqa: F401
    from gspl_amd.optimizers import FusedAdam
    n = 5003                                   # odd sizes: exercises the non-multiple-of-4 tails
    ours, theirs = _model(n, 1, "cuda"), _model(n, 1, "cuda")
    lrs = [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 1.25e-4]
    o1 = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(ours, lrs)], eps=1e-15)
    o2 = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(theirs, lrs)], eps=1e-15)
    g = torch.Generator().manual_seed(5)
    for it in range(4):
        for a, b in zip(ours, theirs):
            grad = torch.randn(a.shape, generator=g).cuda() * (0.1 if it % 2 else 1.0)
            a.grad, b.grad = grad.clone(), grad.clone()
        o1.step(); o2.step()
    for a, b in zip(ours, theirs):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    for a, b in zip(ours, theirs):
        sa, sb = o1.state[a], o2.state[b]
        assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 1e-6
        assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 1e-6


@pytest.mark.gpu
def test_selective_adam_matches_restatement_and_leaves_hidden_rows():
    import gspl_amd  # noqa: F401
    from gspl_amd.optimizers import SelectiveAdam
    n = 4099
    params = _model(n, 2, "cuda")
    ref = [p.detach().double().cpu().clone() for p in params]
    ms = [torch.zeros_like(r) for r in ref]; vs = [torch.zeros_like(r) for r in ref]
    lrs = [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 1.25e-4]
    opt = SelectiveAdam([{"params": [p], "lr": lr} for p, lr in zip(params, lrs)], eps=1e-15, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(9)
    before = [p.detach().clone() for p in params]
    never = torch.zeros(n, dtype=torch.bool)
    never[::7] = True                          # rows that are never visible
    for it in range(3):
        vis = (torch.rand(n, generator=g) < 0.6) & ~never
        for p, r, m, v, lr in zip(params, ref, ms, vs, lrs):
            grad = torch.randn(p.shape, generator=g)
            p.grad = grad.cuda()
            A.selective_adam_step(r, grad.double(), m, v, vis, lr, 0.9, 0.999, 1e-15)
        opt.step(vis.cuda())
    for p, r, b in zip(params, ref, before):
        assert float((p.detach().cpu().double() - r).abs().max()) <= 2e-6 * max(1.0, float(r.abs().max()))
        assert torch.equal(p.detach()[never.cuda()], b[never.cuda()])          # bit-identical: never written
    for p in params:
        assert float(opt.state[p]["exp_avg"][never.cuda()].abs().max()) == 0.0


@pytest.mark.gpu
def test_selective_adam_rejects_bad_inputs():
    import gspl_amd  # noqa: F401
    from gspl_amd.optimizers import SelectiveAdam
    p = torch.zeros(8, 3, device="cuda", requires_grad=True)
    p.grad = torch.ones_like(p)
    opt = SelectiveAdam([p])
    with pytest.raises(ValueError):
        opt.step(torch.ones(7, dtype=torch.bool, device="cuda"))
    cpu = torch.zeros(8, 3, requires_grad=True)
    cpu.grad = torch.ones_like(cpu)
    with pytest.raises(RuntimeError):
        SelectiveAdam([cpu]).step(torch.ones(8, dtype=torch.bool))


@pytest.mark.gpu
def test_fused_adam_resumes_a_torch_adam_state_and_accepts_its_kwargs():
    """A reference checkpoint's optimizer state (torch.optim.Adam: `step` is a tensor) loads into HipFusedAdam's optimizer and
    the next steps match torch.optim.Adam's; Adam's keyword arguments are accepted at their defaults, refused otherwise."""
    import gspl_amd  # noqa: F401
    from gspl_amd import optimizers as gopt
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(1000, 3, generator=g)
    grads = [torch.randn(1000, 3, generator=g) for _ in range(4)]
    ref_p = p0.clone().cuda().requires_grad_(True)
    ref = torch.optim.Adam([{"params": [ref_p], "name": "means"}], lr=1e-2, eps=1e-15)
    for k in range(2):
        ref_p.grad = grads[k].cuda()
        ref.step()
    import copy
    sd = copy.deepcopy(ref.state_dict())          # what a checkpoint holds: state[0]["step"] is a tensor (a copy: load_state_dict
                                                  # keeps same-device tensors by reference, the two optimizers must not share moments)
    assert isinstance(sd["state"][0]["step"], torch.Tensor)
    my_p = ref_p.detach().clone().requires_grad_(True)
    mine = gopt.HipFusedAdam().instantiate([{"params": [my_p], "name": "means"}], lr=1e-2, eps=1e-15, weight_decay=0.0, amsgrad=False)
    mine.load_state_dict(sd)
    for k in range(2, 4):
        ref_p.grad = grads[k].cuda()
        my_p.grad = grads[k].cuda()
        ref.step()
        mine.step()
    assert float((ref_p - my_p).abs().max()) <= 2e-6
    assert mine.state[my_p]["step"] == 4 and isinstance(mine.state[my_p]["step"], int)
    with pytest.raises(NotImplementedError):
        gopt.FusedAdam([my_p], lr=1e-3, weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        gopt.FusedAdam([my_p], lr=1e-3, amsgrad=True)
    with pytest.raises(TypeError):
        gopt.FusedAdam([my_p], lr=1e-3, nonsense=1)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fused", "selective"])
def test_deferred_update_is_bit_identical_and_keeps_its_gradient_alive(kind):
    """`FusedAdam(deferred=("shs_rest",))`: the update of that group runs on the rasterizer's colour stream with a bounded grid.
    Same kernel, same inputs: after several steps on given gradients every parameter and moment equals the all-on-one-stream run
    bit for bit — also when the caller's stream allocates and overwrites memory right after `step()` (the optimizer has taken the
    gradient of the deferred parameter and keeps it alive until the caller's stream has waited for the update)."""
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    from gspl_amd.optimizers import FusedAdam, SelectiveAdam
    n = 300001
    names = ("means", "scales", "rotations", "opacities", "shs_dc", "shs_rest")
    lrs = [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 1.25e-4]

    def run(deferred):
        params = _model(n, 1, "cuda")
        cls = FusedAdam if kind == "fused" else SelectiveAdam
        opt = cls([{"params": [p], "lr": lr, "name": nm} for p, lr, nm in zip(params, lrs, names)], eps=1e-15,
                  deferred=("shs_rest",) if deferred else None)
        g = torch.Generator().manual_seed(5)
        for it in range(5):
            vis = (torch.rand(n, generator=g) < 0.7).cuda()
            for a in params:
                a.grad = (torch.randn(a.shape, generator=g) * (0.1 if it % 2 else 1.0)).cuda()
            opt.step() if kind == "fused" else opt.step(vis)
            if deferred:
                assert params[5].grad is None and params[5].data_ptr() in ops.PENDING_UPDATES
                # the caller's stream goes on allocating and writing: blocks of the size of the gradient the update is still reading
                junk = [torch.full_like(params[5], float("nan")) for _ in range(3)]
                del junk
            for a in params:
                a.grad = None
        sd = opt.state_dict()                      # joins
        assert not ops.PENDING_UPDATES and not opt._inflight
        torch.cuda.synchronize()
        return [p.detach().clone() for p in params], [opt.state[p][k].clone() for p in params for k in ("exp_avg", "exp_avg_sq")]

    base_p, base_m = run(False)
    got_p, got_m = run(True)
    for a, b in zip(base_p + base_m, got_p + got_m):
        assert torch.isfinite(b).all() and torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("consumer", ["fused", "staged", "sh_view_colors", "sh_view_colors_batched"])
def test_readers_of_a_deferred_parameter_see_the_finished_update(consumer):
    """The kernels of the package that read shs_rest — the fused Inria call (colour kernel on the colour stream itself: stream
    order), the staged rasterizer, `sh_view_colors` and its batched form (they wait for the update's event) — launched RIGHT AFTER a
    `step()` whose shs_rest update is still running on the colour stream, give the very image / colours (the forward pass is
    deterministic) that they give once the device has been synchronised."""
    import gspl_amd  # noqa: F401
    from gspl_amd import ops, synthetic
    from gspl_amd.optimizers import FusedAdam
    from oracle import gsplat_oracle as O
    dev = torch.device("cuda:0")
    W, H, n = 320, 208, 400000
    means, scales, quats, opac, shs = O.synthetic_scene(n, seed=3)
    cam = synthetic.camera_set(W, H, 300.0, count=2)[1]
    params = [t.clone().contiguous().to(dev).requires_grad_(True) for t in (means, scales, quats, opac, shs[:, :1], shs[:, 1:])]
    m, s, q, o, dc, rest = params
    names = ("means", "scales", "rotations", "opacities", "shs_dc", "shs_rest")
    opt = FusedAdam([{"params": [p], "lr": 1e-2, "name": nm} for p, nm in zip(params, names)], eps=1e-15, deferred=("shs_rest",))
    settings = ops.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev), scale_modifier=1.0,
        viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=3, campos=cam["camera_center"].to(dev))
    center = cam["camera_center"].to(dev)

    def read():
        with torch.no_grad():
            if consumer == "sh_view_colors":
                return ops.sh_view_colors(3, m, center, dc, rest, None)
            if consumer == "sh_view_colors_batched":
                return ops.sh_view_colors_batched(3, m, torch.stack([center, center + 0.1]), dc, rest, None)
            return ops.GaussianRasterizer(settings)(means3D=m, means2D=torch.empty_like(m), opacities=o, shs=dc, shs_rest=rest, scales=s, rotations=q)[0]

    fused = ops.FUSED_INRIA
    ops.FUSED_INRIA = consumer != "staged"
    try:
        g = torch.Generator().manual_seed(2)
        before = read()
        for it in range(4):
            for p in params:
                p.grad = torch.randn(p.shape, generator=g).to(dev)
            torch.cuda.synchronize()
            opt.step()
            first = read()                         # the update of shs_rest is in flight on the colour stream
            torch.cuda.synchronize()
            assert rest.data_ptr() in ops.PENDING_UPDATES
            settled = read()
            assert torch.equal(first, settled), f"step {it}: a reader saw shs_rest before its update had finished"
            assert not torch.equal(settled, before)
            before = settled
    finally:
        ops.FUSED_INRIA = fused
        opt.join()


@pytest.mark.gpu
@pytest.mark.parametrize("reader", ["v1_merged_features", "v0_pre_activated"])
def test_torch_reads_of_a_deferred_parameter_wait_for_its_update(reader):
    """`_await_updates` ties an in-flight update to a parameter by its data pointer, which a TORCH read cannot
    carry — `get_features` is a `torch.cat` on the caller's stream, a dtype / layout copy makes a new tensor.  Those paths join every
    update in flight first (`ops.join_pending_updates`): HipGSplatV1Renderer with separate_sh=False and a renderer handed a
    pre-activated model (one merged `get_features`) must give the image of the settled state."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gspl_amd  # noqa: F401
    from fakes import FakeCamera, FakeGaussianModel
    from gspl_amd import ops, synthetic
    from gspl_amd.optimizers import FusedAdam
    from gspl_amd.renderers import HipGSplatRenderer, HipGSplatV1Renderer
    from oracle import gsplat_oracle as O
    dev = torch.device("cuda:0")
    W, H, n = 320, 208, 400000
    params = O.synthetic_scene(n, seed=3)
    cam = FakeCamera(O.synthetic_camera(W, H, 300.0), dev)
    model = FakeGaussianModel(*[p.to(dev) for p in params])
    names = ("means", "scales", "rotations", "opacities", "shs_dc", "shs_rest")
    opt = FusedAdam([{"params": [p], "lr": 1e-2, "name": nm} for p, nm in zip(model.leaves(), names)], eps=1e-15, deferred=("shs_rest",))
    bg = torch.zeros(3, device=dev)
    if reader == "v1_merged_features":
        renderer = HipGSplatV1Renderer(separate_sh=False).instantiate()
    elif reader == "v0_pre_activated":
        renderer = HipGSplatRenderer()
        model.is_pre_activated = True          # `model_sh_pair` then reads `get_features` (one merged tensor)

    def read():
        with torch.no_grad():
            return renderer(cam, model, bg)["render"]

    g = torch.Generator().manual_seed(2)
    before = read()
    try:
        for it in range(3):
            for p in model.leaves():
                p.grad = torch.randn(p.shape, generator=g).to(dev)
            torch.cuda.synchronize()
            opt.step()
            first = read()                         # the update of shs_rest is in flight on the colour stream
            torch.cuda.synchronize()
            settled = read()
            assert torch.equal(first, settled), f"step {it}: a torch read saw shs_rest before its update had finished"
            assert not torch.equal(settled, before)
            before = settled
    finally:
        opt.join()


def test_a_stray_gradient_on_a_parameter_the_backward_has_updated_raises():
    """with `fuse_into_backward` the rasterizer's backward updates a claimed parameter in place.  A second autograd
    path into the same parameter (a regulariser, an extra loss) leaves a gradient in `.grad`; `step()` used to apply it as a SECOND
    Adam update with a second step count.  Now it raises, names the parameter, and the optimizer is usable again afterwards (host
    logic only: the claim is made by hand, nothing is launched)."""
    import gspl_amd  # noqa: F401
    from gspl_amd.optimizers import FusedAdam
    p = torch.nn.Parameter(torch.zeros(8, 3))
    q = torch.nn.Parameter(torch.zeros(8, 1))
    opt = FusedAdam([{"params": [p], "name": "scales", "lr": 1e-3}, {"params": [q], "name": "opacities", "lr": 1e-3}], fuse_into_backward=True)
    opt._claimed.update({id(p), id(q)})          # what the rasterizer's backward leaves behind (optimizers.claim_backward_update)
    p.grad = torch.ones_like(p)                  # ... and what a regulariser on the scales adds
    with pytest.raises(RuntimeError, match="second autograd path") as e:
        opt.step()
    assert "scales" in str(e.value) and "opacities" not in str(e.value)
    assert not opt._claimed and opt.state[p].get("step", 0) in (0, None)      # nothing was applied, the claim is gone
    p.grad = None
    opt.step()                                   # nothing to do, nothing raised
