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
