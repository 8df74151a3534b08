"""Fused densification statistics (include/gspl_hip.h §11) against the restated reference lines: counts and radii
bit-exact, the accumulated gradient norm within 1e-6 relative (fp32: sqrt of a two-term sum vs torch.norm)."""
import numpy as np
import pytest
import torch

from oracle import density_oracle as D


def _case(n, cols, seed, int_radii=True):
    g = torch.Generator().manual_seed(seed)
    grad = torch.randn(n, cols, generator=g) * 1e-3
    radii = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32)
    radii[torch.rand(n, generator=g) < 0.3] = 0
    vis = radii > 0
    vis[torch.rand(n, generator=g) < 0.1] = False           # a filter narrower than radii > 0 (the distributed renderer's)
    state = (torch.rand(n, generator=g) * 30, torch.rand(n, 1, generator=g), torch.randint(0, 5, (n, 1), generator=g).float())
    return grad, (radii if int_radii else radii.float()), vis, state


def test_oracle_matches_a_plain_loop():
    grad, radii, vis, (mr, acc, den) = _case(200, 3, 0)
    mr2, acc2, den2 = D.update_states(mr, acc, den, grad, vis, radii, scale=torch.tensor([[3.0, 5.0]]))
    for i in range(200):
        if vis[i]:
            assert mr2[i] == max(mr[i], float(radii[i]))
            assert abs(float(acc2[i, 0]) - float(acc[i, 0]) - float(np.hypot(float(grad[i, 0]) * 3.0, float(grad[i, 1]) * 5.0))) < 1e-6
            assert den2[i, 0] == den[i, 0] + 1
        else:
            assert mr2[i] == mr[i] and acc2[i, 0] == acc[i, 0] and den2[i, 0] == den[i, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("cols,scale_kind,int_radii", [(3, "none", True), (2, "tensor2", True), (2, "float", False), (3, "tensor1", True)])
@pytest.mark.parametrize("n", [1, 1000, 100_003])
def test_fused_stats_vs_oracle(n, cols, scale_kind, int_radii):
    import gspl_amd  # noqa: F401
    from gspl_amd.density import update_densification_stats
    dev = "cuda:0"
    grad, radii, vis, (mr, acc, den) = _case(n, cols, n + cols, int_radii)
    scale = {"none": None, "float": 2.5, "tensor2": torch.tensor([[960.0, 540.0]]), "tensor1": torch.tensor(7.0)}[scale_kind]
    ref = D.update_states(mr, acc, den, grad, vis, radii, scale=scale)
    d_mr, d_acc, d_den = mr.to(dev), acc.to(dev), den.to(dev)
    update_densification_stats(grad.to(dev), vis.to(dev), radii.to(dev), d_acc, d_den, d_mr,
                               scale=scale.to(dev) if isinstance(scale, torch.Tensor) else scale)
    assert torch.equal(d_mr.cpu(), ref[0])
    assert torch.equal(d_den.cpu(), ref[2])
    np.testing.assert_allclose(d_acc.cpu().numpy(), ref[1].numpy(), rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_mixin_reads_outputs_like_the_reference_controller():
    import gspl_amd  # noqa: F401
    from gspl_amd.density import HipDensityStatsMixin
    dev = "cuda:0"
    n = 5000
    grad, radii, vis, (mr, acc, den) = _case(n, 2, 9)

    class Cfg:
        absgrad = True

    class Ctl(HipDensityStatsMixin):
        config = Cfg()

    c = Ctl()
    c.max_radii2D, c.xyz_gradient_accum, c.denom = mr.to(dev), acc.to(dev), den.to(dev)
    vp = torch.zeros(n, 2, device=dev, requires_grad=True)
    vp.grad = torch.zeros(n, 2, device=dev)
    vp.absgrad = grad.abs().to(dev)
    scale = torch.tensor([[100.0, 50.0]], device=dev)
    c.update_states({"viewspace_points": vp, "visibility_filter": vis.to(dev), "radii": radii.to(dev), "viewspace_points_grad_scale": scale})
    ref = D.update_states(mr, acc, den, grad.abs(), vis, radii, scale=scale.cpu())
    assert torch.equal(c.max_radii2D.cpu(), ref[0]) and torch.equal(c.denom.cpu(), ref[2])
    np.testing.assert_allclose(c.xyz_gradient_accum.cpu().numpy(), ref[1].numpy(), rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("n_views", [1, 3, 8])
def test_stats_of_every_camera_of_a_sharded_step_in_one_launch(n_views):
    """`update_densification_stats_views` (gspl_densify_stats_views; round 6): the loop over the step's cameras of
    `DistributedVanillaDensityControllerImpl.update_states` (distributed_vanilla_density_controller.py:22-47) as one launch —
    the buffers equal those of one `update_densification_stats` call per camera, in list order, bit for bit; and the mixin reads
    `outputs` the way the reference's controller does."""
    import types
    import gspl_amd  # noqa: F401
    from gspl_amd.density import update_densification_stats, update_densification_stats_views, HipDistributedDensityStatsMixin
    dev, n = "cuda:0", 50_003
    g = torch.Generator().manual_seed(n_views)
    grads = [torch.randn(n, 2, generator=g).to(dev) for _ in range(n_views)]
    radii = [torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).to(dev) * (torch.rand(n, generator=g) > 0.3).to(dev) for _ in range(n_views)]
    masks = [r > 0 for r in radii]
    scale = 0.5 * torch.tensor([[640.0, 480.0]], device=dev)
    start = [torch.rand(n, generator=g).to(dev), torch.randint(0, 5, (n,), generator=g).float().to(dev), torch.randint(0, 30, (n,), generator=g).float().to(dev)]
    seq = [t.clone() for t in start]
    for gr, m, r in zip(grads, masks, radii):
        update_densification_stats(gr, m, r, *seq, scale=scale)
    one = [t.clone() for t in start]
    update_densification_stats_views(grads, masks, radii, *one, scale=scale)
    assert all(torch.equal(a, b) for a, b in zip(one, seq))
    assert not torch.equal(one[0], start[0])
    # the mixin, on an `outputs` dictionary as HipGSplatDistributedRenderer returns it
    class _Base:
        def update_states(self, outputs):
            raise AssertionError("the mixin must not fall through")
    class Ctrl(HipDistributedDensityStatsMixin, _Base):
        pass
    c = Ctrl()
    c.config = types.SimpleNamespace(absgrad=False)
    c.xyz_gradient_accum, c.denom, c.max_radii2D = start[0].clone().reshape(n, 1), start[1].clone().reshape(n, 1), start[2].clone()
    xys = []
    for gr in grads:
        x = torch.zeros(n, 2, device=dev, requires_grad=True)
        x.grad = gr.clone()
        xys.append(x)
    cams = [types.SimpleNamespace(width=torch.tensor(640), height=torch.tensor(480)) for _ in range(n_views)]
    c.update_states({"cameras": cams, "projection_results_list": [(r, x) for r, x in zip(radii, xys)], "visible_mask_list": masks, "xys_grad_scale_required": True})
    assert torch.equal(c.xyz_gradient_accum.reshape(-1), seq[0]) and torch.equal(c.denom.reshape(-1), seq[1]) and torch.equal(c.max_radii2D, seq[2])
