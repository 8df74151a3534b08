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
