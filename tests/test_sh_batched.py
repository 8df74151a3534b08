"""`gspl_sh_fwd_batched` / `gspl_sh_bwd_batched` (C cameras per launch, coefficients read once) against the one-camera
entry points they batch and against the fp64 oracle (oracle pinned by the reference's sh_utils.py through
tests/golden/ref_sh.npz)."""
import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as O
from hip_helpers import assert_close_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("degree,n_coeffs", [(0, 1), (1, 4), (3, 16), (2, 16), (4, 25)])
@pytest.mark.parametrize("merged", [False, True])
def test_batched_sh_matches_single_camera_ops_and_oracle(degree, n_coeffs, merged):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    g = torch.Generator().manual_seed(11 + degree)
    N, C = 1037, 3
    means = torch.randn(N, 3, generator=g)
    centers = torch.randn(C, 3, generator=g) * 3
    coeffs = torch.randn(N, n_coeffs, 3, generator=g) * 0.4
    radii = torch.randint(-1, 4, (C, N), generator=g, dtype=torch.int32)
    w = torch.randn(C, N, 3, generator=g)

    def leaves():
        if merged:
            return coeffs.clone().to(DEV).requires_grad_(True), None
        return coeffs[:, :1].contiguous().to(DEV).requires_grad_(True), coeffs[:, 1:].contiguous().to(DEV).requires_grad_(True)

    dc, rest = leaves()
    out = ops.sh_view_colors_batched(degree, means.to(DEV), centers.to(DEV), dc, rest, radii.to(DEV))
    assert out.shape == (C, N, 3)
    (out * w.to(DEV)).sum().backward()

    dc1, rest1 = leaves()
    singles = [ops.sh_view_colors(degree, means.to(DEV), centers[c].to(DEV), dc1, rest1, radii[c].to(DEV) > 0) for c in range(C)]
    sum((s * w[c].to(DEV)).sum() for c, s in enumerate(singles)).backward()
    for c in range(C):
        assert torch.equal(out[c], singles[c]), f"camera {c}"
    assert_close_scaled(dc.grad.cpu().numpy(), dc1.grad.cpu().numpy(), 2e-6, "dc grad vs single-camera ops")
    if rest is not None and rest.shape[1] > 0:
        assert_close_scaled(rest.grad.cpu().numpy(), rest1.grad.cpu().numpy(), 2e-6, "rest grad vs single-camera ops")

    co = coeffs.double().requires_grad_(True)
    ref = torch.stack([torch.where((radii[c] > 0)[:, None], O.sh_colors(degree, co, means.double(), centers[c].double(), detach_dirs=True),
                                   torch.zeros((), dtype=torch.float64)) for c in range(C)])
    (ref * w.double()).sum().backward()
    assert np.abs(out.detach().cpu().numpy() - ref.detach().numpy()).max() <= 2e-5
    got = dc.grad if merged else torch.cat([dc.grad, rest.grad], dim=1)
    assert_close_scaled(got.cpu().numpy(), co.grad.numpy(), 1e-4, "coefficient grad vs oracle")
