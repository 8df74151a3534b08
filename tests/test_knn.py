"""simple_knn stand-in: oracle self-consistency on CPU, HIP vs oracle on the GPU."""
import sys

import numpy as np
import pytest
import torch

from oracle import knn_oracle as K


def _clouds():
    rng = np.random.default_rng(7)
    uniform = rng.uniform(-1.3, 1.3, size=(4000, 3)).astype(np.float32)                    # Blender init box
    clustered = np.concatenate([rng.normal(c, 0.02, size=(700, 3)) for c in rng.uniform(-2, 2, size=(5, 3))]).astype(np.float32)
    plane = np.concatenate([rng.uniform(-1, 1, size=(3000, 2)), np.zeros((3000, 1))], 1).astype(np.float32)
    line = np.stack([np.linspace(0, 5, 1500), np.zeros(1500), np.ones(1500)], 1).astype(np.float32)
    dup = np.repeat(rng.uniform(-1, 1, size=(500, 3)), 4, axis=0).astype(np.float32)      # every point four times
    outlier = np.concatenate([rng.uniform(-1, 1, size=(2000, 3)), [[500.0, -300.0, 80.0]]]).astype(np.float32)
    return {"uniform": uniform, "clustered": clustered, "plane": plane, "line": line, "duplicates": dup, "outlier": outlier}


def test_oracle_formulations_agree():
    for name, pts in _clouds().items():
        sub = pts[:600]
        a, b = K.mean_dist2_brute(sub), K.mean_dist2_kdtree(sub)
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12, err_msg=name)


def test_oracle_small_n():
    assert K.mean_dist2_brute(np.zeros((0, 3))).shape == (0,)
    assert K.mean_dist2_brute(np.ones((1, 3)))[0] == 0.0
    two = np.array([[0, 0, 0], [3, 4, 0]], dtype=np.float64)
    np.testing.assert_allclose(K.mean_dist2_brute(two), [25.0, 25.0])
    np.testing.assert_allclose(K.mean_dist2_kdtree(two), [25.0, 25.0])
    three = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], dtype=np.float64)
    np.testing.assert_allclose(K.mean_dist2_brute(three), [(1 + 4) / 2, (1 + 5) / 2, (4 + 5) / 2])


def test_shim_registers_simple_knn():
    import gspl_amd  # noqa: F401
    from gspl_amd import compat
    had = "simple_knn" in sys.modules
    compat.install()
    from simple_knn._C import distCUDA2
    from gspl_amd import ops
    if not had:
        assert distCUDA2 is ops.distCUDA2
    with pytest.raises(RuntimeError):
        ops.distCUDA2(torch.zeros(4, 3))          # CPU tensor: no silent fallback


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["uniform", "clustered", "plane", "line", "duplicates", "outlier"])
def test_hip_matches_oracle(name):
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    pts = _clouds()[name]
    got = ops.distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy().astype(np.float64)
    want = K.mean_dist2_kdtree(pts)
    # fp32 distances: relative 1e-5 of the value plus the rounding of coordinates of magnitude |p| (d^2 ~ 2 |p| eps |d|)
    scale = np.abs(pts).max()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6 * scale * scale)


@pytest.mark.gpu
def test_hip_small_and_large():
    import gspl_amd  # noqa: F401
    from gspl_amd import ops
    assert ops.distCUDA2(torch.zeros(0, 3).cuda()).shape == (0,)
    assert float(ops.distCUDA2(torch.ones(1, 3).cuda())[0]) == 0.0
    two = torch.tensor([[0.0, 0, 0], [3, 4, 0]]).cuda()
    np.testing.assert_allclose(ops.distCUDA2(two).cpu().numpy(), [25.0, 25.0], rtol=1e-6)
    same = torch.ones(100, 3).cuda()                  # all coincident
    assert float(ops.distCUDA2(same).abs().max()) == 0.0
    # 1 M points (the metric workload's cloud): property checks + a sampled comparison with the k-d tree
    g = torch.Generator().manual_seed(42)
    big = (torch.rand(1_000_000, 3, generator=g) * 2 - 1) * 1.3
    d = ops.distCUDA2(big.cuda()).cpu().numpy()
    assert np.isfinite(d).all() and (d > 0).all()
    from scipy.spatial import cKDTree
    idx = np.random.default_rng(0).choice(big.shape[0], 2000, replace=False)
    dist, _ = cKDTree(big.numpy().astype(np.float64)).query(big.numpy()[idx].astype(np.float64), k=4)
    np.testing.assert_allclose(d[idx], (dist[:, 1:] ** 2).mean(1), rtol=2e-5, atol=1e-9)
