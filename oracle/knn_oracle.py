"""CPU oracle for `distCUDA2` (mean squared distance to the three nearest other points).

TEST INFRASTRUCTURE ONLY (imported by tests/ and __graft_entry__.smoke()); the product path is
gaussian-splatting-lightning_amd/csrc/knn.hip.

Restates the published behaviour of simple-knn's `distCUDA2` (yzslab/simple-knn@44f76429, un-vendored: the reference's
only in-tree trace of it is the call site internal/models/vanilla_gaussian.py:122-125).  **Parity unpinned** against
that CUDA package; pinned here by two independent formulations that must agree: an O(N^2) brute force in fp64 and
scipy's k-d tree.
"""
from __future__ import annotations

import numpy as np


def mean_dist2_brute(points: np.ndarray) -> np.ndarray:
    """O(N^2), fp64: for each point the mean of the 3 smallest squared distances to OTHER points
    (fewer than 3 others: mean over those that exist; a lone point: 0)."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    n = p.shape[0]
    out = np.zeros(n)
    for i in range(n):
        d = ((p - p[i]) ** 2).sum(1)
        d = np.delete(d, i)
        k = min(3, d.shape[0])
        if k:
            out[i] = np.sort(d)[:k].mean()
    return out


def mean_dist2_kdtree(points: np.ndarray) -> np.ndarray:
    """Same quantity through scipy.spatial.cKDTree (k = 4 including the query point itself)."""
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    n = p.shape[0]
    if n <= 1:
        return np.zeros(n)
    k = min(4, n)
    d, idx = cKDTree(p).query(p, k=k)
    d = np.atleast_2d(d)
    # drop ONE occurrence of the query point (column 0 is a zero distance: the point itself or an exact duplicate —
    # either way one zero belongs to "self")
    return (d[:, 1:] ** 2).mean(1)
