"""No sanitizer exists for the HIP side on this pool: the blocks the fused Inria call asks its allocation call-back for (geometry, image state,
lists, sort workspaces, packed gradient rows, checkpoints — the library carves every per-frame array out of them) are handed out here with a
poisoned band on either side, and the bands must be intact after forward + backward: a write past either end of any block fails the test."""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 4096
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def _guarded_allocator(monkeypatch):
    from gspl_amd import _lib as L
    from gspl_amd.ops import inria
    outers = []

    def guarded(_ctx, tag, nbytes):
        holder = inria._ALLOC_TLS.holder
        try:
            n = max(int(nbytes), 1)
            n_up = (n + 255) // 256 * 256            # (the inner block stays 256-byte aligned, as torch's own blocks are)
            outer = torch.empty((n_up + 2 * GUARD,), dtype=torch.uint8, device=holder["device"])
            outer[:GUARD] = 0xA5
            outer[GUARD + n:] = 0xA5
            inner = outer[GUARD:GUARD + n]
            holder.setdefault(tag, []).append(inner)
            outers.append((tag, n, outer))
            return inner.data_ptr()
        except Exception as e:      # noqa: BLE001
            holder["error"] = e
            return 0

    monkeypatch.setattr(inria, "_ALLOC_CB", L.ALLOC_FN(guarded))
    return outers


def _check(outers, what):
    torch.cuda.synchronize()
    assert outers, "the fused call did not allocate through the call-back"
    for tag, n, outer in outers:
        assert bool((outer[:GUARD] == 0xA5).all()), f"{what}: a write BELOW block {tag} ({n} bytes)"
        assert bool((outer[GUARD + n:] == 0xA5).all()), f"{what}: a write ABOVE block {tag} ({n} bytes)"
    return len(outers)


@pytest.mark.parametrize("segmented", [True, "always"])
def test_no_write_outside_the_blocks_of_the_fused_call(monkeypatch, segmented):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import fuzz_parity as FP
    from gspl_amd.ops._state import STATE as S
    outers = _guarded_allocator(monkeypatch)
    monkeypatch.setattr(S, "segmented_backward", segmented)
    hip, dev = FP.hip, FP.dev
    blocks = 0
    for seed in list(range(7000, 7040)) + [1126, 2312]:
        desc, (means, scales, quats, opac, shs, cam, wimg, bg) = FP.random_case(seed)
        W, H = cam["width"], cam["height"]
        deg = int(math.isqrt(shs.shape[1])) - 1
        for frame in range(2):      # the second frame runs on the first one's capacity hints (speculative emission)
            del outers[:]
            leaves = [t.requires_grad_(True) for t in FP.cuda(means, scales, quats, opac, shs)]
            m, s, q, o, c = leaves
            st = hip.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(dev), scale_modifier=1.0,
                                                   viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=deg,
                                                   campos=cam["camera_center"].to(dev))
            img, radii = hip.GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, shs=c, scales=s, rotations=q)
            (img * wimg.to(dev)).sum().backward()
            blocks += _check(outers, f"seed {seed} {desc} frame {frame}")
    assert blocks > 400


@pytest.mark.parametrize("workload,segmented", [("S-1080p-1M", True), ("S-1080p-1M-surfaces", "always"), ("S-1080p-1M-inside", True)])
def test_no_write_outside_the_blocks_at_the_metric_size(monkeypatch, workload, segmented):
    """The same at 1 M Gaussians and 1920 x 1080 (three frames of three views: a cold frame, a hit, and a view whose list is longer)."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gspl_amd  # noqa: F401
    from gspl_amd import ops, synthetic
    from gspl_amd.ops._state import STATE as S
    outers = _guarded_allocator(monkeypatch)
    monkeypatch.setattr(S, "segmented_backward", segmented)
    wl = synthetic.WORKLOADS[workload]
    W, H = wl["width"], wl["height"]
    dev = torch.device("cuda:0")
    params = [t.to(dev) for t in synthetic.workload_scene(wl, seed=42)]
    cams = synthetic.camera_set(W, H, wl["fx"], count=16, distance=wl.get("distance", 4.0))
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).to(dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    for k in (0, 0, 5):
        cam = cams[k]
        del outers[:]
        leaves = [t.detach().clone().requires_grad_(True) for t in params]
        m, s, q, o, c = leaves
        st = ops.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
                                               viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=3,
                                               campos=cam["camera_center"].to(dev))
        img, radii = ops.GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, shs=c, scales=s, rotations=q)
        (img * wimg).sum().backward()
        assert _check(outers, f"{workload} view {k}") >= 4
        assert all(bool(torch.isfinite(t.grad).all()) for t in leaves)
