"""usage (on the GPU box): python tools/fuzz_differential.py <minutes> [first seed]
Differential campaign over the package's OWN code paths on the random cases of tools/fuzz_parity.py — no oracle in the loop, so thousands
of cases per minute.  For every case the Inria rasterizer is run
    fused (default)  |  stage by stage  |  without the speculative emission  |  with the segmented backward forced on every frame  |
    a second time on warm speculation state (the previous case's capacity hints: too small, too large)
and the gsplat-v0 ops with tile sizes 8, 16 and 32.  Asserted: images and radii BIT-EQUAL across the Inria variants that take no checkpoints (the blending
order of a pixel is the depth order however the lists were built), within 4e-6 for the checkpointed ones; gradients finite and equal within the
reordering of fp32 atomic sums (see `same`); finite outputs for the three tile sizes."""
import os, sys, time, math, traceback
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import numpy as np
import torch
import fuzz_parity as F
from gspl_amd.ops._state import STATE as S
hip, dev = F.hip, F.dev


def inria(case, **state):
    means, scales, quats, opac, shs, cam, wimg, bg = case
    W, H = cam["width"], cam["height"]
    deg = int(math.isqrt(shs.shape[1])) - 1
    old = {k: getattr(S, k) for k in state}
    for k, v in state.items():
        setattr(S, k, v)
    try:
        leaves = [t.requires_grad_(True) for t in F.cuda(means, scales, quats, opac, shs)]
        m, s, q, o, c = leaves
        st = hip.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(dev), scale_modifier=1.0,
                                               viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=deg, campos=cam["camera_center"].to(dev))
        screen = torch.zeros_like(m, requires_grad=True)
        img, radii = hip.GaussianRasterizer(st)(means3D=m, means2D=screen, opacities=o, shs=c, scales=s, rotations=q)
        (img * wimg.to(dev)).sum().backward()
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
        return img.detach(), radii, [z(t) for t in leaves] + [z(screen)]
    finally:
        for k, v in old.items():
            setattr(S, k, v)


def gsplat(case, tile):
    means, scales, quats, opac, shs, cam, wimg, bg = case
    W, H = cam["width"], cam["height"]
    deg = int(math.isqrt(shs.shape[1])) - 1
    leaves = [t.requires_grad_(True) for t in F.cuda(means, scales, quats, opac, shs)]
    m, s, q, o, c = leaves
    vm = cam["world_to_camera"].T.contiguous().float().to(dev)
    xys, depths, radii, conics, comp, tiles, _ = hip.project_gaussians(m, s, 1.0, q, vm[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, tile)
    rgbs = hip.sh_view_colors(deg, m, cam["camera_center"].to(dev), c, None, radii > 0)
    img = hip.rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, o * comp[:, None], H, W, tile, bg.to(dev)).permute(2, 0, 1)
    (img * wimg.to(dev)).sum().backward()
    z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
    return img.detach(), radii, [z(t) for t in leaves]


NAMES = ("means", "scales", "quats", "opacities", "shs", "screen")


def same(a, b, what, img_tol=0.0, rel=5e-5, frac=0.999, cap=2e-2):
    """images within img_tol (0: bit-equal), radii equal, gradients finite and — atomic fp32 sums are added in another order from run to
    run, and the covariance chain of a needle amplifies that — `frac` of the elements of each gradient within rel (|ref| + rms), all
    within `cap`."""
    if img_tol == 0.0:
        assert torch.equal(a[0], b[0]), f"{what}: images differ (max {float((a[0] - b[0]).abs().max()):.3e})"
    else:
        assert float((a[0] - b[0]).abs().max()) <= img_tol, f"{what}: images differ by {float((a[0] - b[0]).abs().max()):.3e} (> {img_tol:g})"
    assert torch.equal(a[1], b[1]), f"{what}: radii differ"
    for name, x, y in zip(NAMES, a[2], b[2]):
        assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(y).all()), f"{what}: non-finite {name} gradient"
        if x.numel() == 0:
            continue
        rms = float(y.double().pow(2).mean().sqrt()) + 1e-30
        ratio = (x - y).abs().double() / (y.abs().double() + rms)
        ok = float((ratio <= rel).double().mean())
        assert ok >= min(frac, 1.0 - 3.0 / ratio.numel()) and float(ratio.max()) <= cap, \
            f"{what}: {name} gradient differs: {ok:.5f} of the elements within {rel:g}, worst ratio {float(ratio.max()):.3e}"


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    t0, done, failed = time.time(), 0, []
    while time.time() - t0 < minutes * 60:
        desc, case = F.random_case(seed)
        try:
            plain = dict(segmented_backward=False)
            base = inria(case, **plain)              # speculation state left by the previous case: hints of another scene
            same(inria(case, **plain), base, "warm second frame")
            same(inria(case, fused_inria=False, **plain), base, "stage by stage")
            same(inria(case, speculative_emit=False, **plain), base, "no speculative emission")
            same(inria(case, fused_inria=False, device_side_list_length=False, **plain), base, "stage by stage, host-side list length")
            # checkpointed frames sum the colour per segment and the backward re-associates T: ulps on the image
            same(inria(case, segmented_backward="always"), base, "segmented backward", img_tol=4e-6, rel=2e-4)
            same(inria(case), base, "adaptive", img_tol=4e-6, rel=2e-4)
            for tile in (8, 16, 32):                 # (the IMAGE depends on the tile size in this API: the 3-sigma rectangle is cut at tile
                g = gsplat(case, tile)               # granularity and an opaque splat is still above 1/255 beyond it, as in gsplat v0)
                assert bool(torch.isfinite(g[0]).all()) and all(bool(torch.isfinite(t).all()) for t in g[2]), f"tile {tile}: non-finite output"
            print(f"seed {seed} {desc} ok", flush=True)
        except Exception as e:      # noqa: BLE001
            failed.append(seed)
            msg = str(e).strip().splitlines()
            print(f"seed {seed} {desc} FAILED: {type(e).__name__}: {msg[0] if msg else ''}", flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
        done += 1
        seed += 1
    print(f"{done} cases, {len(failed)} failed: {failed}")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
