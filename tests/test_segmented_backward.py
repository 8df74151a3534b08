"""The segmented compositing backward of the fused Inria call (csrc/gspl_composite.h, `SegState`; round 5): the forward leaves a
checkpoint of every pixel's running state each 256 list entries, and a tile whose walk is longer than that is cut into segments that
independent workgroups of the backward process — in a scene with heavy-tailed lists (a trained model) the launch no longer lasts as
long as its longest tile.

Checked here: on a scene whose longest walks run to several segments, radii are identical and the image equal to rounding (the
forward sums the colour per segment) with the feature off
(`ops.SEGMENTED_BACKWARD = False`: the plain one-workgroup-per-tile walk), all five parameter gradients and `viewspace_points.grad`
agree with it to the spread of the backward's fp32 atomics, the backward did publish segments, and a frame without a long walk
publishes none.
Against the fp64 oracle the same path runs in tests/test_locked_parity.py / test_metric_point_parity.py (`scene_surfaces`, vanilla API)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEG = 256      # csrc/gspl_composite.h


def _render(params, cam, segmented):
    from gspl_amd import ops
    ops.SEGMENTED_BACKWARD = "always" if segmented else False
    ops.KEEP_LAST_RASTER = True
    try:
        W, H = cam["width"], cam["height"]
        leaves = [t.to(DEV).requires_grad_(True) for t in params]
        m, s, q, o, c = leaves
        settings = ops.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.tensor([0.1, 0.2, 0.3], device=DEV), scale_modifier=1.0,
            viewmatrix=cam["world_to_camera"].to(DEV), projmatrix=cam["full_projection"].to(DEV), sh_degree=3, campos=cam["camera_center"].to(DEV))
        screen = torch.zeros_like(m, requires_grad=True)
        render, radii = ops.GaussianRasterizer(settings)(means3D=m, means2D=screen, opacities=o, shs=c, scales=s, rotations=q)
        last = ops.LAST_RASTER
        th, tw = (H + 15) // 16, (W + 15) // 16
        pad = torch.zeros((th * 16, tw * 16), dtype=torch.int32, device=DEV)
        pad[:H, :W] = last["last_ids"]
        walked = (pad.view(th, 16, tw, 16).amax(dim=(1, 3)).reshape(-1) - last["offsets"][:th * tw]).clamp_min(0)
        wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(9)).to(DEV)
        (render * wimg).sum().backward()
        torch.cuda.synchronize()
        count = None if last.get("segment_count") is None else int(last["segment_count"].item())      # (published by the backward)
        return (render.detach().clone(), radii.clone(), [t.grad.clone() for t in leaves] + [screen.grad.clone()], count, int(walked.max()),
                int(((walked + SEG - 1) // SEG - 1).clamp_min(0).sum()))
    finally:
        ops.SEGMENTED_BACKWARD = True
        ops.KEEP_LAST_RASTER = False


def _close(a, b, name, rel=5e-5):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    rms = float(np.sqrt(np.mean(b * b))) + 1e-30
    ratio = np.abs(a - b) / (np.abs(b) + rms)
    assert ratio.max() <= rel, f"{name}: segmented vs plain backward differ by {ratio.max():.3e} of |ref| + rms"


@pytest.mark.parametrize("workload, scale", [("S-smoke-surfaces", 1.0), ("S-smoke-surfaces", 1.6)])
def test_segmented_backward_equals_the_plain_walk(workload, scale):
    import gspl_amd  # noqa: F401
    from gspl_amd import synthetic
    wl = synthetic.WORKLOADS[workload]
    means, scales, quats, opac, shs = synthetic.workload_scene(wl, seed=42)
    params = (means, scales * scale, quats, opac, shs)
    cam = synthetic.camera(wl["width"], wl["height"], wl["fx"])
    img_s, radii_s, grads_s, count, longest, expected = _render(params, cam, True)
    img_p, radii_p, grads_p, count_p, _, _ = _render(params, cam, False)
    assert count_p is None                                          # off: no checkpoints were taken
    assert longest > 1024, f"the scene has no long walk (longest {longest}): nothing is being tested"
    assert count == expected and count >= 2, (count, expected)      # the backward published exactly the segments beyond each tile's first
    # the forward sums the colour per segment when it takes checkpoints: the image equals the plain one to fp32 rounding, not bit for bit
    assert torch.equal(radii_s, radii_p) and float((img_s - img_p).abs().max()) <= 2e-6
    names = ("means", "scales", "quats", "opacities", "shs", "viewspace_points.grad")
    for a, name in zip(grads_s, names):
        assert bool(torch.isfinite(a).all()), name
    # what the compositing backward delivers directly (the screen-space gradient, dL/dopacity, dL/dcolour through the linear SH
    # backward): equal to the spread of its fp32 atomics
    for k in (5, 3, 4):
        _close(grads_s[k], grads_p[k], names[k])
    # behind conic -> cov2D -> cov3D a difference of 1e-6 in dL/dconic is amplified by the splat's conditioning (the needles of
    # scene_surfaces: kappa ~ 4000, tests/test_locked_parity.py): nearly every element to the same 5e-5 (the handful of needle rows beyond it are what
    # tests/test_locked_parity.py bounds against the oracle, element by element)
    for k in (0, 1, 2):
        a, b = grads_s[k].double().cpu().numpy(), grads_p[k].double().cpu().numpy()
        rms = float(np.sqrt(np.mean(b * b))) + 1e-30
        ratio = np.abs(a - b) / (np.abs(b) + rms)
        assert (ratio <= 5e-5).mean() >= 0.995 and ratio.max() <= 0.5, (names[k], float((ratio > 5e-5).mean()), float(ratio.max()))


def test_a_frame_of_short_walks_lists_no_segment():
    import gspl_amd  # noqa: F401
    from gspl_amd import synthetic
    wl = synthetic.WORKLOADS["S-smoke"]
    means, scales, quats, opac, shs = synthetic.workload_scene(wl, seed=42)
    params = (means, scales * 0.5, quats, opac, shs)
    cam = synthetic.camera(wl["width"], wl["height"], wl["fx"])
    _, _, _, count, longest, expected = _render(params, cam, True)
    assert longest <= SEG and expected == 0, longest
    assert count in (0, None)                                      # (None: the whole list shorter than one segment — no checkpoints at all)
