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
        _render.packed = ops.LAST_RASTER["packed_grads"].clone()      # the compositing backward's own rows: x y | a b c | opacity | r g b
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
    packed_s = _render.packed
    img_p, radii_p, grads_p, count_p, _, _ = _render(params, cam, False)
    packed_p = _render.packed
    assert count_p is None                                          # off: no checkpoints were taken
    # the compositing kernel's OWN gradients (dL/dmeans2d, dL/dconic, dL/dopacity, dL/dcolour per splat), before the
    # ill-conditioned cov chain amplifies anything: segmented and plain walks agree column by column to 1e-5 of |ref| + rms
    # (dL/dmeans2d = A Sx + B Sy, two products far larger than their sum along a needle: 5e-5 there)
    for lo, hi, name, rel in ((0, 2, "dL/dmeans2d", 5e-5), (2, 5, "dL/dconic", 2e-5), (5, 6, "dL/dopacity", 2e-5), (6, 9, "dL/dcolour", 2e-5)):
        _close(packed_s[:, lo:hi], packed_p[:, lo:hi], "compositing " + name, rel=rel)
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


def test_the_adaptive_switch_asks_for_a_tail_not_for_length():
    """Round 6: the segmented form switches itself on when the longest walk of a frame exceeds 768 entries AND 4.5 times the mean walk, and remembers the verdict per view
    (gspl_composite.h, ADAPTIVE: every backward leaves its tiles' walk lengths, the next forward reduces them).  A trained-scene-shaped
    frame (heavy tail) turns it on within a few frames; a frame whose walks are uniformly long — every tile of a close-up view of a
    translucent cloud — must NOT (there is no tail to spread; the checkpoints and the second launch cost 38 us per step at S-1080p-1M,
    profiles/r22_records_48B_ab.txt)."""
    import time
    import gspl_amd  # noqa: F401
    from gspl_amd import ops, synthetic

    def frames(params, cam, n):
        ops.SEGMENTED_BACKWARD = True
        ops.KEEP_LAST_RASTER = True
        try:
            W, H = cam["width"], cam["height"]
            leaves = [t.to(DEV).requires_grad_(True) for t in params]
            m, s, q, o, c = leaves
            settings = ops.GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=DEV), scale_modifier=1.0,
                viewmatrix=cam["world_to_camera"].to(DEV), projmatrix=cam["full_projection"].to(DEV), sh_degree=3, campos=cam["camera_center"].to(DEV))
            on, longest, mean = [], 0, 0.0
            for _ in range(n):
                render, radii = ops.GaussianRasterizer(settings)(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, shs=c, scales=s, rotations=q)
                last = ops.LAST_RASTER
                on.append(last.get("segment_count") is not None)
                th, tw = (H + 15) // 16, (W + 15) // 16
                pad = torch.zeros((th * 16, tw * 16), dtype=torch.int32, device=DEV)
                pad[:H, :W] = last["last_ids"]
                walked = (pad.view(th, 16, tw, 16).amax(dim=(1, 3)).reshape(-1) - last["offsets"][:th * tw]).clamp_min(0)
                longest, mean = int(walked.max()), float(walked[walked > 0].float().mean())
                render.sum().backward()
                torch.cuda.synchronize()
                time.sleep(0.01)          # (the host's word is written by the NEXT forward's kernel and read without a synchronisation)
            return on, longest, mean
        finally:
            ops.KEEP_LAST_RASTER = False

    # (a) uniformly long walks: a dense translucent cloud seen from close by
    wl = synthetic.WORKLOADS["S-smoke"]
    means, scales, quats, opac, shs = synthetic.scene(60_000, seed=5)
    cam = synthetic.camera(wl["width"], wl["height"], wl["fx"], distance=3.0)
    # (72 frames: whatever an earlier test of this process left of the 64-frame stickiness has run out by the end)
    on, longest, mean = frames((means, scales * 4.0, quats, opac * 0.15, shs), cam, 72)
    print(f"uniform cloud: longest walk {longest}, mean {mean:.0f}, segmented frames {sum(on)} of {len(on)}")
    assert longest > 768 and longest < 4.0 * mean, (longest, mean)          # long, but no tail
    assert not any(on[-4:]), f"frames of uniformly long walks (longest {longest}, mean {mean:.0f}) keep the segmented form on"
    # (b) a heavy tail: the same kind of cloud, thinner, plus a knot of 5000 faint splats behind one corner of the image — a handful of
    # tiles walk thousands of entries, the others a hundred
    means, scales, quats, opac, shs = synthetic.scene(25_000, seed=6)
    g = torch.Generator().manual_seed(7)
    knot = torch.tensor([-1.0, -0.6, 0.5]) + 0.04 * torch.randn(5000, 3, generator=g)
    means = torch.cat([means, knot])
    scales = torch.cat([scales * 3.0, torch.full((5000, 3), 0.02)])
    quats = torch.cat([quats, torch.nn.functional.normalize(torch.randn(5000, 4, generator=g), dim=-1)])
    opac = torch.cat([opac * 0.3, torch.full((5000, 1), 0.01)])
    shs = torch.cat([shs, 0.2 * torch.randn(5000, 16, 3, generator=g)])
    cam = synthetic.camera(wl["width"], wl["height"], wl["fx"])
    on, longest, mean = frames((means, scales, quats, opac, shs), cam, 8)
    print(f"cloud with a knot: longest walk {longest}, mean {mean:.0f}, segmented frames {on}")
    assert longest > 768 and longest > 8.0 * mean, (longest, mean)
    assert not on[0] and any(on[2:]), f"a frame with a tail (longest {longest}, mean {mean:.0f}) never switched the segmented form on: {on}"
