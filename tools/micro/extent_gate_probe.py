# How many integer radii of the fp32 preprocess differ from the fp64 oracle's, and how close to an integer the fp64 extent
# 3 sqrt(lambda_max) of those splats is (profiles/r05b_extent_probe.txt).  A variant that re-evaluated gated splats in fp64 inside
# the kernel (round 4) removed them at +12 us per frame (102 VGPRs + a call frame for a one-in-a-million event) and was not kept.
import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import gspl_amd
from gspl_amd import ops, synthetic, _lib
from oracle import gsplat_oracle as O
dev = torch.device("cuda:0")
for n, W, H, fx in ((1_000_000, 1920, 1080, 1600.0), (300_000, 800, 800, 1111.1)):
    means, scales, quats, opac, shs = synthetic.scene(n, seed=42)
    cam = synthetic.camera(W, H, fx)
    settings = ops.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
        scale_modifier=1.0, viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=3, campos=cam["camera_center"].to(dev))
    with torch.no_grad():
        _, radii = ops.GaussianRasterizer(settings)(means3D=means.to(dev), means2D=None, opacities=opac.to(dev), shs=shs.to(dev), scales=scales.to(dev), rotations=quats.to(dev))
    d = lambda t: t.double().to(dev)
    xy, depths, r64, conics, mask = O.inria_preprocess(d(means), d(scales), 1.0, d(quats), d(cam["world_to_camera"]), d(cam["full_projection"]),
                                                       cam["tanfovx"], cam["tanfovy"], H, W)
    mism = (radii.cpu() != r64.cpu())
    con = conics.cpu()[mism]
    if len(con):
        a, b, c = con[:, 0], con[:, 1], con[:, 2]
        mid = 0.5 * (a + c)
        lam_min = mid - torch.sqrt(torch.clamp(mid * mid - (a * c - b * b), min=0.0))
        v = 3.0 / torch.sqrt(lam_min)
        for k in range(len(con)):
            print(f"   splat {int(torch.nonzero(mism)[k])}: hip radius {int(radii.cpu()[mism][k])}, fp64 radius {int(r64.cpu()[mism][k])}, "
                  f"fp64 extent {float(v[k]):.9f} (distance to an integer {abs(float(v[k]) - round(float(v[k]))):.2e})")
    print(os.environ.get("GSPL_HIP_LIB", "default"), f"N={n}: {int(mism.sum())} radii differ from the fp64 oracle's (visible {int((r64 > 0).sum())})")
