"""Per camera of the bench's heterogeneous set: longest walk, mean walk over the non-empty tiles (what the adaptive switch of the segmented
backward looks at, gspl_composite.h) and the compositing backward's time with the segmented form forced off / on."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gspl_amd  # noqa
from gspl_amd import ops, synthetic, _lib

DEV = "cuda:0"
for name in sys.argv[1:] or ["S-1080p-1M", "S-1080p-1M-surfaces"]:
    wl = synthetic.WORKLOADS[name]
    W, H = wl["width"], wl["height"]
    params = [t.to(DEV).requires_grad_(True) for t in synthetic.workload_scene(wl, seed=42)]
    m, s, q, o, c = params
    cams = synthetic.camera_set(W, H, wl["fx"], 16, distance=wl.get("distance", 4.0))
    ops.KEEP_LAST_RASTER = True
    for i, cam in enumerate(cams):
        st = ops.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=DEV), scale_modifier=1.0,
                                               viewmatrix=cam["world_to_camera"].to(DEV), projmatrix=cam["full_projection"].to(DEV), sh_degree=3, campos=cam["camera_center"].to(DEV))
        res = {}
        for mode in (False, "always"):
            ops.SEGMENTED_BACKWARD = mode
            ts = []
            for rep in range(4):
                for t in params:
                    t.grad = None
                render, radii = ops.GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, shs=c, scales=s, rotations=q)
                loss = (render - 0.5).abs().mean()
                torch.cuda.synchronize()
                _lib.profile_start(("gspl_composite_bwd_packed", "gspl_composite_bwd", "gspl_composite_fwd"), period=1)
                loss.backward()
                torch.cuda.synchronize()
                prof = _lib.profile_stop()
                ts.append(sum(prof.get("gspl_composite_bwd_packed", [0.0])))
            res[mode] = min(ts[1:])
        last = ops.LAST_RASTER
        th, tw = (H + 15) // 16, (W + 15) // 16
        pad = torch.zeros((th * 16, tw * 16), dtype=torch.int32, device=DEV)
        pad[:H, :W] = last["last_ids"]
        walked = (pad.view(th, 16, tw, 16).amax(dim=(1, 3)).reshape(-1) - last["offsets"][:th * tw]).clamp_min(0)
        ne = walked[walked > 0].float()
        print(f"{name} cam {i:2d}: longest {int(walked.max()):5d} mean(non-empty) {float(ne.mean()):7.1f} ratio {float(walked.max()) / float(ne.mean()):5.2f} p99 {float(ne.quantile(0.99)):7.1f} "
              f"non-empty {ne.numel()} of {walked.numel()}; bwd plain {res[False]:.4f} ms, segmented {res['always']:.4f} ms", flush=True)
    ops.SEGMENTED_BACKWARD = True
