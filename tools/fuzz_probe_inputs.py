"""usage (on the GPU box): python tools/fuzz_probe_inputs.py seed [seed ...]
For a case of tools/fuzz_parity.py: how far the per-splat values the GPU composites (pixel position, conic, colour, opacity) are from the
fp64 oracle's, per renderer API, and the worst pixel — separates 'the compositing differs' from 'its inputs differ by their fp32 rounding'.
Test infrastructure: imports oracle/."""
import os, sys, math, types
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tools"))
import numpy as np
import torch
import fuzz_parity as F
from hip_helpers import fragile_rows
argv = sys.argv
from gspl_amd.ops._state import STATE as S
O, hip, dev = F.O, F.hip, F.dev
S.keep_last_raster = True

for seed in [int(a) for a in argv[1:]]:
    desc, case = F.random_case(seed)
    means, scales, quats, opac, shs, cam, wimg, bg = case
    W, H = cam["width"], cam["height"]
    deg = int(math.isqrt(shs.shape[1])) - 1
    print(seed, desc)
    for api in ("gsplat", "inria"):
        m, s, q, o, c = F.cuda(means, scales, quats, opac, shs)
        dl = [t.double() for t in (means, scales, quats, opac, shs)]
        if api == "gsplat":
            vm = cam["world_to_camera"].T.contiguous().float().to(dev)
            xys, depths, radii, conics, comp, tiles, _ = hip.project_gaussians(m, s, 1.0, q, vm[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, 16)
            rgbs = hip.sh_view_colors(deg, m, cam["camera_center"].to(dev), c, None, radii > 0)
            img = hip.rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, o * comp[:, None], H, W, 16, bg.to(dev)).permute(2, 0, 1)
            r = O.render_gsplat(*dl, deg, cam["world_to_camera"].double(), cam["fx"], cam["fy"], cam["cx"], cam["cy"], W, H, bg.double(), cam["camera_center"].double())
            vis = (radii > 0).cpu().numpy() & r["mask"].numpy()
            mode = O.MODE_GSPLAT
        else:
            st = hip.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(dev), scale_modifier=1.0,
                                                   viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=deg, campos=cam["camera_center"].to(dev))
            img, radii = hip.GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), opacities=o, shs=c, scales=s, rotations=q)
            r = O.render_inria(*dl, deg, cam["world_to_camera"].double(), cam["full_projection"].double(), cam["camera_center"].double(), cam["tanfovx"], cam["tanfovy"], W, H, bg.double())
            vis = (radii > 0).cpu().numpy() & (r["radii"].numpy() > 0)
            mode = O.MODE_INRIA
        lr = S.last_raster
        xy = (r["xys"] if "xys" in r else r["xy"]).detach().numpy()
        g = {k: lr[k].detach().cpu().numpy().astype(np.float64) for k in ("means2d", "conics", "colors")}
        gop = lr["opacities"].detach().cpu().numpy().astype(np.float64).reshape(-1)
        rop = (r["opacities"] if "opacities" in r else dl[3]).detach().numpy().reshape(-1)
        dxy = np.abs(g["means2d"] - xy)[vis]
        rc = r["conics"].detach().numpy()
        dcon = (np.abs(g["conics"] - rc) / (np.abs(rc).max(axis=1, keepdims=True) + 1e-30))[vis]
        dcol = np.abs(g["colors"] - r["rgbs"].detach().numpy())[vis]
        dop = np.abs(gop - rop)[vis]
        d = np.abs(img.detach().cpu().numpy().astype(np.float64) - r["render"].detach().numpy()).max(axis=0)
        rows, frag = fragile_rows(mode, r, W, H, bg.double(), opacities=dl[3], gpu_radii=radii)
        y, x = np.unravel_index(np.argmax(np.where(frag, 0, d)), d.shape)
        print(f"  {api}: visible {int(vis.sum())}; |xy| max {dxy.max():.2e} px (|xy| up to {np.abs(xy[vis]).max():.0f}); conic rel max {dcon.max():.2e}; colour max {dcol.max():.2e}; "
              f"opacity max {dop.max():.2e}; worst unflagged pixel {d[y, x]:.2e} at ({x},{y}), worst pixel {d.max():.2e}")
        if os.environ.get("PROBE_PIXEL"):
            # the worst unflagged pixel: per blended splat, alpha from the GPU's per-splat values and from the oracle's (fp64 arithmetic on both)
            off, ids = np.asarray(r["offsets"]).reshape(-1), np.asarray(r["flatten_ids"]).reshape(-1)
            tw = (W + 15) // 16
            t = (y // 16) * tw + (x // 16)
            lo = int(off[t]); hi = int(off[t + 1]) if t + 1 < off.size else ids.size
            px, py = (x + 0.5, y + 0.5) if api == "gsplat" else (float(x), float(y))
            T_g = T_o = 1.0
            for k in ids[lo:hi]:
                al = []
                for src_xy, src_con, src_op in ((g["means2d"][k], g["conics"][k], gop[k]), (xy[k], rc[k], rop[k])):
                    dx, dy = src_xy[0] - px, src_xy[1] - py
                    sig = 0.5 * (src_con[0] * dx * dx + src_con[2] * dy * dy) + src_con[1] * dx * dy
                    al.append(min(0.99, src_op * math.exp(-sig)) if sig >= 0 else 0.0)
                if max(al) < 1.0 / 255:
                    continue
                print(f"    splat {k}: alpha gpu-inputs {al[0]:.8f} oracle {al[1]:.8f} (diff {al[0] - al[1]:+.2e}); dxy {g['means2d'][k] - xy[k]}; xy {xy[k]}; conic {rc[k]}; "
                      f"T before {T_o:.5f}; colour {r['rgbs'].detach().numpy()[k]}")
                T_g *= 1 - al[0]; T_o *= 1 - al[1]
