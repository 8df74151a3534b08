"""GPU diagnostic for tests/hip_helpers.assert_close_attributed: on the scenes of the free-running pipeline tests, how many gradient
elements beyond 1e-4 lie OUTSIDE the rows a fragile decision reaches (must be 0 for the attribution assert to be usable)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gspl_amd  # noqa
from gspl_amd import ops
from oracle import gsplat_oracle as O
from hip_helpers import fragile_rows

DEV = "cuda:0"


def run(api, n, W, H, fx, seed, mul):
    ops.KEEP_LAST_RASTER = True
    means, scales, quats, opac, shs = O.synthetic_scene(n, seed=seed)
    scales = scales * mul
    cam = O.synthetic_camera(W, H, fx, fx - 5.0)
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(seed))
    bg = torch.tensor([0.25, 0.5, 0.125])
    leaves = [t.to(DEV).requires_grad_(True) for t in (means, scales, quats, opac, shs)]
    m, s, q, o, c = leaves
    dl = [t.double().requires_grad_(True) for t in (means, scales, quats, opac, shs)]
    if api == "inria":
        st = ops.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(DEV), scale_modifier=1.0,
                                               viewmatrix=cam["world_to_camera"].to(DEV), projmatrix=cam["full_projection"].to(DEV), sh_degree=3, campos=cam["camera_center"].to(DEV))
        screen = torch.zeros_like(m, requires_grad=True)
        render, radii = ops.GaussianRasterizer(st)(means3D=m, means2D=screen, opacities=o, shs=c, scales=s, rotations=q)
        r = O.render_inria(*dl, 3, cam["world_to_camera"].double(), cam["full_projection"].double(), cam["camera_center"].double(), cam["tanfovx"], cam["tanfovy"], W, H, bg.double())
        mode = O.MODE_INRIA
    else:
        vm = cam["world_to_camera"].T.contiguous().to(DEV)
        xys, depths, radii, conics, comp, tiles, _ = ops.project_gaussians(m, s, 1.0, q, vm[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, 16)
        rgbs = ops.sh_view_colors(3, m, cam["camera_center"].to(DEV), c, None, radii > 0)
        render = ops.rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, o * comp[:, None], H, W, 16, bg.to(DEV)).permute(2, 0, 1)
        r = O.render_gsplat(*dl, 3, cam["world_to_camera"].double(), cam["fx"], cam["fy"], cam["cx"], cam["cy"], W, H, bg.double(), cam["camera_center"].double())
        mode = O.MODE_GSPLAT
    (render * wimg.to(DEV)).sum().backward()
    (r["render"] * wimg.double()).sum().backward()
    rows, frag = fragile_rows(mode, r, W, H, bg.double(), opacities=dl[3], gpu_radii=radii)
    d = (render.detach().cpu().double() - r["render"].detach()).abs().max(dim=0).values.numpy()
    print(f"== {api} n={n} {W}x{H} mul={mul}: fragile px {int(frag.sum())}/{frag.size}; fragile rows {int(rows.sum())}/{rows.size}; radii differ {int((radii.cpu().numpy().reshape(-1) != r['radii'].numpy().reshape(-1)).sum())}; "
          f"px beyond 1e-5: {int((d > 1e-5).sum())} (unflagged: {int(((d > 1e-5) & ~frag).sum())}, worst unflagged {d[~frag].max():.2e})")
    unfl = (d > 1e-5) & ~frag
    if unfl.any():
        last = ops.LAST_RASTER
        vals = [last[k].detach().cpu().double() for k in ("means2d", "conics", "colors", "opacities")]
        lk, _, _, fl = O.composite_fwd(mode, vals[0], vals[1], vals[2], vals[3].reshape(-1), bg.double(), W, H, last["offsets"].cpu().numpy(), last["flatten_ids"].cpu().numpy())
        dl_ = np.abs(render.detach().cpu().double().permute(1, 2, 0).numpy() - lk).max(axis=-1)
        ys, xs = np.nonzero(unfl)
        for y, x in list(zip(ys, xs))[:6]:
            print(f"      unflagged px ({x},{y}): |gpu - free oracle| {d[y, x]:.2e}; |gpu - oracle at the GPU's values| {dl_[y, x]:.2e} (flagged there: {int(fl[y, x])})")
    for got, ref, name in zip(leaves, dl, ("means", "scales", "quats", "opacities", "shs")):
        g, rf = got.grad.cpu().double().numpy(), ref.grad.numpy()
        rms = np.sqrt(np.mean(rf * rf)) + 1e-30
        ratio = np.abs(g - rf) / (np.abs(rf) + rms)
        rw = np.broadcast_to(rows.reshape((-1,) + (1,) * (rf.ndim - 1)), rf.shape)
        bad = ratio > 1e-4
        un = bad & ~rw
        print(f"   {name}: beyond 1e-4: {int(bad.sum())} of {bad.size}; unattributed {int(un.sum())}; worst unattributed ratio {float(np.where(~rw, ratio, 0).max()):.2e}; worst {ratio.max():.2e}")
        if un.any():
            idx = np.argwhere(un)[:5]
            for ix in idx:
                print("      row", ix[0], "ratio", ratio[tuple(ix)], "ref", rf[tuple(ix)], "got", g[tuple(ix)], "radius", int(r["radii"].reshape(-1)[ix[0]]))


for api in ("inria", "gsplat"):
    run(api, 20000, 320, 208, 300.0, 42, 4)
    run(api, 200000, 960, 544, 800.0, 7, 1)
    run(api, 1000000, 1920, 1080, 1600.0, 42, 1)
