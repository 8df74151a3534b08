"""usage (on the GPU box): python tools/fuzz_parity.py <minutes> [first seed | seed,seed,...]
Randomised end-to-end parity campaign: scene size, image size (odd, tiny, long-and-thin), tile-unfriendly extents, SH degree, splat scale
(sub-pixel to screen-filling), opacity distribution, camera pose (rotated, close, partly behind the near plane), background — both renderer
APIs against the fp64 oracle with the tests' attribution bars (tests/hip_helpers.assert_pipeline_attributed).  Test infrastructure: imports
oracle/.  Prints one line per case and a summary; exit code 1 if any case failed."""
import os, sys, time, math, traceback
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch
import gspl_amd  # noqa: F401
from gspl_amd import ops as hip, _lib
from oracle import gsplat_oracle as O
from hip_helpers import assert_pipeline_attributed

_lib.lib()
dev = torch.device("cuda:0")


def cuda(*ts):
    return [t.detach().float().contiguous().to(dev) for t in ts]


def random_case(seed):
    g = torch.Generator().manual_seed(seed)
    u = lambda: float(torch.rand((), generator=g))
    pick = lambda xs: xs[int(torch.randint(len(xs), (), generator=g))]
    n = int(pick([1, 2, 7, 63, 64, 65, 300, 1000, 2049, 5000, 12000]))
    W = int(pick([1, 5, 16, 17, 31, 64, 100, 199, 320, 333, 640]))
    H = int(pick([1, 3, 16, 33, 48, 97, 128, 211, 240, 400]))
    deg = int(pick([0, 1, 2, 3]))
    means, scales, quats, opac, shs = O.synthetic_scene(n, seed=seed, sh_degree=deg)
    smul = float(pick([0.05, 0.5, 1, 4, 4, 12]))
    scales = scales * smul
    kind = pick(["plain", "opaque", "faint", "mixed"])
    if kind == "opaque":
        opac = opac * 0 + 0.999
    elif kind == "faint":
        opac = opac * 0.02
    elif kind == "mixed":
        opac = torch.where(torch.rand(n, 1, generator=g) < 0.3, torch.full_like(opac, 0.0035), opac)      # around the 1/255 skip
    fx = float(pick([0.4, 0.9, 1.5])) * max(W, H) + 1.0
    dist = float(pick([0.7, 1.5, 4.0, 4.0, 9.0]))
    cam = O.synthetic_camera(W, H, fx, fx * (0.9 + 0.2 * u()), distance=dist)
    rotated = u() < 0.6
    if rotated:      # rotate the camera about the scene centre
        ax = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0)
        ang = (u() - 0.5) * 2.0
        Kx = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = torch.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * (Kx @ Kx)
        w2c = cam["world_to_camera"].clone()
        w2c[:3, :3] = R.T      # transposed storage: rows 0-2 hold R^T, row 3 the translation
        P = torch.linalg.inv(cam["world_to_camera"]) @ cam["full_projection"]
        cam["world_to_camera"] = w2c
        cam["full_projection"] = w2c @ P
        cam["camera_center"] = torch.linalg.inv(w2c)[3, :3]
    wimg = torch.randn(3, H, W, generator=g)
    bg = torch.rand(3, generator=g) if u() < 0.7 else torch.zeros(3)
    return dict(n=n, W=W, H=H, deg=deg, kind=kind, scale=smul, dist=dist, fx=round(fx), rot=rotated), (means, scales, quats, opac, shs, cam, wimg, bg)


def run_gsplat(case):
    means, scales, quats, opac, shs, cam, wimg, bg = case
    W, H = cam["width"], cam["height"]
    deg = int(math.isqrt(shs.shape[1])) - 1
    leaves = [t.requires_grad_(True) for t in cuda(means, scales, quats, opac, shs)]
    m, s, q, o, c = leaves
    vm = cam["world_to_camera"].T.contiguous().float().to(dev)
    xys, depths, radii, conics, comp, tiles, _ = hip.project_gaussians(m, s, 1.0, q, vm[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, 16)
    rgbs = hip.sh_view_colors(deg, m, cam["camera_center"].to(dev), c, None, radii > 0)
    img = hip.rasterize_gaussians(xys, depths, radii, conics, tiles, rgbs, o * comp[:, None], H, W, 16, bg.to(dev))
    render = img.permute(2, 0, 1)
    (render * wimg.to(dev)).sum().backward()
    dl = [t.double().requires_grad_(True) for t in (means, scales, quats, opac, shs)]
    r = O.render_gsplat(*dl, deg, cam["world_to_camera"].double(), cam["fx"], cam["fy"], cam["cx"], cam["cy"], W, H, bg.double(), cam["camera_center"].double())
    (r["render"] * wimg.double()).sum().backward()
    mism = int(np.sum((radii > 0).cpu().numpy() != r["mask"].numpy()))
    assert mism <= max(1, len(means) // 2000), f"visibility differs on {mism} splats"
    zero = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
    assert_pipeline_attributed(O.MODE_GSPLAT, r, W, H, bg.double(), render.detach().cpu().numpy(),
                               [(name, zero(got).cpu().numpy(), zero(ref).numpy()) for got, ref, name in zip(leaves, dl, ("means", "scales", "quats", "opacities", "shs"))],
                               gpu_radii=radii, quiet=True)


def run_inria(case):
    means, scales, quats, opac, shs, cam, wimg, bg = case
    W, H = cam["width"], cam["height"]
    deg = int(math.isqrt(shs.shape[1])) - 1
    leaves = [t.requires_grad_(True) for t in cuda(means, scales, quats, opac, shs)]
    m, s, q, o, c = leaves
    settings = hip.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg.to(dev), scale_modifier=1.0,
        viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev), sh_degree=deg, campos=cam["camera_center"].to(dev))
    screen = torch.zeros_like(m, requires_grad=True)
    render, radii = hip.GaussianRasterizer(settings)(means3D=m, means2D=screen, opacities=o, shs=c, scales=s, rotations=q)
    (render * wimg.to(dev)).sum().backward()
    dl = [t.double().requires_grad_(True) for t in (means, scales, quats, opac, shs)]
    r = O.render_inria(*dl, deg, cam["world_to_camera"].double(), cam["full_projection"].double(), cam["camera_center"].double(),
                       cam["tanfovx"], cam["tanfovy"], W, H, bg.double())
    (r["render"] * wimg.double()).sum().backward()
    mism = int(np.sum(radii.cpu().numpy() != r["radii"].numpy()))
    assert mism <= max(1, len(means) // 500), f"radii differ on {mism} splats"
    zero = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
    ref_ndc = zero(r["xy"]).numpy() * np.array([0.5 * W, 0.5 * H])
    assert_pipeline_attributed(O.MODE_INRIA, r, W, H, bg.double(), render.detach().cpu().numpy(),
                               [(name, zero(got).cpu().numpy(), zero(ref).numpy()) for got, ref, name in zip(leaves, dl, ("means", "scales", "quats", "opacities", "shs"))]
                               + [("viewspace_points.grad", zero(screen)[:, :2].cpu().numpy(), ref_ndc)], opacities=dl[3], gpu_radii=radii, quiet=True)


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    seeds = None
    if len(sys.argv) > 2 and "," in sys.argv[2]:
        seeds = [int(x) for x in sys.argv[2].split(",") if x]
    seed = seeds.pop(0) if seeds is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
    t0, done, failed = time.time(), 0, []
    while time.time() - t0 < minutes * 60:
        desc, case = random_case(seed)
        for api, fn in (("gsplat", run_gsplat), ("inria", run_inria)):
            try:
                fn(case)
                print(f"seed {seed} {api} {desc} ok", flush=True)
            except Exception as e:      # noqa: BLE001
                failed.append((seed, api, desc))
                msg = str(e).strip().splitlines()
                print(f"seed {seed} {api} {desc} FAILED: {type(e).__name__}: {msg[0] if msg else ''}", flush=True)
                if not isinstance(e, AssertionError):
                    traceback.print_exc()
            done += 1
        if seeds is None:
            seed += 1
        elif seeds:
            seed = seeds.pop(0)
        else:
            break
    print(f"{done} cases, {len(failed)} failed: {failed}")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
