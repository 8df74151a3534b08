"""
make_golden.py — generates tests/golden/*.npz by IMPORTING the reference's own pure-PyTorch code
(internal/utils/gaussian_projection.py, internal/utils/sh_utils.py) in the build container.

Run from the repo root:   python tests/golden/make_golden.py
It needs /root/reference (read-only, not present on the GPU box), which is why its outputs are
committed.  Nothing under tests/ reads /root/reference at test time.

Fixtures
  ref_projection.npz   project_gaussians() on a seeded 1500-Gaussian scene (two cameras: centred and
                       off-axis/rotated) + autograd gradients of a fixed random linear loss
  ref_sh.npz           eval_sh / eval_sh_decomposed for degrees 0..4 + gradients
  ref_sortkey.npz      build_gaussian_sort_key() (python-loop key builder) on a 200-Gaussian subset
  ref_kat.npz          the literal 4-Gaussian known-answer vector of the reference's
                       tests/gaussian_projection_test.py:30-113 (inputs and expected values), plus the
                       current reference function's output on it (xys differ from the literals by the
                       documented +0.5 px convention change)
  ref_ssim.npz         l1_loss() / ssim() of internal/utils/ssim.py on seeded image pairs (smooth, noisy, constant,
                       odd sizes that do not fill the 16x16 tiles, batch of 2) + autograd gradients w.r.t. the first
                       image of the reference loss 0.8 L1 + 0.2 (1 - SSIM)
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


gp = _load("ref_gaussian_projection", "internal/utils/gaussian_projection.py")
sh = _load("ref_sh_utils", "internal/utils/sh_utils.py")


def scene(n, seed):
    g = torch.Generator().manual_seed(seed)
    means = (torch.rand(n, 3, generator=g) * 2 - 1) * 1.3
    scales = torch.exp(torch.randn(n, 3, generator=g) * 0.6 - 3.2)
    quats = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    return means, scales, quats


def look_at_w2c(eye, target, up):
    """transposed (row-vector) world->camera like internal/cameras/cameras.py:147-153"""
    eye, target, up = [torch.tensor(v, dtype=torch.float64) for v in (eye, target, up)]
    z = torch.nn.functional.normalize(target - eye, dim=0)
    x = torch.nn.functional.normalize(torch.linalg.cross(z, up), dim=0)
    y = torch.linalg.cross(z, x)
    R = torch.stack([x, y, z])          # rows: camera axes in world
    t = -R @ eye
    m = torch.eye(4, dtype=torch.float64)
    m[:3, :3] = R
    m[:3, 3] = t
    return m.T.contiguous().float()


def gen_projection():
    out = {}
    means, scales, quats = scene(1500, 7)
    cams = [
        dict(w2c=look_at_w2c((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), (0.0, -1.0, 0.0)), fx=420.0, fy=415.0, cx=160.0, cy=120.0, W=320, H=240),
        dict(w2c=look_at_w2c((2.1, -1.2, -2.4), (0.1, 0.2, 0.0), (0.1, -1.0, 0.05)), fx=300.0, fy=310.0, cx=171.5, cy=101.25, W=333, H=211),
    ]
    out["means"], out["scales"], out["quats"] = means.numpy(), scales.numpy(), quats.numpy()
    g = torch.Generator().manual_seed(11)
    for ci, cam in enumerate(cams):
        m = means.clone().requires_grad_(True)
        s = scales.clone().requires_grad_(True)
        q = quats.clone().requires_grad_(True)
        res = gp.project_gaussians(
            means_3d=m, scales=s, scale_modifier=1.0, quaternions=q, world_to_camera=cam["w2c"],
            fx=torch.tensor(cam["fx"]), fy=torch.tensor(cam["fy"]), cx=torch.tensor(cam["cx"]), cy=torch.tensor(cam["cy"]),
            img_height=torch.tensor(cam["H"]), img_width=torch.tensor(cam["W"]), block_width=16)
        xys, depths, radii, conics, comp, tiles, cov3d, mask, rmin, rmax = res
        # fixed random linear loss over the differentiable outputs
        w_xy = torch.randn(xys.shape, generator=g)
        w_d = torch.randn(depths.shape, generator=g)
        w_c = torch.randn(conics.shape, generator=g) * 0.1
        w_k = torch.randn(comp.shape, generator=g)
        loss = (xys * w_xy).sum() + (depths * w_d).sum() + (conics * w_c).sum() + (comp * w_k).sum()
        loss.backward()
        p = f"cam{ci}_"
        out[p + "w2c"] = cam["w2c"].numpy()
        out[p + "intr"] = np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["W"], cam["H"]], np.float64)
        for k, v in dict(xys=xys, depths=depths, radii=radii, conics=conics, comp=comp, tiles=tiles, cov3d=cov3d, mask=mask,
                         rect_min=rmin, rect_max=rmax, w_xy=w_xy, w_d=w_d, w_c=w_c, w_k=w_k,
                         g_means=m.grad, g_scales=s.grad, g_quats=q.grad).items():
            out[p + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "ref_projection.npz"), **out)
    print("ref_projection.npz", {k: v.shape for k, v in out.items() if k.startswith("cam0_")})


def gen_sh():
    out = {}
    g = torch.Generator().manual_seed(5)
    n = 700
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    coeffs = torch.randn(n, 25, 3, generator=g) * 0.3          # [N,K,3], the model's layout
    w = torch.randn(n, 3, generator=g)
    out["dirs"], out["coeffs"], out["w"] = dirs.numpy(), coeffs.numpy(), w.numpy()
    for deg in range(5):
        K = (deg + 1) ** 2
        c = coeffs[:, :K].clone().requires_grad_(True)
        d = dirs.clone().requires_grad_(True)
        rgb = sh.eval_sh(deg, c.transpose(1, 2), d)             # eval_sh wants [N,3,K] (sh_utils.py:57-70)
        (rgb * w).sum().backward()
        out[f"deg{deg}_rgb"] = rgb.detach().numpy()
        out[f"deg{deg}_g_coeffs"] = c.grad.numpy()
        out[f"deg{deg}_g_dirs"] = (d.grad if d.grad is not None else torch.zeros_like(d)).numpy()
        rgb2 = sh.eval_sh_decomposed(deg, coeffs[:, :1], coeffs[:, 1:K], dirs)
        out[f"deg{deg}_rgb_decomposed"] = rgb2.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_sh.npz"), **out)
    print("ref_sh.npz ok")


def gen_sortkey():
    means, scales, quats = scene(200, 3)
    w2c = look_at_w2c((0.0, 0.0, -3.0), (0.0, 0.0, 0.0), (0.0, -1.0, 0.0))
    W, H = 200, 136
    res = gp.project_gaussians(means, scales, 1.0, quats, w2c, torch.tensor(260.0), torch.tensor(260.0),
                               torch.tensor(100.0), torch.tensor(68.0), torch.tensor(H), torch.tensor(W), 16)
    xys, depths, radii, conics, comp, tiles, cov3d, mask, rmin, rmax = res
    bounds = gp.build_tile_bounds(torch.tensor(H), torch.tensor(W), 16, device="cpu")
    cum = torch.cumsum(tiles, dim=0)
    keys, gids = gp.build_gaussian_sort_key(depths, rmin * mask[:, None], rmax * mask[:, None], bounds, cum)
    np.savez_compressed(os.path.join(HERE, "ref_sortkey.npz"), xys=xys.numpy(), depths=depths.numpy(), radii=radii.numpy(),
                        tiles=tiles.numpy(), keys_unsorted=keys.numpy(), gids_unsorted=gids.numpy(),
                        tile_bounds=bounds.numpy(), wh=np.array([W, H]))
    print("ref_sortkey.npz", keys.shape)


def gen_kat():
    """Literal vector from the reference's tests/gaussian_projection_test.py:30-113."""
    means = np.array([[4.9744410514831543, -1.6869305372238159, -1.0178891420364380],
                      [0.1855451613664627, 0.2173379510641098, -1.6864157915115356],
                      [14.9114608764648438, -4.6346273422241211, 1.8997575044631958],
                      [5.0085635185241699, -3.8657102584838867, -1.3707503080368042]], np.float32)
    scales = np.array([[0.1152868643403053, 0.0463323593139648, 0.0125905377790332],
                       [0.0036764058750123, 0.0155582446604967, 0.0025763553567231],
                       [0.0729999020695686, 0.1261776685714722, 0.0579524375498295],
                       [1.8269745111465454, 0.1552953571081161, 0.2113087177276611]], np.float32)
    quats = np.array([[0.6251348853111267, -0.7321968674659729, 0.2666733860969543, 0.0444900505244732],
                      [0.9881987571716309, -0.0445680879056454, -0.1419259905815125, 0.0365220829844475],
                      [0.9662694931030273, 0.1446461081504822, -0.1685470491647720, 0.1303553283214569],
                      [0.8739961385726929, -0.3649578392505646, 0.1373531222343445, -0.2899493575096130]], np.float32)
    w2c = np.array([[9.9991554021835327e-01, -1.2848137877881527e-02, -1.9360868027433753e-03, 0.0],
                    [-5.9221056289970875e-04, -1.9391909241676331e-01, 9.8101717233657837e-01, 0.0],
                    [-1.2979693710803986e-02, -9.8093330860137939e-01, -1.9391019642353058e-01, 0.0],
                    [-3.2830274105072021e-01, -1.9259561300277710e+00, 3.9580578804016113e+00, 1.0]], np.float32)
    intr = np.array([961.40997314453125, 962.802490234375, 648.5, 420.0, 1297, 840], np.float64)
    expected = dict(
        exp_xys_rows13_ndc_convention=np.array([[622.13409423828125, 351.81060791015625],
                                                [11359.6181640625, 656.73974609375]], np.float32),
        exp_radii=np.array([0, 4, 0, 16783], np.int32),
        exp_conics_masked=np.array([[1.1229337453842163, 1.4079402387142181e-01, 1.5783417224884033],
                                    [2.3913329982860887e-07, -9.6377800673508318e-07, 4.5153879000281449e-06]], np.float32),
        exp_comp_masked=np.array([0.5893613696098328, 0.9999994039535522], np.float32),
        exp_tiles_masked=np.array([4, 4346], np.int32),
        exp_cov3d_upper_masked=np.array([[1.3772079910268076e-05, -1.3363457583182026e-05, 3.2048776574811200e-06,
                                          2.3899228835944086e-04, -2.2861815523356199e-05, 9.4481683845515363e-06],
                                         [2.1180632114410400, -1.5923748016357422, -6.8420924246311188e-02,
                                          1.2517973184585571, 6.5218225121498108e-02, 3.6743372678756714e-02]], np.float32),
    )
    res = gp.project_gaussians(torch.tensor(means), torch.tensor(scales), 1.0, torch.tensor(quats), torch.tensor(w2c),
                               torch.tensor(intr[0], dtype=torch.float32), torch.tensor(intr[1], dtype=torch.float32),
                               torch.tensor(intr[2], dtype=torch.float32), torch.tensor(intr[3], dtype=torch.float32),
                               torch.tensor(840), torch.tensor(1297), 16)
    cur = dict(cur_xys=res[0].numpy(), cur_depths=res[1].numpy(), cur_radii=res[2].numpy(), cur_conics=res[3].numpy(),
               cur_comp=res[4].numpy(), cur_tiles=res[5].numpy(), cur_cov3d=res[6].numpy(), cur_mask=res[7].numpy())
    np.savez_compressed(os.path.join(HERE, "ref_kat.npz"), means=means, scales=scales, quats=quats, w2c=w2c, intr=intr,
                        **expected, **cur)
    print("ref_kat.npz radii", cur["cur_radii"], "xys", cur["cur_xys"][[1, 3]])


def gen_ssim():
    ss = _load("ref_ssim", "internal/utils/ssim.py")
    g = torch.Generator().manual_seed(11)
    out = {}

    def smooth(c, h, w):
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
        base = torch.stack([0.5 + 0.4 * torch.sin(6.0 * xx + k) * torch.cos(4.0 * yy - k) for k in range(c)])
        return base.clamp(0, 1)

    cases = {
        "smooth_3x37x53": (smooth(3, 37, 53), (smooth(3, 37, 53) + 0.05 * torch.randn(3, 37, 53, generator=g)).clamp(0, 1)),
        "noise_3x64x48": (torch.rand(3, 64, 48, generator=g), torch.rand(3, 64, 48, generator=g)),
        "const_3x20x20": (torch.full((3, 20, 20), 0.3), torch.full((3, 20, 20), 0.5)),
        "tiny_1x7x5": (torch.rand(1, 7, 5, generator=g), torch.rand(1, 7, 5, generator=g)),
        "batch_2x3x33x17": (torch.rand(2, 3, 33, 17, generator=g), torch.rand(2, 3, 33, 17, generator=g)),
    }
    for name, (a, b) in cases.items():
        a = a.clone().requires_grad_(True)
        a4 = a if a.dim() == 4 else a.unsqueeze(0)
        b4 = b if b.dim() == 4 else b.unsqueeze(0)
        l1 = ss.l1_loss(a4, b4)
        s = ss.ssim(a4, b4)
        loss = 0.8 * l1 + 0.2 * (1.0 - s)
        (grad,) = torch.autograd.grad(loss, a)
        out[name + "/img1"] = a.detach().numpy()
        out[name + "/img2"] = b.numpy()
        out[name + "/l1"] = l1.detach().numpy()
        out[name + "/ssim"] = s.detach().numpy()
        out[name + "/grad"] = grad.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_ssim.npz"), **out)
    print("ref_ssim.npz", {k: float(v) for k, v in out.items() if k.endswith("/ssim")})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are committed, nothing to do")
    torch.manual_seed(0)
    gen_projection()
    gen_sh()
    gen_sortkey()
    gen_kat()
    gen_ssim()
