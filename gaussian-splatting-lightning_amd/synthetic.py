"""Synthetic workloads of SURVEY.md §8(d) / BASELINE.md §2 (seed 42 = the reference's
`seed_everything_default`, internal/entrypoints/gspl.py:15).  Used by bench.py and smoke().

    S-1080p-1M : N = 1 000 000, 1920x1080, fx = fy = 1600   (the metric point)
    S-800-100k : N =   100 000,  800x800,  fx = fy = 1111.1 (lego proxy, configs[0]/[1])
"""
from __future__ import annotations

import torch

WORKLOADS = {
    "S-1080p-1M": dict(n=1_000_000, width=1920, height=1080, fx=1600.0),
    "S-800-100k": dict(n=100_000, width=800, height=800, fx=1111.1),
    "S-smoke": dict(n=20_000, width=320, height=208, fx=300.0),
    # stress points (not headline): garden-like count at the reference's images_4 resolution, and 4K
    "S-garden-6M": dict(n=6_000_000, width=1297, height=840, fx=961.4),
    "S-4k-2M": dict(n=2_000_000, width=3840, height=2160, fx=3200.0),
    # BASELINE.json configs[2] proxy: a garden-sized model (~6 M Gaussians) at the metric resolution
    "S-1080p-6M": dict(n=6_000_000, width=1920, height=1080, fx=1600.0),
    # camera INSIDE the cloud (distance 0.4 from its centre): roughly half of the Gaussians are behind the camera or outside
    # the frustum, as in real captures; exercises the visibility-masked SH loads
    "S-1080p-1M-inside": dict(n=1_000_000, width=1920, height=1080, fx=1600.0, distance=0.4),
    # BASELINE.json configs[4] proxy (MatrixCity aerial partition, configs/matrixcity/gsplat-aerial.yaml:25 + configs/gsplat-absgrad.yaml:6-8):
    # ~20 M small Gaussians at SH degree 0 (56 B of parameters each), gsplat API, absgrad densification statistics
    "S-1080p-20M-sh0-absgrad": dict(n=20_000_000, width=1920, height=1080, fx=1600.0, sh_degree=0, scale_mul=0.4, api="gsplat", absgrad=True),
    # a scene shaped like a TRAINED capture (`scene_surfaces`): opaque closed surfaces, opacity mass near 1, a heavy tail of large
    # anisotropic splats, an empty sky band — early termination after tens of splats, long lists in a few tiles
    "S-1080p-1M-surfaces": dict(n=1_000_000, width=1920, height=1080, fx=1600.0, scene="surfaces"),
    "S-smoke-surfaces": dict(n=30_000, width=320, height=208, fx=300.0, scene="surfaces", scale_mul=5.0),
}


def workload_scene(wl: dict, seed: int = 42):
    """`scene` with the workload's SH degree and scale multiplier."""
    make = scene_surfaces if wl.get("scene") == "surfaces" else scene
    means, scales, quats, opac, shs = make(wl["n"], seed=seed, sh_degree=wl.get("sh_degree", 3))
    return means, scales * wl.get("scale_mul", 1.0), quats, opac, shs


def scene(n: int, seed: int = 42, sh_degree: int = 3):
    """means U[-1.3,1.3]^3 (Blender init box, blender_dataparser.py:137), scales exp(N(-4.6,0.6)),
    unit quaternions, opacity sigmoid(N(0,1)), SH [N,K,3] ~ N(0, 0.2)."""
    g = torch.Generator().manual_seed(seed)
    K = (sh_degree + 1) ** 2
    means = (torch.rand(n, 3, generator=g) * 2 - 1) * 1.3
    scales = torch.exp(torch.randn(n, 3, generator=g) * 0.6 - 4.6)
    quats = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    opac = torch.sigmoid(torch.randn(n, 1, generator=g))
    shs = torch.randn(n, K, 3, generator=g) * 0.2
    return means, scales, quats, opac, shs


def scene_surfaces(n: int, seed: int = 42, sh_degree: int = 3):
    """A scene with the statistics of a TRAINED model rather than of an initialisation (`scene` above is the Blender init box,
    blender_dataparser.py:137: a uniform translucent cloud, 54 blended splats per pixel, no saturation anywhere):

      * Gaussians lie ON a few closed / opaque surfaces — a ground plane, a back wall and three ellipsoids standing in front of it —
        flattened along the surface normal (normal scale = 0.15 x the tangent scales) and rotated into the tangent frame, the way the
        optimiser of vanilla_gaussian.py leaves them;
      * opacity is bimodal with its mass near 1 (75 % sigmoid(N(3.5, 1.2)), 25 % sigmoid(N(-2, 1)): the survivors of the
        opacity resets and the ones about to be pruned, vanilla_density_controller.py:16-24) — a pixel saturates after tens of splats
        and the rear side of every object sits BEHIND the stop, in the lists but never composited;
      * tangent scales are log-normal with a heavy tail: 2 % needles (one tangent axis x 8), 0.5 % big blobs (both x 6): long lists in
        the tiles they cross;
      * the top ~30 % of the frame of `camera()` is sky: empty tiles next to full ones;
      * colours vary smoothly over a surface (SH dc from a low-frequency field), higher orders decay with the degree.

    Same return signature as `scene` (activated tensors).  Deterministic in `seed`."""
    import math
    g = torch.Generator().manual_seed(seed)
    K = (sh_degree + 1) ** 2
    rnd = lambda *s: torch.rand(*s, generator=g)
    nrm = lambda *s: torch.randn(*s, generator=g)
    # surface samplers: points p [m,3] and outward unit normals nv [m,3] (world frame of `camera()`: +y is down in the image, the
    # camera sits at z = -distance looking along +z)
    ellipsoids = [((-0.9, 0.3, 0.2), (0.6, 0.6, 0.6)), ((0.6, 0.45, -0.3), (0.45, 0.45, 0.45)), ((0.2, 0.05, 1.2), (0.9, 0.85, 0.7))]
    share = [0.33, 0.19] + [0.16, 0.10, 0.22]               # ground, wall, ellipsoids (roughly by area)
    counts = [int(n * f) for f in share]
    counts[0] += n - sum(counts)
    pts, nrms = [], []
    m = counts[0]                                           # ground y = 0.9, facing the camera's up (-y)
    pts.append(torch.stack([rnd(m) * 6 - 3, torch.full((m,), 0.9), rnd(m) * 6 - 2], dim=1))
    nrms.append(torch.tensor([0.0, -1.0, 0.0]).expand(m, 3))
    m = counts[1]                                           # back wall z = 2.5, facing the camera (-z)
    pts.append(torch.stack([rnd(m) * 5 - 2.5, rnd(m) * 1.5 - 0.6, torch.full((m,), 2.5)], dim=1))
    nrms.append(torch.tensor([0.0, 0.0, -1.0]).expand(m, 3))
    for (c, r), m in zip(ellipsoids, counts[2:]):
        u = torch.nn.functional.normalize(nrm(m, 3), dim=-1)
        c, r = torch.tensor(c), torch.tensor(r)
        pts.append(c + u * r)
        nrms.append(torch.nn.functional.normalize(u / r, dim=-1))
    p, nv = torch.cat(pts), torch.cat(nrms).contiguous()
    perm = torch.randperm(n, generator=g)                   # no surface-by-surface order in memory (a trained model has none)
    p, nv = p[perm], nv[perm]
    means = p + nv * (nrm(n, 1) * 0.003)
    # scales: (tangent 1, tangent 2, normal); the local z axis of the splat is the surface normal
    tang = torch.exp(nrm(n, 2) * 0.5 - 4.8)
    kind = rnd(n)
    needle, blob = kind < 0.02, (kind >= 0.02) & (kind < 0.025)
    tang[:, 0] = torch.where(needle, tang[:, 0] * 8.0, tang[:, 0])
    tang = torch.where(blob[:, None], tang * 6.0, tang)
    scales = torch.cat([tang, 0.15 * tang.min(dim=1, keepdim=True).values], dim=1)
    # rotation (wxyz): the shortest arc taking +z to the normal, after a random turn about z
    w_ = 1.0 + nv[:, 2]
    align = torch.stack([w_, -nv[:, 1], nv[:, 0], torch.zeros(n)], dim=1)
    align = torch.where((w_ < 1e-6)[:, None], torch.tensor([0.0, 1.0, 0.0, 0.0]).expand(n, 4), align)
    align = torch.nn.functional.normalize(align, dim=-1)
    th = rnd(n) * math.pi
    cz, sz = torch.cos(th), torch.sin(th)
    aw, ax, ay, az = align.unbind(dim=1)
    quats = torch.stack([aw * cz - az * sz, ax * cz + ay * sz, ay * cz - ax * sz, aw * sz + az * cz], dim=1)      # align (x) (cz, 0, 0, sz)
    quats = torch.nn.functional.normalize(quats, dim=-1)
    solid = rnd(n, 1) < 0.75
    opac = torch.sigmoid(torch.where(solid, nrm(n, 1) * 1.2 + 3.5, nrm(n, 1) - 2.0))
    freq = torch.tensor([[2.1, 1.3, 0.7], [0.9, 2.3, 1.7], [1.5, 0.6, 2.6]])
    base = 0.5 + 0.35 * torch.sin(p @ freq + torch.tensor([0.3, 1.1, 2.0]))
    shs = torch.empty(n, K, 3)
    shs[:, 0] = (base - 0.5) / 0.28209479177387814 + nrm(n, 3) * 0.05
    for deg in range(1, sh_degree + 1):
        shs[:, deg * deg:(deg + 1) * (deg + 1)] = nrm(n, 2 * deg + 1, 3) * (0.08 / deg)
    return means.contiguous(), scales.contiguous(), quats.contiguous(), opac, shs


def camera(width: int, height: int, fx: float, fy: float = None, distance: float = 4.0):
    """Identity rotation, translation (0,0,distance); matrices in the reference's transposed storage
    (internal/cameras/cameras.py:147-192)."""
    fy = fx if fy is None else fy
    w2c = torch.eye(4)
    w2c[3, 2] = distance
    znear, zfar = 0.01, 100.0
    tanx, tany = 0.5 * width / fx, 0.5 * height / fy
    P = torch.zeros(4, 4)
    P[0, 0], P[1, 1] = 1.0 / tanx, 1.0 / tany
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    full = w2c @ P.T
    center = torch.linalg.inv(w2c)[3, :3]
    return {"world_to_camera": w2c, "full_projection": full, "camera_center": center, "fx": fx, "fy": fy,
            "cx": width / 2.0, "cy": height / 2.0, "width": width, "height": height, "tanfovx": tanx, "tanfovy": tany}


def camera_looking_at_origin(width: int, height: int, fx: float, yaw: float, pitch: float, distance: float, fy: float = None):
    """Camera on a sphere of radius `distance` around the cloud's centre, looking at it (yaw / pitch in radians; 0, 0 is `camera()`:
    identity rotation, the camera at (0, 0, -distance)).  Same dictionary as `camera()`, matrices in the reference's transposed storage."""
    import math
    fy = fx if fy is None else fy
    c = torch.tensor([math.sin(yaw) * math.cos(pitch), math.sin(pitch), -math.cos(yaw) * math.cos(pitch)], dtype=torch.float64) * distance
    f = -c / c.norm()                                            # viewing direction (+z of the camera)
    r = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64), f)
    r = r / r.norm()                                             # +x of the camera
    d = torch.linalg.cross(f, r)                                 # +y of the camera (image rows grow along it)
    R = torch.stack([r, d, f])                                   # p_cam = R (p_world - c)
    w2c = torch.eye(4, dtype=torch.float64)
    w2c[:3, :3] = R.T                                            # row-vector convention: p_cam = p_world @ w2c[:3,:3] + w2c[3,:3]
    w2c[3, :3] = -(R @ c)
    w2c = w2c.float()
    znear, zfar = 0.01, 100.0
    tanx, tany = 0.5 * width / fx, 0.5 * height / fy
    P = torch.zeros(4, 4)
    P[0, 0], P[1, 1] = 1.0 / tanx, 1.0 / tany
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return {"world_to_camera": w2c, "full_projection": w2c @ P.T, "camera_center": torch.linalg.inv(w2c)[3, :3], "fx": fx, "fy": fy,
            "cx": width / 2.0, "cy": height / 2.0, "width": width, "height": height, "tanfovx": tanx, "tanfovy": tany}


def camera_set(width: int, height: int, fx: float, count: int = 16, distance: float = 4.0, kind: str = "heterogeneous"):
    """`count` training views of a synthetic scene, as a data loader would hand them out one per step (internal/dataset.py:146-184).
    View 0 is always `camera()` (the pose the workload's intersection count is quoted on).

    kind "heterogeneous" (default since round 6): views of very DIFFERENT footprint, as a capture has them — the others
        look at the cloud from within +-0.45 rad of yaw and +-0.25 rad of pitch at distances between 0.55 x and 1.75 x `distance`
        (2.2 ... 7 for the default 4), close-ups and far views alternating along the set.  At S-1080p-1M the rect intersections run
        from 6.7 M to 19.8 M (2.9 x), their mean 14.06 M — equal to the orbit set's (14.06 M) by construction of the exponent below,
        so that step times of the two sets compare at equal mean work.  Served in a fresh random permutation every epoch
        (`epoch_order`), as the reference's loader does (internal/dataset.py:216-217,258-259).
    kind "orbit" (rounds 3-5): +-12 % of the distance, list lengths within 22 % of each other, meant to be served in set order."""
    import math
    if kind not in ("heterogeneous", "orbit"):
        raise ValueError(f"camera_set kind must be heterogeneous | orbit, got {kind!r}")
    cams = [camera(width, height, fx, distance=distance)]
    for k in range(1, count):
        a = 2.0 * math.pi * k / count
        yaw = 0.45 * math.sin(a) + 0.08 * math.sin(3 * a)
        pitch = 0.25 * math.sin(2 * a + 0.7)
        if kind == "orbit":
            dist = distance * (1.0 + 0.12 * math.cos(5 * a + 0.3))
        else:
            u = ((k * 7) % (count - 1)) / max(count - 2, 1) if count > 2 else 0.5      # a permutation of 0 .. 1: neighbours far apart
            dist = distance * (0.55 + 1.2 * u ** 1.6)
        cams.append(camera_looking_at_origin(width, height, fx, yaw, pitch, dist))
    return cams


def epoch_order(n_views: int, epoch: int, seed: int = 42):
    """The order in which epoch `epoch` serves the `n_views` training views: a fresh `torch.randperm` per epoch from one seeded
    generator per loader, as `CacheDataLoader.__iter__` draws it (internal/dataset.py:216-217,258-259) — here a function of
    (seed, epoch), so that every rank of a job computes the same order without talking."""
    g = torch.Generator().manual_seed(int(seed) * 1_000_003 + int(epoch))
    return torch.randperm(int(n_views), generator=g).tolist()


class ViewStream:
    """Position k of the job's view stream -> index into the camera set.  `shuffled`: epoch e = positions [e n, (e + 1) n) in
    `epoch_order(n, e)`; otherwise set order, cyclically (rounds 3-5).  Rank r of W takes positions k W + r."""

    def __init__(self, n_views: int, shuffled: bool = True, seed: int = 42):
        self.n, self.shuffled, self.seed = int(n_views), bool(shuffled), int(seed)
        self._epoch, self._order = -1, None

    def view(self, position: int) -> int:
        e, i = divmod(int(position), self.n)
        if not self.shuffled:
            return i
        if e != self._epoch:
            self._epoch, self._order = e, epoch_order(self.n, e, self.seed)
        return self._order[i]


class CameraObject:
    """The fields of the reference's `Camera` (internal/cameras/cameras.py:13-43) the renderers read, on `device`."""

    def __init__(self, cam: dict, device, idx: int = 0):
        import math
        t = lambda v, dt=torch.float32: torch.tensor(v, dtype=dt, device=device)
        self.world_to_camera = cam["world_to_camera"].to(device)
        self.full_projection = cam["full_projection"].to(device)
        self.camera_center = cam["camera_center"].to(device)
        self.fx, self.fy, self.cx, self.cy = t(cam["fx"]), t(cam["fy"]), t(cam["cx"]), t(cam["cy"])
        self.width, self.height = t(cam["width"], torch.int32), t(cam["height"], torch.int32)
        self.fov_x, self.fov_y = t(2 * math.atan(cam["tanfovx"])), t(2 * math.atan(cam["tanfovy"]))
        self.idx = t(idx, torch.int32)
        self.device = device

    def to_device(self, device):
        for k, v in list(vars(self).items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        self.device = device
        return self


class ModelObject(torch.nn.Module):
    """Getter surface of the reference's Gaussian model (internal/models/gaussian.py:122-323) over ACTIVATED tensors
    (means, scales, unit quaternions, opacities, SH dc / rest), as `pre_activate_all_properties` leaves them."""

    def __init__(self, means, scales, quats, opac, shs, active_sh_degree: int = 3):
        super().__init__()
        P = torch.nn.Parameter
        self.means, self.scales_, self.rotations_, self.opacities_ = P(means), P(scales), P(quats), P(opac)
        self.shs_dc, self.shs_rest = P(shs[:, :1].contiguous()), P(shs[:, 1:].contiguous())
        self.active_sh_degree = active_sh_degree
        self.is_pre_activated = False

    get_xyz = property(lambda s: s.means)
    get_scaling = property(lambda s: s.scales_)
    get_rotation = property(lambda s: s.rotations_)
    get_opacity = property(lambda s: s.opacities_)
    get_features = property(lambda s: torch.cat((s.shs_dc, s.shs_rest), dim=1))

    def get_means(self): return self.means
    def get_scales(self): return self.scales_
    def get_rotations(self): return self.rotations_
    def get_opacities(self): return self.opacities_
    def get_shs_dc(self): return self.shs_dc
    def get_shs_rest(self): return self.shs_rest
    def leaves(self): return [self.means, self.scales_, self.rotations_, self.opacities_, self.shs_dc, self.shs_rest]
