"""Duck-typed stand-ins for the reference's Camera (internal/cameras/cameras.py:13-43) and
VanillaGaussianModel getters (internal/models/gaussian.py:122-323) — enough surface for the renderers."""
import math

import torch


class FakeCamera:
    def __init__(self, cam: dict, device):
        t = lambda v, dt=torch.float32: torch.tensor(v, dtype=dt, device=device)
        self.world_to_camera = cam["world_to_camera"].to(device)
        self.full_projection = cam["full_projection"].to(device)
        self.camera_center = cam["camera_center"].to(device)
        self.fx, self.fy, self.cx, self.cy = t(cam["fx"]), t(cam["fy"]), t(cam["cx"]), t(cam["cy"])
        self.width, self.height = t(cam["width"], torch.int32), t(cam["height"], torch.int32)
        self.fov_x = t(2 * math.atan(cam["tanfovx"]))
        self.fov_y = t(2 * math.atan(cam["tanfovy"]))
        self.idx = t(int(cam.get("idx", 0)), torch.int32)
        self.device = device

    def to_device(self, device):
        for k, v in list(vars(self).items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        self.device = device
        return self


class FakeGaussianModel(torch.nn.Module):
    """Stores activated values directly (like `pre_activate_all_properties`), with shs split dc/rest."""

    def __init__(self, means, scales, quats, opac, shs, active_sh_degree=3):
        super().__init__()
        P = torch.nn.Parameter
        self.means, self.scales_, self.rotations_, self.opacities_ = P(means), P(scales), P(quats), P(opac)
        self.shs_dc, self.shs_rest = P(shs[:, :1].contiguous()), P(shs[:, 1:].contiguous())
        self.active_sh_degree = active_sh_degree
        self.max_sh_degree = int(math.isqrt(shs.shape[1])) - 1
        self.is_pre_activated = False

    # vanilla getters
    get_xyz = property(lambda s: s.means)
    get_scaling = property(lambda s: s.scales_)
    get_rotation = property(lambda s: s.rotations_)
    get_opacity = property(lambda s: s.opacities_)
    get_features = property(lambda s: torch.cat((s.shs_dc, s.shs_rest), dim=1))

    # v1-style getters
    def get_means(self): return self.means
    def get_scales(self): return self.scales_
    def get_rotations(self): return self.rotations_
    def get_opacities(self): return self.opacities_
    def get_shs_dc(self): return self.shs_dc
    def get_shs_rest(self): return self.shs_rest
    def leaves(self): return [self.means, self.scales_, self.rotations_, self.opacities_, self.shs_dc, self.shs_rest]


class FakePropertyModel(torch.nn.Module):
    """Model with a `properties` dict the way internal/models/gaussian.py:122-190 keeps one (name -> Parameter), storing
    activated values; what `training_setup` / `random_redistribute` of the sharded renderer and the density controllers'
    optimizer surgery operate on."""

    NAMES = ("means", "shs_dc", "shs_rest", "opacities", "scales", "rotations")

    def __init__(self, means, scales, quats, opac, shs, active_sh_degree=3, extra=None):
        super().__init__()
        P = torch.nn.Parameter
        self._props = {"means": P(means), "shs_dc": P(shs[:, :1].contiguous()), "shs_rest": P(shs[:, 1:].contiguous()),
                       "opacities": P(opac), "scales": P(scales), "rotations": P(quats)}
        for k, v in (extra or {}).items():
            self._props[k] = P(v, requires_grad=False)
        self.active_sh_degree = active_sh_degree
        self.max_sh_degree = int(math.isqrt(shs.shape[1])) - 1
        self.is_pre_activated = False

    @property
    def properties(self):
        return self._props

    @properties.setter
    def properties(self, new):
        self._props = dict(new)

    @property
    def n_gaussians(self):
        return self._props["means"].shape[0]

    def get_property_names(self):
        return tuple(self._props.keys())

    def get_property(self, name):
        return self._props[name]

    get_xyz = property(lambda s: s._props["means"])
    get_scaling = property(lambda s: s._props["scales"])
    get_rotation = property(lambda s: s._props["rotations"])
    get_opacity = property(lambda s: s._props["opacities"])
    get_features = property(lambda s: torch.cat((s._props["shs_dc"], s._props["shs_rest"]), dim=1))

    def get_means(self): return self._props["means"]
    def get_scales(self): return self._props["scales"]
    def get_rotations(self): return self._props["rotations"]
    def get_opacities(self): return self._props["opacities"]
    def get_shs_dc(self): return self._props["shs_dc"]
    def get_shs_rest(self): return self._props["shs_rest"]

    def named_optimizer(self, lrs=None, cls=torch.optim.Adam, **kw):
        lrs = lrs or {}
        groups = [{"params": [self._props[n]], "lr": lrs.get(n, 1e-3), "name": n} for n in self.NAMES]
        return cls(groups, **kw)
