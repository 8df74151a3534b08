"""TEST INFRASTRUCTURE ONLY (never imported by the product).

A minimal restatement of what sits on the other side of the renderer boundary during training, so that the renderer
plugins can be driven through densify / prune / opacity-reset steps (N changes between steps) without Lightning:

  * `TrainableGaussians`       — the parameter container and getters of the reference's `VanillaGaussianModel`
                                 (internal/models/vanilla_gaussian.py:66-460, internal/models/gaussian.py:122-323): raw
                                 parameters (log scales, logit opacities, unnormalised quaternions), activated getters.
  * `DensityControllerOracle`  — `VanillaDensityControllerImpl` (internal/density_controllers/vanilla_density_controller.py:
                                 41-300) with the optimizer surgery of `density_controller.Utils` (:36-203) restated.
  * `train`                    — the order of calls of `GaussianSplatting.training_step`
                                 (internal/gaussian_splatting.py:329-397): forward, loss, before_backward, backward,
                                 after_backward (statistics, densify / prune, opacity reset), optimizer step.

Pinned by tests/test_training_loop.py: with the reference tree importable (lightning stubbed) the same loop runs on the
reference's real `VanillaGaussianModel` + `VanillaDensityControllerImpl` and must make the same decisions (same N after every
step, same tensors).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import torch
from torch import nn


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class TrainableGaussians(nn.Module):
    NAMES = ("means", "shs_dc", "shs_rest", "opacities", "scales", "rotations")

    def __init__(self, means, scales, quats, opacities, shs, active_sh_degree: int = 0, max_sh_degree: int = 3):
        """Activated inputs (scales > 0, opacities in (0,1), shs [N,K,3]) are stored as raw parameters."""
        super().__init__()
        P = lambda t: nn.Parameter(t.clone().contiguous().requires_grad_(True))
        self.gaussians = {
            "means": P(means), "shs_dc": P(shs[:, :1]), "shs_rest": P(shs[:, 1:]),
            "opacities": P(inverse_sigmoid(opacities.reshape(-1, 1))), "scales": P(torch.log(scales)), "rotations": P(quats),
        }
        self.active_sh_degree = active_sh_degree
        self.max_sh_degree = max_sh_degree
        self.is_pre_activated = False

    # --- container API (gaussian.py:122-190)
    @property
    def properties(self) -> Dict[str, torch.Tensor]:
        return self.gaussians

    @properties.setter
    def properties(self, new: Dict[str, torch.Tensor]):
        self.gaussians = dict(new)

    def update_properties(self, new: Dict[str, torch.Tensor]):
        self.gaussians.update(new)

    property_names = property(lambda s: s.NAMES)

    def get_property_names(self):
        return self.NAMES

    def get_property(self, name):
        return self.gaussians[name]

    n_gaussians = property(lambda s: s.gaussians["means"].shape[0])

    # --- activations (vanilla_gaussian.py:341-364)
    scale_inverse_activation = staticmethod(torch.log)
    opacity_inverse_activation = staticmethod(inverse_sigmoid)

    get_xyz = property(lambda s: s.gaussians["means"])
    get_scaling = property(lambda s: torch.exp(s.gaussians["scales"]))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s.gaussians["rotations"]))
    get_opacity = property(lambda s: torch.sigmoid(s.gaussians["opacities"]))
    get_features = property(lambda s: torch.cat((s.gaussians["shs_dc"], s.gaussians["shs_rest"]), dim=1))

    def get_means(self): return self.gaussians["means"]
    def get_scales(self): return torch.exp(self.gaussians["scales"])
    def get_rotations(self): return torch.nn.functional.normalize(self.gaussians["rotations"])
    def get_opacities(self): return torch.sigmoid(self.gaussians["opacities"])
    def get_shs_dc(self): return self.gaussians["shs_dc"]
    def get_shs_rest(self): return self.gaussians["shs_rest"]

    def make_optimizers(self, spatial_lr_scale: float, cls=torch.optim.Adam, **kw) -> List[torch.optim.Optimizer]:
        """The two optimizers of `VanillaGaussianModel.training_setup` (vanilla_gaussian.py:266-330): means; everything else."""
        g = self.gaussians
        means_opt = cls([{"params": [g["means"]], "name": "means"}], lr=0.00016 * spatial_lr_scale, eps=1e-15, **kw)
        rest = cls([
            {"params": [g["shs_dc"]], "lr": 0.0025, "name": "shs_dc"},
            {"params": [g["shs_rest"]], "lr": 0.0025 / 20.0, "name": "shs_rest"},
            {"params": [g["scales"]], "lr": 0.005, "name": "scales"},
            {"params": [g["rotations"]], "lr": 0.001, "name": "rotations"},
            {"params": [g["opacities"]], "lr": 0.05, "name": "opacities"},
        ], lr=0.0, eps=1e-15, **kw)
        return [means_opt, rest]


def build_rotation(r):
    """internal/utils/general_utils.py:142-163 (normalises, then the standard wxyz matrix)."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.size(0), 3, 3), device=r.device, dtype=r.dtype)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


# ---- density_controller.Utils restated (density_controller.py:44-203) -------------------------------------------------------
def _surgery(optimizers, fn_param: Callable, fn_state: Callable, only: Optional[set] = None) -> Dict[str, nn.Parameter]:
    out = {}
    for opt in optimizers:
        for group in opt.param_groups:
            if only is not None and group["name"] not in only:
                continue
            assert len(group["params"]) == 1
            old = group["params"][0]
            state = opt.state.get(old, None)
            new = nn.Parameter(fn_param(group["name"], old).requires_grad_(True))
            if state is not None:
                state["exp_avg"], state["exp_avg_sq"] = fn_state(group["name"], state["exp_avg"], new), fn_state(group["name"], state["exp_avg_sq"], new)
                del opt.state[old]
                opt.state[new] = state
            group["params"][0] = new
            out[group["name"]] = new
    return out


def cat_tensors_to_properties(new_properties, model, optimizers):
    return _surgery(optimizers, lambda n, p: torch.cat((p, new_properties[n]), dim=0),
                    lambda n, s, new: torch.cat((s, torch.zeros_like(new_properties[n])), dim=0))


def prune_properties(keep_mask, model, optimizers):
    return _surgery(optimizers, lambda n, p: p[keep_mask], lambda n, s, new: s[keep_mask])


def replace_tensors_to_properties(tensors, optimizers):
    return _surgery(optimizers, lambda n, p: tensors[n], lambda n, s, new: torch.zeros_like(new), only=set(tensors))


class DensityControllerOracle:
    def __init__(self, n_gaussians: int, device, cameras_extent: float, prune_extent: Optional[float] = None, *,
                 percent_dense=0.01, densification_interval=100, opacity_reset_interval=3000, opacity_reset_value=0.01,
                 densify_from_iter=500, densify_until_iter=15_000, densify_grad_threshold=0.0002, cull_opacity_threshold=0.005,
                 absgrad=False):
        self.c = dict(percent_dense=percent_dense, densification_interval=densification_interval, opacity_reset_interval=opacity_reset_interval,
                      opacity_reset_value=opacity_reset_value, densify_from_iter=densify_from_iter, densify_until_iter=densify_until_iter,
                      densify_grad_threshold=densify_grad_threshold, cull_opacity_threshold=cull_opacity_threshold, absgrad=absgrad)
        self.cameras_extent = cameras_extent
        self.prune_extent = cameras_extent if prune_extent is None else prune_extent
        self._init_state(n_gaussians, device)

    def _init_state(self, n, device):
        self.max_radii2D = torch.zeros((n,), device=device)
        self.xyz_gradient_accum = torch.zeros((n, 1), device=device)
        self.denom = torch.zeros((n, 1), device=device)

    def before_backward(self, outputs, global_step):                                            # :69-76
        if global_step >= self.c["densify_until_iter"]:
            return
        outputs["viewspace_points"].retain_grad()

    def after_backward(self, outputs, model, optimizers, global_step, white_background=False):  # :78-99
        if global_step >= self.c["densify_until_iter"]:
            return
        with torch.no_grad():
            self.update_states(outputs)
            if global_step > self.c["densify_from_iter"] and global_step % self.c["densification_interval"] == 0:
                size_threshold = 20 if global_step > self.c["opacity_reset_interval"] else None
                self._densify_and_prune(size_threshold, model, optimizers)
            if global_step % self.c["opacity_reset_interval"] == 0 or (white_background and global_step == self.c["densify_from_iter"]):
                self._reset_opacities(model, optimizers)

    def update_states(self, outputs):                                                           # :101-123
        vp, vis, radii = outputs["viewspace_points"], outputs["visibility_filter"], outputs["radii"]
        scale = outputs.get("viewspace_points_grad_scale", None)
        self.max_radii2D[vis] = torch.max(self.max_radii2D[vis], radii[vis])
        grad = vp.absgrad if self.c["absgrad"] is True else vp.grad
        g = grad[vis, :2]
        if scale is not None:
            g = g * scale
        self.xyz_gradient_accum[vis] += torch.norm(g, dim=-1, keepdim=True)
        self.denom[vis] += 1

    def _densify_and_prune(self, max_screen_size, model, optimizers):                           # :125-152
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self._densify_and_clone(grads, model, optimizers)
        self._densify_and_split(grads, model, optimizers)
        prune_mask = (model.get_opacities() < self.c["cull_opacity_threshold"]).squeeze()
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = model.get_scales().max(dim=1).values > 0.1 * self.prune_extent
            prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_vs), big_ws)
        self._prune_points(prune_mask, model, optimizers)

    def _densify_and_clone(self, grads, model, optimizers):                                     # :154-174
        sel = torch.where(torch.norm(grads, dim=-1) >= self.c["densify_grad_threshold"], True, False)
        sel = torch.logical_and(sel, torch.max(model.get_scales(), dim=1).values <= self.c["percent_dense"] * self.cameras_extent)
        self._densification_postfix({k: v[sel] for k, v in model.properties.items()}, model, optimizers)

    def _densify_and_split(self, grads, model, optimizers, N=2):                                # :176-253
        device = model.get_property("means").device
        n_init = model.n_gaussians
        scales = model.get_scales()
        padded = torch.zeros((n_init,), device=device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = torch.where(padded >= self.c["densify_grad_threshold"], True, False)
        sel = torch.logical_and(sel, torch.max(scales, dim=1).values > self.c["percent_dense"] * self.cameras_extent)
        stds = scales[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=device), std=stds)
        rots = build_rotation(model.get_property("rotations")[sel]).repeat(N, 1, 1)
        new = {"means": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + model.get_means()[sel].repeat(N, 1),
               "scales": model.scale_inverse_activation(scales[sel].repeat(N, 1) / (0.8 * N))}
        for key, value in model.properties.items():
            if key not in new:
                new[key] = value[sel].repeat(N, *[1 for _ in range(value[sel].dim() - 1)])
        self._densification_postfix(new, model, optimizers)
        prune = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=device, dtype=torch.bool)))
        self._prune_points(prune, model, optimizers)

    def _densification_postfix(self, new_properties, model, optimizers):                        # :255-260
        model.properties = cat_tensors_to_properties(new_properties, model, optimizers)
        self._init_state(model.n_gaussians, model.get_property("means").device)

    def _prune_points(self, mask, model, optimizers):                                           # :262-276
        keep = ~mask
        model.properties = prune_properties(keep, model, optimizers)
        self.xyz_gradient_accum, self.denom, self.max_radii2D = self.xyz_gradient_accum[keep], self.denom[keep], self.max_radii2D[keep]

    def _reset_opacities(self, model, optimizers):                                              # :278-286
        op = model.get_opacities()
        new = model.opacity_inverse_activation(torch.min(op, torch.ones_like(op) * self.c["opacity_reset_value"]))
        model.update_properties(replace_tensors_to_properties({"opacities": new}, optimizers))


def psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a.double() - b.double()) ** 2)))


def train(model, controller, optimizers, render: Callable, cameras: list, targets: list, steps: int, background,
          sh_degree_up_interval: int = 1000, max_sh_degree: int = 3, on_step: Optional[Callable] = None,
          controller_is_reference: bool = False, pl_module=None):
    """`render(camera, model, background) -> outputs dict` (a renderer plugin).  Returns the per-step (loss, N) history."""
    history = []
    for global_step in range(1, steps + 1):
        k = (global_step - 1) % len(cameras)
        outputs = render(cameras[k], model, background)
        loss = (outputs["render"] - targets[k]).abs().mean()                  # L1 (lambda_dssim = 0)
        if controller_is_reference:
            controller.before_backward(outputs, None, model, optimizers, global_step, pl_module)
        else:
            controller.before_backward(outputs, global_step)
        loss.backward()
        if controller_is_reference:
            controller.after_backward(outputs, None, model, optimizers, global_step, pl_module)
        else:
            controller.after_backward(outputs, model, optimizers, global_step)
        for opt in optimizers:
            opt.step()
            opt.zero_grad(set_to_none=True)
        if global_step % sh_degree_up_interval == 0 and model.active_sh_degree < max_sh_degree:   # vanilla_gaussian.py:333-339
            model.active_sh_degree = model.active_sh_degree + 1
        history.append((float(loss.detach()), int(model.n_gaussians)))
        if on_step is not None:
            on_step(global_step, outputs)
    return history
