"""bench_loop.py — the consumer side of the renderer boundary for `bench.py --loop reference-shaped` (bench infrastructure; the
package never imports it).

BASELINE.md §3 asks for images/s "from the unchanged Lightning loop".  Lightning, the reference's model and its density controller
are not on the GPU box (no reference tree there), so this module restates just enough of them to drive the renderer PLUGIN through
what that loop does to it — and times that:

  * `RawGaussians`        raw parameters (log scales, logit opacities, unnormalised quaternions, shs_dc / shs_rest) behind the
                          activated getters of `VanillaGaussianModel` (internal/models/vanilla_gaussian.py:341-364): exp / sigmoid /
                          normalize forward and backward every step — by torch around the renderer, as the reference has them
                          (`HipVanillaRenderer(fuse_activations=False)`), or inside its preprocess kernels (the plugin's default
                          for a model whose getters are exactly these: `renderer.model_raw_parameters`);
  * `DensityController`   `VanillaDensityControllerImpl` (internal/density_controllers/vanilla_density_controller.py:69-286):
                          `retain_grad` on the screen-space means, statistics after every backward (the package's fused kernel, as
                          `HipDensityStatsMixin` wires it), clone / split / prune every `densification_interval` steps with the
                          optimizer surgery of `density_controller.Utils` (:44-203), opacity reset — N changes between steps;
  * `run`                 the order of calls of `GaussianSplatting.training_step` (internal/gaussian_splatting.py:329-397): forward,
                          photometric loss (0.8 L1 + 0.2 (1 - SSIM), vanilla_metrics.py:57-70), before_backward, backward,
                          after_backward, optimizer step, SH-degree raise (vanilla_gaussian.py:333-339).

tests/test_bench_loop.py runs it beside the test-side restatement (oracle/training_oracle.py, itself pinned to the reference's real
classes on CPU) and compares the N trajectories.
"""
from __future__ import annotations

import math
import time
from typing import Dict, List, Optional

import torch
from torch import nn


class RawGaussians(nn.Module):
    NAMES = ("means", "shs_dc", "shs_rest", "opacities", "scales", "rotations")

    def __init__(self, means, scales, quats, opacities, shs, active_sh_degree: int = 0, max_sh_degree: int = 3):
        """Takes ACTIVATED values (scales > 0, opacities in (0, 1)) and stores the raw parameters."""
        super().__init__()
        P = lambda t: nn.Parameter(t.clone().contiguous().requires_grad_(True))
        o = opacities.reshape(-1, 1).clamp(1e-6, 1 - 1e-6)
        self.gaussians: Dict[str, torch.Tensor] = {
            "means": P(means), "shs_dc": P(shs[:, :1]), "shs_rest": P(shs[:, 1:]),
            "opacities": P(torch.log(o / (1 - o))), "scales": P(torch.log(scales)), "rotations": P(quats)}
        self.active_sh_degree, self.max_sh_degree = active_sh_degree, max_sh_degree
        self.is_pre_activated = False

    # what the reference's VanillaGaussianModel is recognised by (renderer.model_raw_parameters): getters = exp / normalize / sigmoid
    fused_activations = {"scales": "exp", "rotations": "normalize", "opacities": "sigmoid"}
    properties = property(lambda s: s.gaussians)
    n_gaussians = property(lambda s: s.gaussians["means"].shape[0])

    def get_property(self, name):
        return self.gaussians[name]

    # activated getters (vanilla_gaussian.py:341-364), both spellings the renderers use
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

    def make_optimizers(self, spatial_lr_scale: float, cls, **kw) -> List[torch.optim.Optimizer]:
        """The two optimizers and learning rates of `VanillaGaussianModel.training_setup` (vanilla_gaussian.py:266-330)."""
        g = self.gaussians
        return [cls([{"params": [g["means"]], "name": "means"}], lr=0.00016 * spatial_lr_scale, eps=1e-15, **kw),
                cls([{"params": [g["shs_dc"]], "lr": 0.0025, "name": "shs_dc"},
                     {"params": [g["shs_rest"]], "lr": 0.0025 / 20.0, "name": "shs_rest"},
                     {"params": [g["scales"]], "lr": 0.005, "name": "scales"},
                     {"params": [g["rotations"]], "lr": 0.001, "name": "rotations"},
                     {"params": [g["opacities"]], "lr": 0.05, "name": "opacities"}], lr=0.0, eps=1e-15, **kw)]


def _rotation_matrices(r):
    """internal/utils/general_utils.py:142-163."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def _rebuild(optimizers, rows_of, moments_of, only=None) -> Dict[str, nn.Parameter]:
    """`density_controller.Utils` (:44-203): every named group gets a fresh Parameter (rows_of(name, old)) and its Adam moments follow
    (moments_of(name, old moment, new parameter)); the step counter stays."""
    out = {}
    for opt in optimizers:
        for group in opt.param_groups:
            name = group["name"]
            if only is not None and name not in only:
                continue
            old = group["params"][0]
            new = nn.Parameter(rows_of(name, old).requires_grad_(True))
            state = opt.state.pop(old, None)
            if state is not None:
                for k in ("exp_avg", "exp_avg_sq"):
                    state[k] = moments_of(name, state[k], new)
                opt.state[new] = state
            group["params"][0] = new
            out[name] = new
    return out


class DensityController:
    def __init__(self, n: int, device, cameras_extent: float, *, percent_dense=0.01, densification_interval=100, opacity_reset_interval=3000,
                 opacity_reset_value=0.01, densify_from_iter=500, densify_until_iter=15_000, densify_grad_threshold=0.0002,
                 cull_opacity_threshold=0.005, fused_stats=True):
        self.c = dict(percent_dense=percent_dense, densification_interval=densification_interval, opacity_reset_interval=opacity_reset_interval,
                      opacity_reset_value=opacity_reset_value, densify_from_iter=densify_from_iter, densify_until_iter=densify_until_iter,
                      densify_grad_threshold=densify_grad_threshold, cull_opacity_threshold=cull_opacity_threshold)
        self.extent = cameras_extent
        self.fused_stats = fused_stats
        self.events: List[dict] = []
        self._fresh(n, device)

    def _fresh(self, n, device):
        self.max_radii2D = torch.zeros((n,), device=device)
        self.xyz_gradient_accum = torch.zeros((n, 1), device=device)
        self.denom = torch.zeros((n, 1), device=device)

    def before_backward(self, outputs, step):                                       # :69-76
        self._stats_request = None
        if step < self.c["densify_until_iter"]:
            outputs["viewspace_points"].retain_grad()
            if self.fused_stats and outputs.get("viewspace_points_grad_scale", None) is None \
                    and getattr(outputs["visibility_filter"], "_gspl_radii_positive", False):
                # HipDensityStatsMixin.before_backward: the frame's backward applies the statistics itself
                from gspl_amd.density import request_stats_in_backward
                self._stats_request = request_stats_in_backward(outputs["radii"], self.xyz_gradient_accum, self.denom, self.max_radii2D)

    @torch.no_grad()
    def after_backward(self, outputs, model, optimizers, step):                     # :78-99
        if step >= self.c["densify_until_iter"]:
            return
        self.update_states(outputs)
        if step > self.c["densify_from_iter"] and step % self.c["densification_interval"] == 0:
            before = model.n_gaussians
            self._densify_and_prune(20 if step > self.c["opacity_reset_interval"] else None, model, optimizers)
            self.events.append({"step": step, "n_before": before, "n_after": model.n_gaussians})
        if step % self.c["opacity_reset_interval"] == 0:
            op = model.get_opacities()
            o = torch.min(op, torch.ones_like(op) * self.c["opacity_reset_value"])
            model.gaussians.update(_rebuild(optimizers, lambda n, p: torch.log(o / (1 - o)), lambda n, s, new: torch.zeros_like(new), {"opacities"}))
            self.events.append({"step": step, "opacity_reset": True})

    def update_states(self, outputs):                                               # :101-123
        vp, vis, radii = outputs["viewspace_points"], outputs["visibility_filter"], outputs["radii"]
        scale = outputs.get("viewspace_points_grad_scale", None)
        if self.fused_stats:                                                        # HipDensityStatsMixin: one launch, or none
            from gspl_amd.density import update_densification_stats, withdraw_stats_request
            req, self._stats_request = getattr(self, "_stats_request", None), None
            withdraw_stats_request(req)
            if req is not None and req.applied:
                return
            update_densification_stats(vp.grad, vis, radii, self.xyz_gradient_accum, self.denom, self.max_radii2D, scale=scale)
            return
        self.max_radii2D[vis] = torch.max(self.max_radii2D[vis], radii[vis].float())
        g = vp.grad[vis, :2]
        self.xyz_gradient_accum[vis] += torch.norm(g if scale is None else g * scale, dim=-1, keepdim=True)
        self.denom[vis] += 1

    def _append(self, new_rows, model, optimizers):                                 # :255-260
        model.gaussians.update(_rebuild(optimizers, lambda n, p: torch.cat((p, new_rows[n]), dim=0),
                                        lambda n, s, new: torch.cat((s, torch.zeros_like(new_rows[n])), dim=0)))
        self._fresh(model.n_gaussians, model.get_property("means").device)

    def _prune(self, mask, model, optimizers):                                      # :262-276
        keep = ~mask
        model.gaussians.update(_rebuild(optimizers, lambda n, p: p[keep], lambda n, s, new: s[keep]))
        self.xyz_gradient_accum, self.denom, self.max_radii2D = self.xyz_gradient_accum[keep], self.denom[keep], self.max_radii2D[keep]

    def _densify_and_prune(self, max_screen_size, model, optimizers):               # :125-152
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        thr, dense = self.c["densify_grad_threshold"], self.c["percent_dense"] * self.extent
        # clone (:154-174): small splats with a large mean gradient are duplicated in place
        sel = torch.logical_and(torch.norm(grads, dim=-1) >= thr, model.get_scales().max(dim=1).values <= dense)
        self._append({k: v[sel] for k, v in model.properties.items()}, model, optimizers)
        # split (:176-253): large ones are replaced by two samples of themselves, 1.6 times smaller
        n_now, scales = model.n_gaussians, model.get_scales()
        padded = torch.zeros((n_now,), device=scales.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = torch.logical_and(padded >= thr, scales.max(dim=1).values > dense)
        stds = scales[sel].repeat(2, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=scales.device), std=stds)
        rots = _rotation_matrices(model.get_property("rotations")[sel]).repeat(2, 1, 1)
        new = {"means": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + model.get_means()[sel].repeat(2, 1),
               "scales": torch.log(scales[sel].repeat(2, 1) / (0.8 * 2))}
        for key, value in model.properties.items():
            if key not in new:
                new[key] = value[sel].repeat(2, *[1 for _ in range(value.dim() - 1)])
        self._append(new, model, optimizers)
        self._prune(torch.cat((sel, torch.zeros(2 * int(sel.sum()), device=sel.device, dtype=torch.bool))), model, optimizers)
        # prune (:139-152)
        mask = (model.get_opacities() < self.c["cull_opacity_threshold"]).squeeze()
        if max_screen_size:
            mask = mask | (self.max_radii2D > max_screen_size) | (model.get_scales().max(dim=1).values > 0.1 * self.extent)
        self._prune(mask, model, optimizers)


def perturbed(params, seed=7):
    """A model that is NOT yet the scene its targets were rendered from (what training sees): positions off by a few per cent of
    the splat spacing, sizes and opacities off by ~10 %, colours washed out."""
    means, scales, quats, opac, shs = params
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return (means + 0.01 * r(*means.shape), scales * torch.exp(0.1 * r(*scales.shape)), quats + 0.02 * r(*quats.shape),
            (opac * torch.exp(0.1 * r(*opac.shape))).clamp(0.02, 0.98), shs * 0.8)


def run(renderer, model: RawGaussians, controller: DensityController, optimizers, cameras: list, targets: list, steps: int, background,
        loss_fn, sh_degree_up_interval: int = 1000, on_step=None, view_stream=None) -> dict:
    """The training loop; every step leaves a device event, so the per-step spans come out without a host synchronisation inside
    the loop.  Returns the history and the timing.  `view_stream` (synthetic.ViewStream): the order the views are served in — a fresh
    permutation per epoch, as the reference's loader (internal/dataset.py:216-217); None: set order, cyclically."""
    marks, n_hist, losses = [], [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(1, steps + 1):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        k = view_stream.view(step - 1) if view_stream is not None else (step - 1) % len(cameras)
        outputs = renderer(cameras[k], model, background)
        loss = loss_fn(outputs["render"], targets[k])
        controller.before_backward(outputs, step)
        loss.backward()
        controller.after_backward(outputs, model, optimizers, step)
        for opt in optimizers:
            opt.step()
            opt.zero_grad(set_to_none=True)
        if step % sh_degree_up_interval == 0 and model.active_sh_degree < model.max_sh_degree:
            model.active_sh_degree += 1
        n_hist.append(model.n_gaussians)
        losses.append(loss.detach())
        if on_step is not None:
            on_step(step, outputs)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append(e)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    spans = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    return {"elapsed_s": elapsed, "step_ms": spans, "n": n_hist, "loss": [float(l) for l in losses]}
