"""Renderer plugin base.

Inside the reference repository (yzslab/gaussian-splatting-lightning on PYTHONPATH) the classes are the
reference's own (`internal/renderers/renderer.py:43-117`), so `isinstance(renderer, Renderer)` checks in
`internal/gaussian_splatting.py:75-77` hold.  Stand-alone (tests, bench, this container, where
`lightning` is not installed) an interface-identical stub is used.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict

import torch

try:  # pragma: no cover - only inside the reference repo
    from internal.renderers.renderer import (  # type: ignore
        Renderer, RendererConfig, RendererOutputInfo, RendererOutputTypes)
    INSIDE_REFERENCE = True
except Exception:  # ImportError, or lightning missing
    INSIDE_REFERENCE = False

    class RendererOutputTypes:
        RGB: int = 1
        GRAY: int = 2
        NORMAL_MAP: int = 3
        FEATURE_MAP: int = 4
        OTHER: int = 65535

    @dataclass
    class RendererOutputInfo:
        key: str
        type: int = RendererOutputTypes.RGB
        visualizer: Callable = None

        def __post_init__(self):
            if self.type == RendererOutputTypes.OTHER and self.visualizer is None:
                raise ValueError("Visualizer must be provided when `type` is `OTHER`")

    class Renderer(torch.nn.Module):
        """Same surface as the reference's `Renderer` (renderer.py:43-111)."""

        def forward(self, viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, render_types: list = None, **kwargs):
            pass

        def training_forward(self, step: int, module, viewpoint_camera, pc, bg_color: torch.Tensor, render_types: list = None, **kwargs):
            return self(viewpoint_camera=viewpoint_camera, pc=pc, bg_color=bg_color, render_types=render_types, **kwargs)

        def before_training_step(self, step: int, module):
            return

        def after_training_step(self, step: int, module):
            return

        def setup(self, stage: str, *args: Any, **kwargs: Any) -> Any:
            pass

        def training_setup(self, module):
            return None, None

        def on_load_checkpoint(self, module, checkpoint):
            pass

        def setup_web_viewer_tabs(self, viewer, server, tabs):
            pass

        def get_available_outputs(self) -> Dict[str, RendererOutputInfo]:
            return {"rgb": RendererOutputInfo("render")}

    @dataclass
    class RendererConfig:
        def instantiate(self, *args, **kwargs) -> Renderer:
            raise NotImplementedError()


def camera_scalars(viewpoint_camera, names):
    """Python numbers of the camera's 0-d tensor fields `names` (width, height, fov_x, idx, ...).  The reference reads each of
    them with `.item()` on every call (gsplat_renderer.py:61-62,71-74, vanilla_renderer.py:59-60): a device read-back drains
    the queue — the whole previous step — before the new step's first kernel can be enqueued.  Here a field is read ONCE per
    camera object (all missing fields in one transfer) and kept on the object, keyed on the tensor's identity and version
    counter, so that a field replaced or modified in place is read again."""
    cache = getattr(viewpoint_camera, "_gspl_scalars", None)
    if cache is None:
        cache = {}
        try:
            viewpoint_camera._gspl_scalars = cache
        except AttributeError:          # an object that takes no attributes: read every time
            pass
    out, missing = {}, []
    for n in names:
        v = getattr(viewpoint_camera, n)
        hit = cache.get(n)
        if isinstance(v, torch.Tensor):
            if hit is not None and hit[0] is v and hit[1] == v._version:
                out[n] = hit[2]
            else:
                missing.append((n, v))
        else:
            out[n] = v
    if missing:
        on_dev = [m for m in missing if m[1].is_cuda]
        if len(on_dev) > 1:
            vals = torch.stack([v.detach().reshape(()).double() for _, v in on_dev]).tolist()      # one read-back for all of them
        else:
            vals = [v.item() for _, v in on_dev]
        got = {n: (int(round(x)) if not v.is_floating_point() else float(x)) for (n, v), x in zip(on_dev, vals)}
        for n, v in missing:
            val = got[n] if n in got else v.item()
            cache[n] = (v, v._version, val)
            out[n] = val
    return tuple(out[n] for n in names)


def camera_hw(viewpoint_camera):
    """(width, height) as python ints (read back once per camera object: `camera_scalars`)."""
    w, h = camera_scalars(viewpoint_camera, ("width", "height"))
    return int(w), int(h)


_GRAD_SCALES: dict = {}


def viewspace_grad_scale(W: int, H: int, like: torch.Tensor) -> torch.Tensor:
    """0.5 * [[W, H]] on `like`'s device/dtype (what the reference builds every call with
    `0.5 * torch.tensor([[W, H]]).to(xys)`, gsplat_renderer.py:196): cached, so that a step does not pay a
    host->device copy for a constant."""
    k = (W, H, like.device, like.dtype)
    t = _GRAD_SCALES.get(k)
    if t is None:
        if len(_GRAD_SCALES) > 256:
            _GRAD_SCALES.clear()
        t = _GRAD_SCALES[k] = (0.5 * torch.tensor([[W, H]])).to(like)
    return t


def model_sh_pair(pc):
    """(shs_dc, shs_rest) — the model's two SH parameters as they are stored (internal/models/gaussian.py:218-254) — when the model
    has them, else (`get_features`, None): a pre-activated model keeps one "shs" tensor (vanilla_gaussian.py:370-390).  The kernels
    read either form in place; `get_features` on a two-parameter model is a `torch.cat` of 192 B per Gaussian every step."""
    if not getattr(pc, "is_pre_activated", False) and hasattr(pc, "get_shs_dc") and hasattr(pc, "get_shs_rest"):
        try:
            return pc.get_shs_dc(), pc.get_shs_rest()
        except KeyError:
            pass
    # `get_features` reads the parameters through torch (a cat, or the stored tensor of a pre-activated model): an update of the
    # coefficients still in flight on the colour stream (FusedAdam(deferred=...)) has to land first
    from .. import ops
    ops.join_pending_updates(pc.get_xyz.device)
    return pc.get_features, None


_REFERENCE_ACTIVATIONS = {"scale_activation": "exp", "rotation_activation": "normalize", "opacity_activation": "sigmoid"}


def _defined_by_reference_vanilla_model(fn, name: str) -> bool:
    """`fn` is the function object `VanillaGaussianModel` of the reference defines (internal/models/vanilla_gaussian.py) — not an
    override of a subclass, not the identity a pre-activated model installs on the instance."""
    fn = getattr(fn, "__func__", fn)
    module = getattr(fn, "__module__", "") or ""
    return (module == "internal.models.vanilla_gaussian" or module.endswith(".internal.models.vanilla_gaussian")) and \
        (getattr(fn, "__qualname__", "") or "") == "VanillaGaussianModel." + name


def model_raw_parameters(pc):
    """(raw scales, raw rotations, raw opacities) when the model's activated getters are exactly exp / F.normalize / sigmoid of
    parameters it stores — then the rasterizer applies them inside its preprocess kernels (`raw_parameters=True`) and the step
    loses the ten elementwise launches and the reduction that `get_scaling` / `get_rotation` / `get_opacity` and their autograd
    backward cost (profiles/r05f_loop_sequence_torch_activations.txt: ~0.17 ms of a 1.48 ms step at 1 M Gaussians) — else None (any other model:
    the getters are called, as the reference's renderer does, vanilla_renderer.py:62-77).  Two ways to qualify:
      * the model DECLARES it: `fused_activations = {"scales": "exp", "rotations": "normalize", "opacities": "sigmoid"}` next to
        `get_property(name)` returning the stored tensors (this package's models, bench_loop.RawGaussians);
      * it is the reference's `VanillaGaussianModel` with its own activation methods and vanilla getters untouched
        (vanilla_gaussian.py:345-358, 421-441) and not pre-activated; a subclass that overrides one of them (MipSplatting's
        filtered scales, a glossy model's opacity, ...) does not qualify."""
    if getattr(pc, "is_pre_activated", False):
        return None
    declared = getattr(pc, "fused_activations", None)
    if declared is not None:
        if dict(declared) != {"scales": "exp", "rotations": "normalize", "opacities": "sigmoid"}:
            return None
    else:
        cls = type(pc)
        for name in _REFERENCE_ACTIVATIONS:
            if not _defined_by_reference_vanilla_model(getattr(pc, name, None), name):
                return None
        for name in ("get_scaling", "get_rotation", "get_opacity"):
            prop = getattr(cls, name, None)
            if not isinstance(prop, property) or not _defined_by_reference_vanilla_model(prop.fget, name):
                return None
    try:
        return pc.get_property("scales"), pc.get_property("rotations"), pc.get_property("opacities")
    except (KeyError, AttributeError):
        return None


_TILE_NOTE = set()


def implementation_tile_size(block_size: int) -> int:
    """The kernels of this package bin and composite on 16 x 16 tiles.  `block_size` is a configuration field of the reference's
    gsplat renderers (gsplat_renderer.py:6,34-43, gsplat_v1_renderer.py:23-41) that selects the tile side of ITS rasterizer: the
    rendered image and every gradient are independent of it (it decides how the per-tile lists are cut, nothing else), so a renderer
    configured with another value renders the same result on 16 x 16 tiles; said once per value.  (The op-level entry points —
    `ops.isect_tiles`, `ops.rasterize_to_pixels`, `ops.bin_gaussians`, `ops.rasterize_gaussians` — take tile sizes 8, 16 and 32: there
    the per-tile lists ARE the interface; 16 is the fast path, which is why the renderers stay on it.)"""
    if int(block_size) != 16 and block_size not in _TILE_NOTE:
        _TILE_NOTE.add(block_size)
        import warnings
        warnings.warn(f"block_size={block_size}: this rasterizer tiles the image 16 x 16; the image and the gradients do not depend on "
                      f"the tile size, the configured value is kept for the checkpoint only", stacklevel=3)
    return 16
