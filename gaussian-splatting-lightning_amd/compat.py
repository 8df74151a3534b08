"""Import shims for the reference's un-vendored native packages that this package replaces.

Besides the two small helpers below (`simple_knn`, `fused_ssim`), `install()` registers stand-ins for the two rasterizer
packages themselves — `diff_gaussian_rasterization` and the yzslab `gsplat` fork — under the module paths and function names the
reference imports (internal/renderers/vanilla_renderer.py:4, gsplat_renderer.py:2-4, gsplat_v1_renderer.py:8-20,
pypreprocess_gsplat_renderer.py:1-2, gsplat_hit_pixel_count_renderer.py:5, internal/optimizers.py:34 ...), each bound to the HIP
op of `gspl_amd.ops` with the same signature.  With them the reference's OWN renderer classes (`VanillaRenderer`,
`GSPlatRenderer`, `GSplatV1Renderer`, the research renderers built on their static helpers) run unedited on the HIP kernels; the
`Hip*` plugins of `gspl_amd.renderers` remain the faster route (fused calls, list-only binning, channels-first images).
Functions of the fork that are not built (`compute_relocation`, `depth_to_normal`, `rasterize_to_vis_aware_weights`) are left
out: importing them raises ImportError as it would without the package.

The reference imports them by their own module names inside functions, e.g.
`from simple_knn._C import distCUDA2` (internal/models/vanilla_gaussian.py:122) or `from fused_ssim import fused_ssim`
(internal/metrics/vanilla_metrics.py:36).  `install()` registers stand-in
modules under those names — only for packages that are NOT importable — so that the reference runs unedited once
`gspl_amd.renderers` has been imported (which the `--model.renderer gspl_amd.renderers.<Name>` option does while the
configuration is parsed, long before the model is initialised from a point cloud).
"""
from __future__ import annotations

import importlib.util
import sys
import types


def _missing(name: str) -> bool:
    if name in sys.modules:
        return False
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError):
        return True


def _late(name: str):
    """Call-time binding to `gspl_amd.ops.<name>` (so that the op can be looked up — or replaced in a test — after install)."""
    def fn(*args, **kwargs):
        from . import ops
        return getattr(ops, name)(*args, **kwargs)
    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = f"gspl_amd.ops.{name} (HIP)"
    return fn


def _isect_tiles_tile_based_culling(means2d, radii, depths, conics, opacities, tile_size, tile_width, tile_height, packed=False,
                                    n_cameras=1, camera_ids=None, gaussian_ids=None):
    """The fork's `isect_tiles_tile_based_culling` as the reference calls it (gsplat_v1_renderer.py:497-510): a (tile, Gaussian)
    pair is listed only if the Gaussian can reach alpha >= 1/255 in the tile.  Served by the list-only two-level binning, which
    produces the sorted lists directly: the `isect_ids` it returns is an empty tensor that carries the tile offsets to
    `isect_offset_encode_tile_based_culling` (the reference hands it straight over, :511-517)."""
    import torch
    from . import ops
    if packed or camera_ids is not None or gaussian_ids is not None or n_cameras not in (None, 1):
        raise NotImplementedError("one camera per call, unpacked (what the reference uses)")
    flat, offsets = ops.bin_gaussians(means2d.reshape(-1, 2), depths.reshape(-1), radii.reshape(-1), int(tile_height) * int(tile_size),
                                      int(tile_width) * int(tile_size), int(tile_size), conics=conics.reshape(-1, 3),
                                      opacities=opacities.reshape(-1))
    carrier = torch.empty((0,), dtype=torch.int64, device=flat.device)
    carrier._gspl_offsets = offsets.reshape(1, int(tile_height), int(tile_width))
    return None, carrier, flat


def _isect_offset_encode_tile_based_culling(isect_ids, flatten_ids, n_cameras, tile_width, tile_height):
    offsets = getattr(isect_ids, "_gspl_offsets", None)
    if offsets is None:
        raise ValueError("isect_ids must come from gspl_amd's isect_tiles_tile_based_culling")
    return offsets, flatten_ids


def _rasterize_to_pixels_fork(*args, **kwargs):
    """The fork's rasterize_to_pixels ALWAYS leaves `means2d.has_hit_any_pixels` behind (gsplat_v1_renderer.py:287 reads it
    unconditionally); ops.rasterize_to_pixels does so on request."""
    from . import ops
    kwargs.setdefault("track_hits", True)
    return ops.rasterize_to_pixels(*args, **kwargs)


def _module(name: str, doc: str, **attrs):
    mod = types.ModuleType(name)
    mod.__doc__ = doc
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def _install_rasterizer_packages(installed: list):
    if _missing("diff_gaussian_rasterization"):
        from . import ops
        _module("diff_gaussian_rasterization", "gspl_amd stand-in for diff_gaussian_rasterization (HIP; gspl_amd.ops.GaussianRasterizer)",
                GaussianRasterizationSettings=ops.GaussianRasterizationSettings, GaussianRasterizer=ops.GaussianRasterizer)
        installed.append("diff_gaussian_rasterization")
    if _missing("gsplat"):
        doc = "gspl_amd stand-in for the gsplat fork (HIP ops of gspl_amd.ops under the fork's module paths)"
        pkg = _module("gsplat", doc, spherical_harmonics=_late("spherical_harmonics"))
        pkg.__path__ = []
        pkg.sh = _module("gsplat.sh", doc, spherical_harmonics=_late("spherical_harmonics"))
        pkg.sh_decomposed = _module("gsplat.sh_decomposed", doc, spherical_harmonics_decomposed=_late("spherical_harmonics_decomposed"))
        pkg.rasterize = _module("gsplat.rasterize", doc, rasterize_gaussians=_late("rasterize_gaussians"))
        pkg.project_gaussians = _module("gsplat.project_gaussians", doc, project_gaussians=_late("project_gaussians"))
        pkg.v0_interfaces = _module("gsplat.v0_interfaces", doc, project_gaussians=_late("project_gaussians"),
                                    rasterize_gaussians=_late("rasterize_gaussians"), rasterize_to_pixels=_rasterize_to_pixels_fork)
        cuda = _module("gsplat.cuda", doc)
        cuda.__path__ = []
        pkg.cuda = cuda
        cuda._wrapper = _module("gsplat.cuda._wrapper", doc, fully_fused_projection=_late("fully_fused_projection"),
                                isect_tiles=_late("isect_tiles"), isect_offset_encode=_late("isect_offset_encode"),
                                spherical_harmonics=_late("spherical_harmonics"), rasterize_to_pixels=_rasterize_to_pixels_fork)
        cuda.isect_tiles_tile_based_culling = _module(
            "gsplat.cuda.isect_tiles_tile_based_culling", doc, isect_tiles_tile_based_culling=_isect_tiles_tile_based_culling,
            isect_offset_encode_tile_based_culling=_isect_offset_encode_tile_based_culling)
        pkg.hit_pixel_count = _module("gsplat.hit_pixel_count", doc, hit_pixel_count=_late("hit_pixel_count"))
        pkg.rasterize_to_weights = _module("gsplat.rasterize_to_weights", doc, rasterize_to_weights=_late("rasterize_to_weights"))
        from . import optimizers
        pkg.optimizers = _module("gsplat.optimizers", doc, SelectiveAdam=optimizers.SelectiveAdam)
        installed.append("gsplat")


def install() -> list:
    """Returns the names of the shim modules that were installed."""
    installed = []
    _install_rasterizer_packages(installed)
    if _missing("simple_knn"):
        from . import ops
        pkg = types.ModuleType("simple_knn")
        pkg.__doc__ = "gspl_amd stand-in for simple_knn (HIP; see gspl_amd.ops.distCUDA2)"
        sub = types.ModuleType("simple_knn._C")
        sub.distCUDA2 = ops.distCUDA2
        pkg._C = sub
        pkg.__path__ = []          # a package, so that `from simple_knn._C import ...` resolves through sys.modules
        sys.modules["simple_knn"] = pkg
        sys.modules["simple_knn._C"] = sub
        installed.append("simple_knn._C")
    if _missing("fused_ssim"):
        from . import ops
        mod = types.ModuleType("fused_ssim")
        mod.__doc__ = "gspl_amd stand-in for fused_ssim (HIP; see gspl_amd.ops.fused_ssim)"
        mod.fused_ssim = ops.fused_ssim
        sys.modules["fused_ssim"] = mod
        installed.append("fused_ssim")
    return installed
