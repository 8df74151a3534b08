"""TEST INFRASTRUCTURE ONLY (never imported by the product).  Per-splat statistics of a compositing pass, fp64, a python
loop over tiles vectorised over the pixels of a tile — the same discrete rules as `gsplat_oracle.composite_autograd`
(alpha >= 1/255, a pixel stops before the splat that would push its transmittance below 1e-4).

PARITY UNPINNED: the kernels these sums stand in for (`hit_pixel_count`, `rasterize_to_weights` of the reference's gsplat
fork, call sites internal/renderers/gsplat_hit_pixel_count_renderer.py:34-44 and
internal/density_controllers/taming_3dgs_density_controller.py:429-439) are not in /root/reference; the definitions are
restated from the published methods (LightGaussian: hit count, opacity, alpha, alpha*T; Taming-3DGS: pixel-weighted
blending weight, pixel count, blending weight, pixel distance)."""
import numpy as np

from . import gsplat_oracle as O

TILE = 16


def scores(mode, means2d, conics, opacities, width, height, offsets, flatten_ids, pixel_weights=None):
    """-> dict of fp64 arrays [N]: count, opacity, alpha, visibility, weighted (zeros without pixel_weights), dist."""
    m = np.asarray(means2d, np.float64)
    c = np.asarray(conics, np.float64)
    o = np.asarray(opacities, np.float64).reshape(-1)
    offsets = np.asarray(offsets).reshape(-1)
    flat = np.asarray(flatten_ids)
    N = m.shape[0]
    alpha_max, centre = (float(np.float32(0.99)), 0.0) if mode == O.MODE_INRIA else (float(np.float32(0.999)), 0.5)
    alpha_min, t_stop = float(np.float32(1.0) / np.float32(255.0)), float(np.float32(1e-4))
    tile_w, tile_h = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    out = {k: np.zeros(N) for k in ("count", "opacity", "alpha", "visibility", "weighted", "dist")}
    for tile in range(tile_w * tile_h):
        start = int(offsets[tile])
        end = int(offsets[tile + 1]) if tile + 1 < tile_w * tile_h else flat.shape[0]
        ty, tx = divmod(tile, tile_w)
        ys = np.arange(ty * TILE, min((ty + 1) * TILE, height))
        xs = np.arange(tx * TILE, min((tx + 1) * TILE, width))
        py, px = np.meshgrid(ys, xs, indexing="ij")
        pxf, pyf = px + centre, py + centre
        wpx = None if pixel_weights is None else np.asarray(pixel_weights, np.float64)[py, px]
        T = np.ones_like(pxf, dtype=np.float64)
        done = np.zeros_like(T, dtype=bool)
        for i in range(start, end):
            g = int(flat[i])
            dx, dy = m[g, 0] - pxf, m[g, 1] - pyf
            sigma = 0.5 * (c[g, 0] * dx * dx + c[g, 2] * dy * dy) + c[g, 1] * dx * dy
            alpha = np.minimum(alpha_max, o[g] * np.exp(-sigma))
            valid = (~done) & (sigma >= 0) & (alpha >= alpha_min)
            next_T = T * (1 - alpha)
            stop = valid & ((next_T < t_stop) if mode == O.MODE_INRIA else (next_T <= t_stop))
            done |= stop
            contrib = valid & ~stop
            w = np.where(contrib, alpha * T, 0.0)
            out["count"][g] += contrib.sum()
            out["opacity"][g] += contrib.sum() * o[g]
            out["alpha"][g] += np.where(contrib, alpha, 0.0).sum()
            out["visibility"][g] += w.sum()
            if wpx is not None:
                out["weighted"][g] += (w * wpx).sum()
            out["dist"][g] += np.where(contrib, np.sqrt(dx * dx + dy * dy), 0.0).sum()
            T = np.where(contrib, next_T, T)
    return out
