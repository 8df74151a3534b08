/*
 * composite_oracle.c — CPU restatement (fp64 arithmetic, fp64 inputs: fp32 data widens exactly) of the tile compositing
 * stage and its analytic backward.  TEST INFRASTRUCTURE ONLY: imported by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg; the product path never links it.
 *
 * What it restates.  The reference (yzslab/gaussian-splatting-lightning) holds no source for this
 * stage: it calls third-party CUDA ops
 *     gsplat  (yzslab/gsplat @ c27a44d4)                      rasterize_to_pixels / rasterize_gaussians
 *         call sites internal/renderers/gsplat_v1_renderer.py:588-601, gsplat_renderer.py:86-99,
 *                    pypreprocess_gsplat_renderer.py:45-58
 *     diff_gaussian_rasterization (graphdeco-inria @ 59f5f77e) GaussianRasterizer
 *         call site  internal/renderers/vanilla_renderer.py:111-120
 * so this file restates the published 3DGS compositing rule with the per-API constants listed in
 * SURVEY.md Appendix B.  PARITY UNPINNED against those CUDA packages (they are not installable
 * here); the backward is pinned instead against fp64 torch.autograd of an independent
 * differentiable restatement (oracle/gsplat_oracle.py: composite_autograd) in tests/.
 *
 *   per pixel p (centre at +0.5 in gsplat mode, at the integer in Inria mode), front-to-back over
 *   the tile's depth-sorted list:
 *       sigma = 0.5 (a dx^2 + c dy^2) + b dx dy,  d = mean2d - p
 *       alpha = min(alpha_max, opacity * exp(-sigma));  skip if sigma < 0 or alpha < 1/255
 *       next_T = T (1 - alpha);  stop if next_T <= 1e-4 (gsplat)  /  < 1e-4 (Inria)
 *       C += colour * alpha * T;  T = next_T
 *   out = C + T * background;  out_alpha = 1 - T;  last = one past the last contributing index.
 *
 * Discrete decisions (skip / stop / clamp) made within a relative margin of their threshold are
 * reported per pixel ("fragile"), because an fp32 implementation may legitimately decide them the
 * other way; tests exclude those pixels from the strict tolerance and bound how many there are.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC composite_oracle.c -o _build/libgspl_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MODE_GSPLAT 0
#define MODE_INRIA 1
/* side of the tiles the lists are cut on: 16 unless the caller says otherwise (8 / 32: `block_size` of the reference's renderers) */
static int TILE = 16;
void oracle_set_tile(int tile) { TILE = tile; }

typedef struct {
    double alpha_max;
    double centre;
    int stop_inclusive;
    int clamp_kills_grad;
} mode_t_;

static mode_t_ mode_of(int mode) {
    mode_t_ m;
    if (mode == MODE_INRIA) { m.alpha_max = (double)0.99f; m.centre = 0.0; m.stop_inclusive = 0; m.clamp_kills_grad = 0; }
    else { m.alpha_max = (double)0.999f; m.centre = 0.5; m.stop_inclusive = 1; m.clamp_kills_grad = 1; }
    return m;
}

static const double ALPHA_MIN = (double)(1.0f / 255.0f);
static const double T_STOP = (double)1e-4f;
static const double MARGIN = 2e-5;   /* relative margin that flags a decision as fragile */

static int near_rel(double v, double thr) { return fabs(v - thr) <= MARGIN * fabs(thr); }
/* A FREE-RUNNING comparison (an fp32 pipeline against this oracle on its own fp64 per-splat values, not on the pipeline's) also has to
 * allow for the rounding of the compositing's INPUTS: the projected mean carries a few ulps of its own magnitude (up to the image
 * width: 1e-4 pixels at x ~ 1900), the conic and the opacity a few ulps each.  With input_eps = k 2^-24 set, a decision is fragile
 * within MARGIN + the first-order effect of such an error on the quantity decided:
 *     d(alpha)/alpha = d(sigma) + d(o)/o,   d(sigma) <= eps (|a dx + b dy| |x| + |b dx + c dy| |y| + |a dx^2|/2 + |c dy^2|/2 + |b dx dy|),
 *     d(T)/T = sum over the blended splats of d(alpha) / (1 - alpha).
 * 0 (default): the locked comparisons, where the oracle composites AT the pipeline's values. */
static double INPUT_EPS = 0.0;
void oracle_set_input_eps(double e) { INPUT_EPS = e > 0.0 ? e : 0.0; }
static int near_rel_m(double v, double thr, double extra) { return fabs(v - thr) <= (MARGIN + extra) * fabs(thr); }
static double alpha_rel_err(double a, double b, double cc, double dx, double dy, double mx, double my) {
    if (INPUT_EPS == 0.0) return 0.0;
    return INPUT_EPS * (fabs(a * dx + b * dy) * fabs(mx) + fabs(b * dx + cc * dy) * fabs(my)
                        + 0.5 * fabs(a * dx * dx) + 0.5 * fabs(cc * dy * dy) + fabs(b * dx * dy) + 1.0);
}

/* out_colors is HWC [H,W,D]; fragile [H,W] u8 (nullable). */
void oracle_composite_fwd(int mode, int64_t n_isects, int D,
                          const double* means2d, const double* conics, const double* colors,
                          const double* opacities, const double* backgrounds,
                          int width, int height, int tile_w, int tile_h,
                          const int32_t* offsets, const int32_t* flatten_ids,
                          double* out_colors, double* out_alphas, int32_t* last_ids, uint8_t* fragile) {
    const mode_t_ M = mode_of(mode);
    const int n_tiles = tile_w * tile_h;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int start = offsets[tile];
        const int end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
        const int tx = tile % tile_w, ty = tile / tile_w;
        for (int ly = 0; ly < TILE; ++ly) {
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= width || py >= height) continue;
                const double pxf = px + M.centre, pyf = py + M.centre;
                double T = 1.0, t_err = 0.0;
                double acc[16];
                for (int c = 0; c < D; ++c) acc[c] = 0.0;
                int last = start;
                int frag = 0;
                for (int i = start; i < end; ++i) {
                    const int g = flatten_ids[i];
                    const double dx = (double)means2d[g * 2 + 0] - pxf, dy = (double)means2d[g * 2 + 1] - pyf;
                    const double a = conics[g * 3 + 0], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
                    const double sigma = 0.5 * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    const double raw = (double)opacities[g] * exp(-sigma);
                    const double alpha = raw < M.alpha_max ? raw : M.alpha_max;
                    const double a_err = alpha_rel_err(a, b, cc, dx, dy, means2d[g * 2 + 0], means2d[g * 2 + 1]);
                    if (near_rel_m(alpha, ALPHA_MIN, a_err) || fabs(sigma) < 1e-7) frag = 1;
                    if (sigma < 0.0 || alpha < ALPHA_MIN) continue;
                    if (M.clamp_kills_grad && near_rel_m(raw, M.alpha_max, a_err)) frag = 1;
                    const double next_T = T * (1.0 - alpha);
                    t_err += a_err * alpha / (1.0 - alpha);
                    if (near_rel_m(next_T, T_STOP, t_err)) frag = 1;
                    if (M.stop_inclusive ? (next_T <= T_STOP) : (next_T < T_STOP)) break;
                    const double w = alpha * T;
                    for (int c = 0; c < D; ++c) acc[c] += (double)colors[(int64_t)g * D + c] * w;
                    T = next_T;
                    last = i + 1;
                }
                const int64_t pix = (int64_t)py * width + px;
                for (int c = 0; c < D; ++c) out_colors[pix * D + c] = acc[c] + T * (backgrounds ? (double)backgrounds[c] : 0.0);
                out_alphas[pix] = 1.0 - T;
                last_ids[pix] = last;
                if (fragile) fragile[pix] = (uint8_t)frag;
            }
        }
    }
}

/*
 * Backward.  v_out_colors HWC [H,W,D], v_out_alphas [H,W] nullable.  out_alphas / last_ids are the
 * forward's (from oracle_composite_fwd, or from the implementation under test so that both sides
 * differentiate the same discrete path).  All v_* outputs are fp64 and must be zeroed by the caller.
 * fragile_g [N] u8 (nullable): splats that own a pair inside a pixel flagged in fragile_px.
 */
void oracle_composite_bwd(int mode, int N, int64_t n_isects, int D,
                          const double* means2d, const double* conics, const double* colors,
                          const double* opacities, const double* backgrounds,
                          int width, int height, int tile_w, int tile_h,
                          const int32_t* offsets, const int32_t* flatten_ids,
                          const double* out_alphas, const int32_t* last_ids,
                          const double* v_out_colors, const double* v_out_alphas,
                          const uint8_t* fragile_px,
                          double* v_means2d, double* v_means2d_abs, double* v_conics,
                          double* v_colors, double* v_opacities, uint8_t* fragile_g) {
    const mode_t_ M = mode_of(mode);
    const int n_tiles = tile_w * tile_h;
    (void)N;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int start = offsets[tile];
        const int tx = tile % tile_w, ty = tile / tile_w;
        for (int ly = 0; ly < TILE; ++ly) {
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= width || py >= height) continue;
                const int64_t pix = (int64_t)py * width + px;
                const double pxf = px + M.centre, pyf = py + M.centre;
                const double T_final = 1.0 - out_alphas[pix];
                double T = T_final;
                double buffer[16], vo[16];
                double bgdot = 0.0;
                for (int c = 0; c < D; ++c) {
                    buffer[c] = 0.0;
                    vo[c] = v_out_colors[pix * D + c];
                    if (backgrounds) bgdot += (double)backgrounds[c] * vo[c];
                }
                const double voa = v_out_alphas ? v_out_alphas[pix] : 0.0;
                const int frag_px = fragile_px ? fragile_px[pix] : 0;
                for (int i = last_ids[pix] - 1; i >= start; --i) {
                    const int g = flatten_ids[i];
                    const double dx = (double)means2d[g * 2 + 0] - pxf, dy = (double)means2d[g * 2 + 1] - pyf;
                    const double a = conics[g * 3 + 0], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
                    const double sigma = 0.5 * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    const double vis = exp(-sigma);
                    const double o = opacities[g];
                    const double raw = o * vis;
                    const double alpha = raw < M.alpha_max ? raw : M.alpha_max;
                    if (sigma < 0.0 || alpha < ALPHA_MIN) continue;
                    if (frag_px && fragile_g) {
#pragma omp atomic write
                        fragile_g[g] = 1;
                    }
                    const double ra = 1.0 / (1.0 - alpha);
                    T *= ra;
                    const double fac = alpha * T;
                    double v_alpha = T_final * ra * (voa - bgdot);
                    for (int c = 0; c < D; ++c) {
                        const double col = colors[(int64_t)g * D + c];
                        const double vr = fac * vo[c];
#pragma omp atomic
                        v_colors[(int64_t)g * D + c] += vr;
                        v_alpha += (col * T - buffer[c] * ra) * vo[c];
                        buffer[c] += col * fac;
                    }
                    if (M.clamp_kills_grad && !(raw <= M.alpha_max)) continue;
                    const double v_sigma = -raw * v_alpha;
                    const double gx = v_sigma * (a * dx + b * dy), gy = v_sigma * (b * dx + cc * dy);
#pragma omp atomic
                    v_means2d[g * 2 + 0] += gx;
#pragma omp atomic
                    v_means2d[g * 2 + 1] += gy;
                    if (v_means2d_abs) {
#pragma omp atomic
                        v_means2d_abs[g * 2 + 0] += fabs(gx);
#pragma omp atomic
                        v_means2d_abs[g * 2 + 1] += fabs(gy);
                    }
#pragma omp atomic
                    v_conics[g * 3 + 0] += 0.5 * v_sigma * dx * dx;
#pragma omp atomic
                    v_conics[g * 3 + 1] += v_sigma * dx * dy;
#pragma omp atomic
                    v_conics[g * 3 + 2] += 0.5 * v_sigma * dy * dy;
#pragma omp atomic
                    v_opacities[g] += vis * v_alpha;
                }
            }
        }
    }
}

/*
 * Attribution of the free-running tests' gradient tail: the splats whose gradient a flipped decision of a FRAGILE
 * pixel can reach.  For every pixel flagged in fragile_px the whole list is walked front to back with the oracle's own (fp64)
 * decisions, and every splat that is — or within MARGIN could be — blended there is marked: alpha >= ALPHA_MIN (1 - MARGIN) up to
 * the transmittance stop; when the stop itself is the near-threshold decision (next_T within 1 % of T_STOP: one 1/255 flip in
 * front of it moves T by at most 0.4 %) the walk continues on the path that does not stop there.  A splat's gradient is a sum
 * over the pixels that blend it, and a flipped decision in a pixel moves the terms of every splat blended in that pixel (through
 * T in front of it and the colour behind it): these are exactly the rows an fp32 implementation may legitimately miss by more
 * than rounding.  fragile_g [N] u8, zeroed by the caller.
 */
void oracle_fragile_splats(int mode, int64_t n_isects,
                           const double* means2d, const double* conics, const double* opacities,
                           int width, int height, int tile_w, int tile_h,
                           const int32_t* offsets, const int32_t* flatten_ids,
                           const uint8_t* fragile_px, uint8_t* fragile_g) {
    const mode_t_ M = mode_of(mode);
    const int n_tiles = tile_w * tile_h;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int start = offsets[tile];
        const int end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
        const int tx = tile % tile_w, ty = tile / tile_w;
        for (int ly = 0; ly < TILE; ++ly) {
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= width || py >= height) continue;
                if (!fragile_px[(int64_t)py * width + px]) continue;
                const double pxf = px + M.centre, pyf = py + M.centre;
                double T = 1.0;
                for (int i = start; i < end; ++i) {
                    const int g = flatten_ids[i];
                    const double dx = means2d[g * 2 + 0] - pxf, dy = means2d[g * 2 + 1] - pyf;
                    const double a = conics[g * 3 + 0], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
                    const double sigma = 0.5 * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    const double raw = opacities[g] * exp(-sigma);
                    const double alpha = raw < M.alpha_max ? raw : M.alpha_max;
                    const double a_err = alpha_rel_err(a, b, cc, dx, dy, means2d[g * 2 + 0], means2d[g * 2 + 1]);
                    if (sigma < -1e-7 || alpha < ALPHA_MIN * (1.0 - MARGIN - a_err)) continue;
#pragma omp atomic write
                    fragile_g[g] = 1;
                    if (alpha < ALPHA_MIN) continue;               /* the oracle skips it; an fp32 evaluation may blend it */
                    const double next_T = T * (1.0 - alpha);
                    const int stops = M.stop_inclusive ? (next_T <= T_STOP) : (next_T < T_STOP);
                    if (stops && fabs(next_T - T_STOP) > 0.01 * T_STOP) break;
                    T = next_T;
                }
            }
        }
    }
}

/*
 * The one list-level decision a free-running comparison re-takes per PIXEL: the depth order of two splats that both blend there.
 * The lists are sorted on the fp32 bits of the view-space depth (gaussian_projection.py:190-204); an fp32 pipeline and the fp64
 * oracle round that depth differently in the last bits, so two splats whose depths agree to within `tol_rel` (a few fp32 ulps) may
 * be composited in either order — which moves the pixel by up to alpha_i alpha_j |c_i - c_j| and the gradient terms of both.
 * Flags (ORs into fragile_px) every pixel in which two blended splats are that close in depth.  depths [N] as the oracle's own.
 */
void oracle_fragile_order(int mode, int64_t n_isects,
                          const double* means2d, const double* conics, const double* opacities, const double* depths,
                          int width, int height, int tile_w, int tile_h,
                          const int32_t* offsets, const int32_t* flatten_ids, double tol_rel, uint8_t* fragile_px) {
    const mode_t_ M = mode_of(mode);
    const int n_tiles = tile_w * tile_h;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int start = offsets[tile];
        const int end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
        const int tx = tile % tile_w, ty = tile / tile_w;
        for (int ly = 0; ly < TILE; ++ly) {
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= width || py >= height) continue;
                const double pxf = px + M.centre, pyf = py + M.centre;
                double T = 1.0, prev_depth = -1e300;
                for (int i = start; i < end; ++i) {
                    const int g = flatten_ids[i];
                    const double dx = means2d[g * 2 + 0] - pxf, dy = means2d[g * 2 + 1] - pyf;
                    const double a = conics[g * 3 + 0], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
                    const double sigma = 0.5 * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    const double raw = opacities[g] * exp(-sigma);
                    const double alpha = raw < M.alpha_max ? raw : M.alpha_max;
                    if (sigma < 0.0 || alpha < ALPHA_MIN) continue;
                    if (fabs(depths[g] - prev_depth) <= tol_rel * fabs(depths[g])) { fragile_px[(int64_t)py * width + px] = 1; break; }
                    prev_depth = depths[g];
                    const double next_T = T * (1.0 - alpha);
                    if (M.stop_inclusive ? (next_T <= T_STOP) : (next_T < T_STOP)) break;
                    T = next_T;
                }
            }
        }
    }
}

/* fp32 variants of the same loops, used ONLY as the timed CPU baseline (bench.py cpu_baseline,
 * kind "port"): same algorithm, single precision, OpenMP over tiles. */
void oracle_composite_fwd_f32(int mode, int64_t n_isects, int D,
                              const float* means2d, const float* conics, const float* colors,
                              const float* opacities, const float* backgrounds,
                              int width, int height, int tile_w, int tile_h,
                              const int32_t* offsets, const int32_t* flatten_ids,
                              float* out_colors, float* out_alphas, int32_t* last_ids) {
    const mode_t_ M = mode_of(mode);
    const int n_tiles = tile_w * tile_h;
    const float amax = (float)M.alpha_max, centre = (float)M.centre;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int start = offsets[tile];
        const int end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
        const int tx = tile % tile_w, ty = tile / tile_w;
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= width || py >= height) continue;
                const float pxf = px + centre, pyf = py + centre;
                float T = 1.f, acc[16];
                for (int c = 0; c < D; ++c) acc[c] = 0.f;
                int last = start;
                for (int i = start; i < end; ++i) {
                    const int g = flatten_ids[i];
                    const float dx = means2d[g * 2 + 0] - pxf, dy = means2d[g * 2 + 1] - pyf;
                    const float sigma = 0.5f * (conics[g * 3 + 0] * dx * dx + conics[g * 3 + 2] * dy * dy) + conics[g * 3 + 1] * dx * dy;
                    if (sigma < 0.f) continue;
                    float alpha = opacities[g] * expf(-sigma);
                    if (alpha > amax) alpha = amax;
                    if (alpha < 1.f / 255.f) continue;
                    const float next_T = T * (1.f - alpha);
                    if (M.stop_inclusive ? (next_T <= 1e-4f) : (next_T < 1e-4f)) break;
                    const float w = alpha * T;
                    for (int c = 0; c < D; ++c) acc[c] += colors[(int64_t)g * D + c] * w;
                    T = next_T;
                    last = i + 1;
                }
                const int64_t pix = (int64_t)py * width + px;
                for (int c = 0; c < D; ++c) out_colors[pix * D + c] = acc[c] + T * (backgrounds ? backgrounds[c] : 0.f);
                out_alphas[pix] = 1.f - T;
                last_ids[pix] = last;
            }
    }
}

void oracle_composite_bwd_f32(int mode, int N, int64_t n_isects, int D,
                              const float* means2d, const float* conics, const float* colors,
                              const float* opacities, const float* backgrounds,
                              int width, int height, int tile_w, int tile_h,
                              const int32_t* offsets, const int32_t* flatten_ids,
                              const float* out_alphas, const int32_t* last_ids,
                              const float* v_out_colors,
                              float* v_means2d, float* v_conics, float* v_colors, float* v_opacities) {
    const mode_t_ M = mode_of(mode);
    const int n_tiles = tile_w * tile_h;
    const float amax = (float)M.alpha_max, centre = (float)M.centre;
    (void)N; (void)n_isects;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int start = offsets[tile];
        const int tx = tile % tile_w, ty = tile / tile_w;
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= width || py >= height) continue;
                const int64_t pix = (int64_t)py * width + px;
                const float pxf = px + centre, pyf = py + centre;
                const float T_final = 1.f - out_alphas[pix];
                float T = T_final, buffer[16], vo[16], bgdot = 0.f;
                for (int c = 0; c < D; ++c) { buffer[c] = 0.f; vo[c] = v_out_colors[pix * D + c]; if (backgrounds) bgdot += backgrounds[c] * vo[c]; }
                for (int i = last_ids[pix] - 1; i >= start; --i) {
                    const int g = flatten_ids[i];
                    const float dx = means2d[g * 2 + 0] - pxf, dy = means2d[g * 2 + 1] - pyf;
                    const float a = conics[g * 3 + 0], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
                    const float sigma = 0.5f * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    if (sigma < 0.f) continue;
                    const float vis = expf(-sigma);
                    const float raw = opacities[g] * vis;
                    const float alpha = raw < amax ? raw : amax;
                    if (alpha < 1.f / 255.f) continue;
                    const float ra = 1.f / (1.f - alpha);
                    T *= ra;
                    const float fac = alpha * T;
                    float v_alpha = -T_final * ra * bgdot;
                    for (int c = 0; c < D; ++c) {
                        const float col = colors[(int64_t)g * D + c];
#pragma omp atomic
                        v_colors[(int64_t)g * D + c] += fac * vo[c];
                        v_alpha += (col * T - buffer[c] * ra) * vo[c];
                        buffer[c] += col * fac;
                    }
                    if (M.clamp_kills_grad && !(raw <= amax)) continue;
                    const float v_sigma = -raw * v_alpha;
#pragma omp atomic
                    v_means2d[g * 2 + 0] += v_sigma * (a * dx + b * dy);
#pragma omp atomic
                    v_means2d[g * 2 + 1] += v_sigma * (b * dx + cc * dy);
#pragma omp atomic
                    v_conics[g * 3 + 0] += 0.5f * v_sigma * dx * dx;
#pragma omp atomic
                    v_conics[g * 3 + 1] += v_sigma * dx * dy;
#pragma omp atomic
                    v_conics[g * 3 + 2] += 0.5f * v_sigma * dy * dy;
#pragma omp atomic
                    v_opacities[g] += vis * v_alpha;
                }
            }
    }
}

#ifdef _OPENMP
#include <omp.h>
void oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
void oracle_set_threads(int n) { (void)n; }
#endif

int oracle_version(void) { return 1; }
