/*
 * gspl_hip.h — C ABI of the MI355X (gfx950) Gaussian-splatting rasterizer hot path.
 *
 * This is the drop-in boundary: every entry point takes raw device
 * pointers + sizes + a hipStream_t passed as void*, returns an int status
 * (GSPL_OK == 0) and never throws.  No torch types cross this line; the Python
 * host side (gaussian-splatting-lightning_amd/ops/) binds it with ctypes and
 * allocates every buffer through torch's caching allocator.
 *
 * Each function names the reference interface it replaces.  Paths are relative to
 * the reference tree (yzslab/gaussian-splatting-lightning); the native ops the
 * reference calls live in un-vendored third-party CUDA packages
 * (gsplat @ yzslab/gsplat c27a44d4, diff_gaussian_rasterization @ 59f5f77e), so the
 * citations are the reference's *call sites* of those ops.
 *
 * Conventions
 *   - all arrays are dense, row-major, fp32 unless stated; "i32"/"i64"/"u8" stated.
 *   - quaternions are (w,x,y,z) and are used as given (the reference hands over
 *     normalised rotations: internal/models/vanilla_gaussian.py:357-358,
 *     internal/utils/gaussian_projection.py:211-232).
 *   - `mode` selects the per-API constants:
 *       GSPL_MODE_GSPLAT : pixel centre at +0.5, alpha <= 0.999, stop when T(1-a) <= 1e-4,
 *                          clamped alpha has zero gradient, tile rect [floor, floor+1)
 *       GSPL_MODE_INRIA  : pixel centre at integers, alpha <= 0.99, stop when T(1-a) < 1e-4,
 *                          clamp ignored in backward, tile rect [(p-r)/T, (p+r+T-1)/T)
 *   - nullable pointers are marked; a NULL optional output is simply not written.
 */
#ifndef GSPL_HIP_H
#define GSPL_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
enum {
    GSPL_OK = 0,
    GSPL_ERR_INVALID_ARG = 1,   /* bad size / NULL required pointer / unsupported channel count */
    GSPL_ERR_WORKSPACE = 2,     /* caller-provided workspace too small */
    GSPL_ERR_LAUNCH = 3,        /* hipGetLastError() != hipSuccess after a launch */
    GSPL_ERR_UNSUPPORTED = 4
};

enum { GSPL_MODE_GSPLAT = 0, GSPL_MODE_INRIA = 1 };

/* camera model of gspl_project_fwd/bwd (gsplat v1 `camera_model`, reference option internal/renderers/gsplat_v1_renderer.py:50) */
enum { GSPL_CAMERA_PINHOLE = 0, GSPL_CAMERA_ORTHO = 1, GSPL_CAMERA_FISHEYE = 2 };

/* image memory layout of composite outputs / incoming image gradients */
enum { GSPL_LAYOUT_HWC = 0, GSPL_LAYOUT_CHW = 1 };

/* ABI version, bumped on any signature change; checked by the ctypes loader. */
int gspl_abi_version(void);
/* Human-readable description of the last launch error on this thread (never NULL). */
const char* gspl_last_error(void);

/* ------------------------------------------------------------------------------------------
 * 1. Projection (EWA splatting): 3D mean/scale/rotation -> 2D mean, depth, conic, radius.
 *    Replaces gsplat `fully_fused_projection` (internal/renderers/gsplat_v1_renderer.py:408-421,
 *    gsplat_distributed_renderer.py:271-283) and gsplat-v0 `project_gaussians`
 *    (gsplat_renderer.py:64-79); math pinned to internal/utils/gaussian_projection.py:6-138.
 *    One thread per (camera, Gaussian); C cameras batched.
 *      viewmats [C,4,4]  world->camera, standard (non-transposed) row-major, device memory
 *      Ks       [C,3,3]  intrinsics, device memory
 *    Outputs (all [C,N,...]): radii i32 (0 = culled), means2d [.,2], depths, conics [.,3],
 *    compensations (nullable), tiles_hit i32 (nullable; v0 `num_tiles_hit`), cov3d [.,6] (nullable): the upper triangle
 *    (xx, xy, xz, yy, yz, zz) of Sigma = (R S)(R S)^T that `project_gaussians` returns as `cov3d` (gaussian_projection.py:47,137).
 *    Culled Gaussians get zeros in every output (gaussian_projection.py:127-136).
 *    camera_model: GSPL_CAMERA_PINHOLE is the in-tree Python above (1.3 tan(fov) clamp of the Jacobian's x/z, y/z).
 *    GSPL_CAMERA_ORTHO / GSPL_CAMERA_FISHEYE (the `camera_model` option the reference passes through,
 *    gsplat_v1_renderer.py:50,154) restate the published gsplat models — ortho: mean2d = (fx x + cx, fy y + cy),
 *    J = diag(fx, fy) | 0; fisheye (equidistant): mean2d = f * (x, y) * atan2(r, z) / r + c with the closed-form Jacobian —
 *    without the clamp; everything after the 2D covariance (low-pass, radius, rect, culling) is shared.
 * ---------------------------------------------------------------------------------------- */
int gspl_project_fwd(int C, int N,
                     const float* means, const float* scales, const float* quats,
                     const float* viewmats, const float* Ks,
                     int width, int height, int tile_size,
                     float scale_modifier, float eps2d, float near_plane, float far_plane,
                     float radius_clip, int camera_model,
                     int32_t* radii, float* means2d, float* depths, float* conics,
                     float* compensations /*nullable*/, int32_t* tiles_hit /*nullable*/, float* cov3d /*nullable*/,
                     void* stream);

/*    Backward of the above (autograd of gsplat's op; reference enters it through
 *    `manual_backward`, internal/gaussian_splatting.py:380).  v_means/v_scales/v_quats are
 *    OVERWRITTEN when C == 1 and must be zero-initialised by the caller when C > 1
 *    (accumulated with atomics across cameras).  v_compensations / v_depths nullable.  The strides let the
 *    columns of gspl_composite_bwd_packed's row buffer be consumed in place (C == 1 only). */
int gspl_project_bwd(int C, int N,
                     const float* means, const float* scales, const float* quats,
                     const float* viewmats, const float* Ks,
                     int width, int height,
                     float scale_modifier, float eps2d, int camera_model,
                     const int32_t* radii,
                     const float* v_means2d, int v_means2d_stride /* floats per row; 0 = dense [.,2] */,
                     const float* v_depths /*nullable = 0*/,
                     const float* v_conics, int v_conics_stride /* 0 = dense [.,3] */,
                     const float* v_compensations /*nullable*/,
                     float* v_means, float* v_scales, float* v_quats,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * 2. Spherical harmonics -> colour.
 *    Replaces gsplat `spherical_harmonics` / `spherical_harmonics_decomposed`
 *    (gsplat_renderer.py:105, gsplat_v1_renderer.py:121-130, gsplat_distributed_renderer.py:416-421);
 *    basis pinned to internal/utils/sh_utils.py:57-171.
 *    Coefficient k of Gaussian n, channel c:
 *        k == 0 : dc  [n*dc_stride   + c]
 *        k >= 1 : rest[n*rest_stride + (k-1)*3 + c]
 *    so the merged [N,K,3] tensor is (dc=base, rest=base+3, both strides 3K) and the decomposed
 *    model storage (shs_dc [N,1,3], shs_rest [N,K-1,3]) is (3, 3(K-1)).
 *    dirs [N,3] need not be normalised (normalised in-kernel, as gsplat does); if `origin`
 *    (device [3], nullable) is given, the direction is dirs[n] - origin (dirs = means).
 *    mask u8 [N] nullable (0 -> colour 0, no gradient).
 *    flags bit0: add 0.5 and clamp at 0 (gsplat_renderer.py:106) inside the kernel;
 *    clamped u8 [N,3] (nullable) then records which channels were clamped (needed by bwd).
 * ---------------------------------------------------------------------------------------- */
enum { GSPL_SH_ADD_HALF_CLAMP = 1 };
int gspl_sh_fwd(int N, int degree,
                const float* dirs, const float* origin /*nullable*/,
                const float* dc, int dc_stride, const float* rest, int rest_stride,
                const uint8_t* mask /*nullable*/, int flags,
                float* colors, uint8_t* clamped /*nullable*/,
                void* stream);
/*    v_dc / v_rest are written with the same strides (every coefficient of every Gaussian is
 *    written, zeros above the active degree / for masked rows, n_coeffs = number of coefficient
 *    rows present per Gaussian); v_dirs [N,3] nullable (gradient w.r.t. dirs, i.e. w.r.t. means
 *    when origin is used — the Inria path needs it, gsplat paths detach: gsplat_renderer.py:104). */
int gspl_sh_bwd(int N, int degree, int n_coeffs,
                const float* dirs, const float* origin /*nullable*/,
                const float* dc, int dc_stride, const float* rest, int rest_stride,
                const uint8_t* mask /*nullable*/, int flags, const uint8_t* clamped /*nullable*/,
                const float* v_colors, int v_colors_stride /* floats per row; 0 = dense [N,3] */,
                float* v_dc, float* v_rest, float* v_dirs /*nullable*/,
                void* stream);

/* C cameras in ONE launch — what the Gaussian-sharded renderer needs per step (every rank evaluates its shard for all W
 * cameras: gsplat_distributed_renderer.py:252-311,416-421, a Python loop of W SH calls there).  The coefficient rows are
 * read once for all cameras; colours = clamp(SH(means - origins[c]) + 0.5, 0) with flags = GSPL_SH_ADD_HALF_CLAMP.
 *    means [N,3]; origins [C,3]; radii [C,N] i32 nullable (rows with radius <= 0 are skipped and get zeros);
 *    colors [C,N,3]; clamped [C,N,3] u8 (nullable without the clamp flag).
 * Backward: v_colors [C,N,3] dense -> v_dc / v_rest (same layouts as gspl_sh_bwd) summed over the cameras and written
 * once; no direction gradient (the reference detaches the means here). */
int gspl_sh_fwd_batched(int C, int N, int degree,
                        const float* means, const float* origins,
                        const float* dc, int dc_stride, const float* rest, int rest_stride,
                        const int32_t* radii /*nullable*/, int flags,
                        float* colors, uint8_t* clamped /*nullable*/, void* stream);
int gspl_sh_bwd_batched(int C, int N, int degree, int n_coeffs,
                        const float* means, const float* origins,
                        int dc_stride, int rest_stride,
                        const int32_t* radii /*nullable*/, int flags, const uint8_t* clamped /*nullable*/,
                        const float* v_colors, float* v_dc, float* v_rest, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3. Tile binning: (tile, depth) keys, radix sort, per-tile ranges.
 *    Replaces gsplat `isect_tiles` + `isect_offset_encode` (gsplat_v1_renderer.py:446-458) and
 *    the binning half of v0 `rasterize_gaussians` (gsplat_renderer.py:86-99); key layout pinned
 *    to internal/utils/gaussian_projection.py:159-208: key = (tile_id << 32) | bits(depth),
 *    tile_id row-major, value = Gaussian index.  Single camera per call.
 *
 *    Step a: per-Gaussian tile counts + inclusive prefix sum (i64; block sums, scan of the sums, per-block scan).  Host reads cum[N-1].
 *    Step b: emit keys, sort them (stable LSD radix, so equal-depth ties keep Gaussian order).
 *    Step c: offsets[t] = first sorted index whose tile id is >= t   (t in [0, tiles)).
 *    gspl_isect_workspace_bytes gives the scratch size for steps a and b.
 * ---------------------------------------------------------------------------------------- */
size_t gspl_isect_workspace_bytes(int N, int64_t n_isects);
int gspl_isect_count(int N, int mode,
                     const float* means2d, const int32_t* radii,
                     int tile_size, int tile_w, int tile_h,
                     int32_t* tiles_per_gauss, int64_t* cum_tiles,
                     void* workspace, size_t workspace_bytes, void* stream);
int gspl_isect_emit_sort(int N, int mode,
                         const float* means2d, const int32_t* radii, const float* depths,
                         const int64_t* cum_tiles,
                         int tile_size, int tile_w, int tile_h, int64_t n_isects,
                         int64_t* isect_ids, int32_t* flatten_ids,
                         void* workspace, size_t workspace_bytes, void* stream);
int gspl_isect_offsets(int64_t n_isects, const int64_t* isect_ids,
                       int tile_w, int tile_h, int32_t* offsets, void* stream);

/*    3b. The same binning for callers that only need the per-tile lists (flatten_ids + offsets), not
 *    the 64-bit keys: the fused Inria rasterizer (vanilla_renderer.py:111-120) and gsplat-v0
 *    `rasterize_gaussians` (gsplat_renderer.py:86-99) never see isect_ids.  Two-level, "depth first":
 *    sort the N splats by depth (32-bit keys), emit tile hits in that order, stable-sort by tile id
 *    only.  Produces exactly the flatten_ids / offsets of steps a-c above at a third of the traffic.
 *      order [N] i32 (splat ids by depth, tile-less splats last), cum_tiles [N] i64 (inclusive prefix
 *      sum of tile counts in that order; host reads cum_tiles[N-1]).
 *    conics [N,3] + opacities [N] (both nullable, together): when given, a (tile, splat) pair is listed
 *    only if the splat can reach alpha >= 1/255 somewhere in the tile (exact ellipse-vs-tile test with a
 *    conservative margin).  The dropped pairs contribute nothing to any pixel, so composited images and
 *    gradients are unchanged, while the lists shrink by ~40 % (the 3-sigma square of the reference's
 *    rect is loose, the more so for low opacities).  Pass the SAME opacities the compositing call uses.
 *    spans: caller-owned scratch of GSPL_BIN_SPAN_BYTES per splat, written by gspl_bin_count (the reachable tile
 *    columns of each tile row of the splat: N records for rows 0..7, then N records for rows 8..15 that only taller
 *    splats touch) and read back, in depth order, by gspl_bin_emit_sort — one 32-byte line per splat instead of
 *    re-gathering means2d / radii / conics / opacities at random addresses.
 * ---------------------------------------------------------------------------------------- */
enum { GSPL_BIN_SPAN_BYTES = 64 };
size_t gspl_bin_workspace_bytes(int N, int64_t n_isects);
int gspl_bin_count(int N, int mode,
                   const float* means2d, const int32_t* radii, const float* depths,
                   const float* conics /*nullable*/, const float* opacities /*nullable*/,
                   int tile_size, int tile_w, int tile_h,
                   int32_t* order, int64_t* cum_tiles /* [N + 1]: inclusive scan of the tile counts in depth order, then
                                                         n_big = the number of splats spanning more than 16 tile rows
                                                         (radius > ~128 px: close-ups, sky blobs) or with a very wide row */,
                   int32_t* big_list /* [N]: the depth-order indices of those splats (n_big entries, ranked by the scan);
                                        the emission deals them out to its workgroups instead of leaving up to 64
                                        consecutive screen-filling splats to one wave */,
                   void* spans /* GSPL_BIN_SPAN_BYTES * N, 16-byte aligned */,
                   int64_t* host_counts /* nullable: two int64 of device-accessible HOST memory (hipHostMalloc / a pinned tensor);
                                           the last kernel stores cum_tiles[N-1] (the list length) and n_big there itself, so a
                                           host that records an event after this call and waits for it needs no copy */,
                   void* workspace, size_t workspace_bytes, void* stream);
int gspl_bin_emit_sort(int N, int mode,
                       const float* means2d, const int32_t* radii,
                       const float* conics /*nullable*/, const float* opacities /*nullable*/,
                       const int32_t* order, const int64_t* cum_tiles, const int32_t* big_list, const void* spans,
                       int tile_size, int tile_w, int tile_h, int64_t n_isects,
                       int32_t* flatten_ids, int32_t* offsets,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The two halves of gspl_bin_emit_sort, so that the emission can be launched SPECULATIVELY while the host still waits
 * for the list length: `capacity` (>= the length, if the guess is good) sizes the workspace
 * (gspl_bin_workspace_bytes(N, capacity)); records past it are dropped.  gspl_bin_sort then takes the real
 * n_isects <= capacity; when the guess was too low the caller repeats gspl_bin_emit with capacity = n_isects. */
int gspl_bin_emit(int N, int mode, const float* means2d, const int32_t* radii,
                  const float* conics /*nullable*/, const float* opacities /*nullable*/,
                  const int32_t* order, const int64_t* cum_tiles, const int32_t* big_list, const void* spans,
                  int tile_size, int tile_w, int tile_h, int64_t capacity,
                  void* workspace, size_t workspace_bytes, void* stream);
int gspl_bin_sort(int N, int tile_w, int tile_h, int64_t n_isects, int64_t capacity,
                  int32_t* flatten_ids, int32_t* offsets, void* workspace, size_t workspace_bytes, void* stream);
/* The same for a host that has not read the list length back yet: it stays on the device (n_isects_dev = &cum_tiles[N - 1]), the
 * sort's grid is sized by `capacity`.  flatten_ids: room for `capacity` ids; offsets: tile_w * tile_h + 1 entries, the last one
 * receives the list length — the compositing entry points take n_isects = -1 with such an array.  The caller compares the
 * length with `capacity` once it has it (gspl_bin_count's host_counts) and repeats emission and sort when the guess was too low. */
int gspl_bin_sort_device_count(int N, int tile_w, int tile_h, const int64_t* n_isects_dev, int64_t capacity,
                               int32_t* flatten_ids, int32_t* offsets, void* workspace, size_t workspace_bytes, void* stream);
/* Sorts and scans of this library are in-tree kernels (csrc/sort.hip): every radix pass is count -> digit-row scan -> scatter
 * over contiguous tile ranges, the scans are block sums -> scan of the sums -> per-block scan.  No workgroup waits for another,
 * so they make progress under any dispatch order and contention, and their output is bit-reproducible. */

/* ------------------------------------------------------------------------------------------
 * 4. Tile compositing, forward.
 *    Replaces gsplat `rasterize_to_pixels` (gsplat_v1_renderer.py:588-601), v0
 *    `rasterize_gaussians` (gsplat_renderer.py:86-99, pypreprocess_gsplat_renderer.py:45-58) and
 *    the render stage of the Inria `GaussianRasterizer` (vanilla_renderer.py:111-120).
 *    One 16x16 tile per workgroup (4 wave64), front-to-back.
 *      colors [N,D], D in {1,2,3,4,8}; backgrounds [D] nullable (treated as 0)
 *      offsets [tile_h*tile_w] i32, flatten_ids [n_isects] i32
 *    Outputs: out_colors ([H,W,D] or [D,H,W] by `layout`), out_alphas [H,W] (= 1 - T_final),
 *             final_Ts [H,W] (the final transmittance itself: backward divides by it, and
 *             1 - (1 - T) in fp32 would lose up to 6e-4 relative at T ~ 1e-4),
 *             last_ids [H,W] i32 (one past the index into flatten_ids of the last contributor;
 *             the tile's range start when the pixel has none).
 * ---------------------------------------------------------------------------------------- */
int gspl_composite_fwd(int N, int64_t n_isects, int D, int mode, int layout,
                       const float* means2d, const float* conics, const float* colors,
                       const float* opacities, const float* backgrounds /*nullable*/,
                       int width, int height, int tile_size, int tile_w, int tile_h,
                       const int32_t* offsets, const int32_t* flatten_ids,
                       float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids,
                       uint8_t* hit_flags /*nullable: [N], zeroed by the caller; 1 = some pixel composited the splat (the
                                            fork's has_hit_any_pixels / acc_vis, gsplat_v1_renderer.py:287)*/,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * 5. Tile compositing, backward (the graded kernel: SURVEY.md §8 a12).
 *    Back-to-front per pixel from last_ids; gradients are reduced over the 64 pixels of a wave
 *    with DPP, then over the 4 waves of the tile in LDS, then one fp32 L2 atomic per value per
 *    (tile, Gaussian).  All v_* outputs must be zero-initialised by the caller.
 *      v_out_colors laid out like out_colors; v_out_alphas [H,W] nullable.
 *      v_means2d_abs [N,2] nullable (gsplat `absgrad`, vanilla_density_controller.py:112-113).
 * ---------------------------------------------------------------------------------------- */
int gspl_composite_bwd(int N, int64_t n_isects, int D, int mode, int layout,
                       const float* means2d, const float* conics, const float* colors,
                       const float* opacities, const float* backgrounds /*nullable*/,
                       int width, int height, int tile_size, int tile_w, int tile_h,
                       const int32_t* offsets, const int32_t* flatten_ids,
                       const float* final_Ts, const int32_t* last_ids,
                       const float* v_out_colors, const float* v_out_alphas /*nullable*/,
                       float* v_means2d, float* v_means2d_abs /*nullable*/,
                       float* v_conics, float* v_colors, float* v_opacities,
                       uint8_t* hit_flags /*nullable: u8 [N], zero-initialised; set to 1 for every splat that some pixel
                                            composited (alpha >= 1/255 before termination) — the fork's
                                            `means2d.has_hit_any_pixels`, internal/optimizers.py:39 */,
                       void* stream);

/*    5b. Same backward with the gradients delivered as ONE packed row per splat:
 *      v_packed [N, packed_stride], packed_stride >= 6 + D (+2 when absgrad != 0); 16 makes every row one 64-B line:
 *        (dL/dx, dL/dy, dL/dconic a, b, c, dL/dopacity, dL/dcolour[0..D), [sum|dL/dx|, sum|dL/dy|]),
 *    zero-initialised by the caller.  One atomic instruction then covers contiguous components of a few
 *    splat rows instead of 64 scattered dwords. */
int gspl_composite_bwd_packed(int N, int64_t n_isects, int D, int mode, int layout,
                              const float* means2d, const float* conics, const float* colors,
                              const float* opacities, const float* backgrounds /*nullable*/,
                              int width, int height, int tile_size, int tile_w, int tile_h,
                              const int32_t* offsets, const int32_t* flatten_ids,
                              const float* final_Ts, const int32_t* last_ids,
                              const float* v_out_colors, const float* v_out_alphas /*nullable*/,
                              float* v_packed, int packed_stride /* floats per row, >= 6+D(+2) */, int absgrad,
                              uint8_t* hit_flags /*nullable*/, void* stream);

/* Deterministic (debug) mode of gspl_composite_bwd_packed (and of everything built on it: the fused Inria backward, the staged
 * rasterizers): the per-splat gradient rows are added up in list order instead of by fp32 atomics in dispatch order, so two runs give
 * the same bits.  Costs three extra passes over the list entries and stream-ordered scratch (hipMallocAsync); needs the list length
 * on the host (n_isects >= 0) and 16-pixel list tiles.  Returns the previous setting.  Process-wide.  Additive entries. */
int gspl_set_deterministic(int on);
int gspl_get_deterministic(void);
/* Name of the kernel template the two backward entry points launch in this build (profile look-ups in bench.py). */
const char* gspl_composite_bwd_kernel_name(void);

/* ------------------------------------------------------------------------------------------
 * 6. Inria-convention preprocess (the front half of the fused `GaussianRasterizer`).
 *    Replaces `diff_gaussian_rasterization.GaussianRasterizer.forward` up to the sort
 *    (vanilla_renderer.py:62-77,111-120): frustum cull (view z <= 0.2), cov3D from scale/rotation
 *    (or cov3D_precomp [N,6]), EWA cov2D + 0.3 low-pass, conic, radius, NDC->pixel mean,
 *    SH->RGB with +0.5 / clamp (or colors_precomp [N,3]).
 *      viewmatrix [16], projmatrix [16]: the reference's transposed (row-vector) 4x4s, device
 *      (internal/cameras/cameras.py:147-189); campos [3] device.
 *    Outputs: radii i32 [N], means2d [N,2] (pixels, integer-centred), depths [N], conics [N,3],
 *             colors [N,3], clamped u8 [N,3], cov3d [N,6] (saved for backward).
 *    `phases` selects the geometry kernel, the colour kernel (which reads the radii the geometry
 *    kernel wrote), or both; a caller that bins between the two hides the sort-size read-back of
 *    gspl_bin_count behind the colour kernel.
 * ---------------------------------------------------------------------------------------- */
enum { GSPL_INRIA_GEOMETRY = 1, GSPL_INRIA_COLOURS = 2, GSPL_INRIA_ALL = 3 };
int gspl_inria_preprocess_fwd(int N, int degree, int n_coeffs,
                              const float* means, const float* scales /*nullable*/,
                              const float* quats /*nullable*/, const float* cov3d_precomp /*nullable*/,
                              const float* shs /*[N,n_coeffs,3] nullable; with shs_rest: the DC rows [N,1,3]*/,
                              const float* shs_rest /*nullable [N,n_coeffs-1,3]: the reference's model keeps the coefficients as two
                                                      parameters, shs_dc and shs_rest (vanilla_gaussian.py:266-300, `get_shs` = a
                                                      torch.cat per step); given both, they are read where they are*/,
                              const float* colors_precomp /*nullable*/,
                              const float* viewmatrix, const float* projmatrix, const float* campos,
                              int width, int height, int tile_size,
                              float tanfovx, float tanfovy, float scale_modifier,
                              int32_t* radii, float* means2d, float* depths, float* conics,
                              float* colors, uint8_t* clamped, float* cov3d,
                              float* sh_jac /* nullable [N,9]: the colour kernel leaves d colour / d (unit view direction) here, and
                                               a backward that is handed it finds the view-direction part of v_means without
                                               reading the 12 n_coeffs bytes of coefficients per splat again */,
                              int phases /* GSPL_INRIA_GEOMETRY | GSPL_INRIA_COLOURS */,
                              void* stream);
/*    Backward.  v_means2d is the composite kernel's pixel-unit gradient [N,2]; the returned
 *    v_means2d_ndc [N,3] is what the reference exposes as `viewspace_points.grad`
 *    (pixel gradient x 0.5*W, 0.5*H; vanilla_renderer.py:55-56).
 *    v_shs [N,n_coeffs,3], v_cov3d_precomp [N,6], v_colors_precomp nullable as their inputs. */
int gspl_inria_preprocess_bwd(int N, int degree, int n_coeffs,
                              const float* means, const float* scales /*nullable*/,
                              const float* quats /*nullable*/, const float* cov3d /*[N,6] from fwd*/,
                              const float* shs /*nullable*/, const float* shs_rest /*nullable, as in the forward*/,
                              const float* viewmatrix, const float* projmatrix, const float* campos,
                              int width, int height,
                              float tanfovx, float tanfovy, float scale_modifier,
                              const int32_t* radii, const uint8_t* clamped,
                              const float* v_means2d, const float* v_conics, const float* v_colors,
                              int grad_stride /* 0: dense [N,2],[N,3],[N,3]; k: columns of one packed [N,k] buffer */,
                              float* v_means, float* v_scales /*nullable*/, float* v_quats /*nullable*/,
                              float* v_cov3d_precomp /*nullable*/, float* v_shs /*nullable*/,
                              float* v_shs_rest /*nullable; given iff shs_rest is: v_shs is then [N,1,3], this one [N,n_coeffs-1,3]*/,
                              float* v_colors_precomp /*nullable*/, float* v_means2d_ndc,
                              const float* v_opacities_packed /*nullable: the opacity column of the packed buffer */,
                              float* v_opacities /*nullable: [N], receives that column densely (the optimizer's
                                                   gradient is then a contiguous tensor, not a strided view) */,
                              const float* sh_jac /*nullable: from the forward*/,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * 6b. The WHOLE Inria rasterizer in one call per direction ("fused gs_rasterize_vanilla_fwd/bwd, Inria
 *    argument list"): what `diff_gaussian_rasterization`'s `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` are to
 *    the reference's wrapper (call site internal/renderers/vanilla_renderer.py:62-120).  Preprocess (geometry, then colours on
 *    `side_stream` next to the depth sort), list-only binning with lossless tile culling and the one host read-back of the
 *    frame (waited for INSIDE the call), compositing.  Memory comes from the caller through `alloc` — the Inria library's own
 *    resize-call-back pattern: the returned device pointers must stay valid until the matching backward has run (the Python
 *    side keeps the torch byte tensors in the autograd context).  Tags tell the call-back what a block is for; one block per tag
 *    and call, except GSPL_BUF_LISTS_WORK which may be asked for twice (speculative emission with `capacity_hint` = a guess of
 *    the list length, e.g. the previous frame's x 1.25; 0 = no speculation).  No hipMalloc, no global state.
 *    out_color [3,H,W], radii [N] are caller-allocated outputs; `state` receives the pointers the backward needs and
 *    n_isects (the list length — feed it back as the next frame's hint); its `flags` field is read BEFORE it is filled.
 * ---------------------------------------------------------------------------------------- */
enum { GSPL_BUF_GEOMETRY = 1, GSPL_BUF_BINNING = 2, GSPL_BUF_IMAGE = 3, GSPL_BUF_LISTS_WORK = 4, GSPL_BUF_LISTS = 5,
       GSPL_BUF_CHECKPOINTS = 6 /* segmented backward: 4 KB per 256 list entries of capacity; kept until the backward like IMAGE / LISTS */,
       GSPL_BUF_PACKED = 7 /* ABI 35, only with GSPL_INRIA_WILL_BACKWARD: 36 N bytes (rounded up to 16), the backward's per-splat rows
                              x y | a b c | opacity | r g b, CLEARED by the forward's compositing kernel — hand it to the backward as `packed` */ };
typedef void* (*gspl_alloc_fn)(void* ctx, int tag, size_t bytes);      /* device memory, 256-byte aligned; NULL = failure */
typedef struct gspl_inria_state {
    int N, width, height;
    int64_t n_isects;
    float* means2d; float* depths; float* conics; float* colors; uint8_t* clamped; float* cov3d; float* sh_jac;      /* GSPL_BUF_GEOMETRY */
    float* alphas; float* final_Ts; int32_t* last_ids; int32_t* offsets;                               /* GSPL_BUF_IMAGE */
    int32_t* flatten_ids;                                                                              /* GSPL_BUF_LISTS */
    float* opacities;      /* GSPL_BUF_GEOMETRY: the opacities compositing read — the caller's tensor, or with GSPL_INRIA_RAW_PARAMS
                              sigmoid(raw) [N] as the forward stored it */
    int flags;             /* IN (forward; the backward reads it back): 0 — a zeroed struct — or GSPL_INRIA_RAW_PARAMS | GSPL_INRIA_NO_SEGMENTS */
    /* segmented backward (ABI 33): per-pixel checkpoints the forward left every 256 list entries and the words
     * [count | - | work items seg_slots] in front of them (one GSPL_BUF_CHECKPOINTS block); seg_ckpt == NULL: this frame is not segmented */
    void* seg_ckpt; uint32_t* seg_words; uint32_t seg_slots; uint32_t seg_reserved;
    /* densification statistics inside the backward (ABI 34; IN, read by gspl_rasterize_inria_bwd / _bwd_adam only — the forward clears
     * them, the caller sets them between the two calls): not NULL = the preprocess-backward kernel applies section 11's update to the
     * rows it has just produced the screen-space gradient of (visible = radii > 0, no scale, v_means2D_ndc as `grad`), with the same
     * arithmetic as gspl_densify_stats — the density controller's launch after the backward goes away.  accum and denom go together. */
    float* stats_accum; float* stats_denom; float* stats_max_radii;
} gspl_inria_state;
/* GSPL_INRIA_RAW_PARAMS: `scales`, `rotations`, `opacities` are the model's RAW parameters and the activations of the reference's
 *    model — scale_activation = exp, rotation_activation = F.normalize (x / max(|x|, 1e-12)), opacity_activation = sigmoid
 *    (internal/models/vanilla_gaussian.py:345-358, applied by `get_scaling` / `get_rotation` / `get_opacity` in
 *    vanilla_renderer.py:62-77 before every render, differentiated by autograd after every backward) — run inside the preprocess
 *    kernels; the backward returns the gradients of the raw parameters.  Needs scales + rotations (no cov3D_precomp). */
enum { GSPL_INRIA_RAW_PARAMS = 1,
       GSPL_INRIA_NO_SEGMENTS = 2 /* never: the backward walks every tile's list with one workgroup, however long */,
       GSPL_INRIA_FORCE_SEGMENTS = 4 /* always take checkpoints (default: only while walks longer than a segment are being met) */,
       GSPL_INRIA_WILL_BACKWARD = 8 /* ABI 35, IN: a backward will follow — the forward asks `alloc` for GSPL_BUF_PACKED and its compositing
                                       kernel (VALU-bound, the memory system idle) clears it, instead of a 7 us fill command in front of the
                                       backward's first kernel */,
       GSPL_INRIA_PACKED_READY = 16 /* OUT (set by the forward in state->flags): the GSPL_BUF_PACKED block is cleared and the backward will NOT
                                       clear its `packed` argument — which must be that block */ };
size_t gspl_rasterize_inria_geometry_bytes(int N);
size_t gspl_rasterize_inria_image_bytes(int width, int height);
size_t gspl_inria_state_bytes(void);      /* sizeof(gspl_inria_state) as the library was built: a binding checks its own layout against it */
int gspl_rasterize_inria_fwd(int N, int degree, int n_coeffs,
                             const float* means3D, const float* scales /*nullable*/, const float* rotations /*nullable*/,
                             const float* cov3D_precomp /*nullable*/, const float* shs /*nullable*/,
                             const float* shs_rest /*nullable: see gspl_inria_preprocess_fwd*/, const float* colors_precomp /*nullable*/,
                             const float* opacities,
                             const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
                             int width, int height, float tanfovx, float tanfovy, float scale_modifier,
                             gspl_alloc_fn alloc, void* alloc_ctx, int64_t capacity_hint,
                             float* out_color, int32_t* radii, gspl_inria_state* state,
                             void* stream, void* side_stream /*nullable: colours on `stream`*/);
/*    Backward: packed [N,9] f32 scratch and hit_flags [N] u8 (nullable) are cleared inside; every v_* is caller-allocated
 *    ([N,3], [N,3] NDC-scaled, [N,n_coeffs,3] (or [N,1,3] and [N,n_coeffs-1,3] with shs_rest) | [N,3], [N], [N,3], [N,4] | [N,6]; the ones that do not apply are NULL). */
int gspl_rasterize_inria_bwd(int degree, int n_coeffs,
                             const float* means3D, const float* scales, const float* rotations, const float* shs, const float* shs_rest,
                             const float* opacities,
                             const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
                             float tanfovx, float tanfovy, float scale_modifier,
                             const int32_t* radii, const gspl_inria_state* state, const float* v_out_color,
                             float* packed, uint8_t* hit_flags /*nullable*/,
                             float* v_means3D, float* v_means2D_ndc, float* v_shs, float* v_shs_rest, float* v_colors_precomp, float* v_opacities,
                             float* v_scales, float* v_rotations, float* v_cov3D, void* stream);

/* The backward with the OPTIMIZER INSIDE (additive entry, round 5): the per-Gaussian kernels that end the backward apply the Adam update
 * to the rows they have just produced the gradient of — moments read and written once, parameters written once, NO parameter gradient in
 * HBM (236 B per Gaussian written by the backward and read back by the optimizer launch otherwise: the two-kernel form of
 * internal/optimizers.py:14-22 / internal/models/vanilla_gaussian.py:266-300 behind gaussian_splatting.py:380-397).
 * means3D ... opacities are the PARAMETERS, updated in place (raw parameters with GSPL_INRIA_RAW_PARAMS in the state's flags, else
 * the activated tensors the forward was given); every row is updated, a Gaussian the frame does not see with a zero gradient
 * (torch.optim.Adam semantics).  shs_rest == NULL: `shs` holds all n_coeffs rows and plan->shs its moments.  scratch_means [N,3] f32
 * and packed [N,9] are scratch; v_means2D_ndc [N,3] (the screen-space gradient) is the one gradient written.
 * Same arithmetic as gspl_rasterize_inria_bwd followed by gspl_selective_adam with no mask (bit-identical parameters and moments for
 * identical gradients: tests/test_fused_backward_adam.py). */
typedef struct gspl_bwd_adam_tensor {
    float* exp_avg;              /* first / second moment, shape of the parameter, f32 contiguous */
    float* exp_avg_sq;
    float lr, beta1, beta2, eps;
    float bias_correction1;      /* 1 - beta1^t   (1: gsplat's uncorrected update) */
    float bias_correction2_sqrt; /* sqrt(1 - beta2^t)   (1: uncorrected) */
} gspl_bwd_adam_tensor;
typedef struct gspl_bwd_adam_plan {
    gspl_bwd_adam_tensor means, scales, rotations, opacities, shs, shs_rest;
} gspl_bwd_adam_plan;
int gspl_rasterize_inria_bwd_adam(int degree, int n_coeffs,
                                  float* means3D, float* scales, float* rotations, float* shs, float* shs_rest /*nullable*/, float* opacities,
                                  const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
                                  float tanfovx, float tanfovy, float scale_modifier,
                                  const int32_t* radii, const gspl_inria_state* state, const float* v_out_color,
                                  float* packed, uint8_t* hit_flags /*nullable*/, float* scratch_means, float* v_means2D_ndc,
                                  const gspl_bwd_adam_plan* plan /* host */, void* stream);

/* A per-device stream of the LOWEST priority the device offers, created on first use and kept: a `side_stream` for
 * gspl_rasterize_inria_fwd whose colour kernel then yields to the kernels on the caller's stream.  NULL on failure. */
void* gspl_low_priority_stream(void);

/* Timing of the compositing launches inside the fused calls (bench.py: the roofline of the graded kernel needs its launch
 * duration from HIP events on the launch stream, and the launches are no longer visible from the host language).
 * gspl_profile_enable(k) starts recording every k-th launch (k = 1: all; an event pair costs the stream ~6 us of idle time on
 * either side of the launch) and drops what was recorded, gspl_profile_read synchronises and returns the count and the summed
 * duration of the timed forward (which = 0) or backward (1) compositing launches since then; enable(0) stops. */
int gspl_profile_enable(int period);
int gspl_profile_enable2(int period_fwd, int period_bwd);      /* the two directions apart (0 = that direction is not timed) */
int gspl_profile_read(int which, int* count, float* total_ms);

/* ------------------------------------------------------------------------------------------
 * 6c. Visible-splat records of the Gaussian-sharded multi-GPU renderer: pack / unpack, forward and backward.
 *    Replaces the per-camera `torch.concat` + boolean-mask selection before, and the `torch.split` after, the all-to-all of
 *    internal/renderers/gsplat_distributed_renderer.py:313-414.  Record = 12 fp32 (48 B):
 *        [x, y, depth, conic a, b, c, compensation, opacity, r, g, b, radius (int32 bits)]
 *    pack: every (camera, local splat) with radius > 0 -> one record, grouped by camera (= destination rank), splat order kept.
 *      records: room for C*N rows; slots [C,N] i32 = row of the pair's record or -1 (the backward's route); ends [C] i64 =
 *      one past each camera's last row (device), host_ends (nullable) the same in pinned host memory, stored by the kernel.
 *    pack_bwd: writes EVERY row of every gradient tensor (zeros where invisible); v_opacities [N] summed over the cameras.
 *    unpack: received records -> per-quantity tensors; fold_compensation: opacities = opacity x compensation (anti-aliased
 *      mode), with the product rule applied by unpack_bwd.  Strides (floats per row, 0 = dense) let unpack_bwd read the columns
 *      of gspl_composite_bwd_packed's row buffer in place.
 * ---------------------------------------------------------------------------------------- */
#define GSPL_RECORD_FLOATS 12
size_t gspl_records_workspace_bytes(int C, int N);
int gspl_records_pack_fwd(int C, int N, const int32_t* radii, const float* means2d, const float* depths, const float* conics,
                          const float* compensations /*nullable = 1*/, const float* opacities /*[N]*/, const float* colors /*[C,N,3]*/,
                          float* records, int32_t* slots, int64_t* ends, int64_t* host_ends /*nullable*/,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The pack in two phases (same slots, ends and records): COUNT needs the radii only — the per-camera ends are on their way to the
 * host (`host_ends`, pinned) before the colours of the frame exist, so the exchange of the counts (gsplat_distributed_renderer.py:
 * 141-160) overlaps the colour kernel —, SCATTER writes the records to their slots.  Workspace as gspl_records_pack_fwd (COUNT only).
 * Additive entries (no existing signature changed: the ABI version stays). */
int gspl_records_count_fwd(int C, int N, const int32_t* radii, int32_t* slots, int64_t* ends, int64_t* host_ends /*nullable*/,
                           void* workspace, size_t workspace_bytes, void* stream);
int gspl_records_scatter_fwd(int C, int N, const int32_t* radii, const int32_t* slots, const float* means2d, const float* depths,
                             const float* conics, const float* compensations /*nullable = 1*/, const float* opacities /*[N]*/,
                             const float* colors /*[C,N,3]*/, float* records, void* stream);
/* The fixed-size exchange format: one record per (camera, local splat), camera-major, rows of invisible splats zeroed (radius 0
 * keeps them out of the receiver's lists); slots[i] = i for the visible rows, -1 otherwise (input of gspl_records_pack_bwd).  No
 * count, no workspace, nothing the host has to wait for: every size of the exchange is known before the step starts. */
int gspl_records_pad_fwd(int C, int N, const int32_t* radii, const float* means2d, const float* depths, const float* conics,
                         const float* compensations /*nullable = 1*/, const float* opacities /*[N]*/, const float* colors /*[C,N,3]*/,
                         float* records /*[C*N,12]*/, int32_t* slots /*[C,N]*/, void* stream);
int gspl_records_pack_bwd(int C, int N, const int32_t* slots, const float* v_records,
                          float* v_means2d, float* v_depths, float* v_conics, float* v_compensations /*nullable*/, float* v_opacities,
                          float* v_colors, void* stream);
int gspl_records_unpack_fwd(int64_t M, int fold_compensation, const float* records, int32_t* radii, float* means2d, float* depths,
                            float* conics, float* opacities, float* colors, void* stream);
int gspl_records_unpack_bwd(int64_t M, int fold_compensation, const float* records,
                            const float* v_means2d /*nullable = 0*/, int v_means2d_stride, const float* v_depths /*nullable*/,
                            const float* v_conics /*nullable*/, int v_conics_stride, const float* v_opacities /*nullable*/, int v_opacities_stride,
                            const float* v_colors /*nullable*/, int v_colors_stride, float* v_records, void* stream);

/*    Direct peer-to-peer transport of the records (csrc/peer.hip): instead of an all-to-all collective, every rank writes its rows
 *    straight into the receive buffer of the destination rank — device memory of the peer process mapped through HIP IPC, reached
 *    over xGMI (or the same GPU) — and raises one flag word per destination; the receiver's stream waits for its flag words.
 *    Replaces torch.distributed.nn.functional.all_to_all of gsplat_distributed_renderer.py:141-202 in the per-step exchange.
 *      gspl_peer_alloc   fine-grained device memory (remote stores are visible without an L2 invalidation) + its 64-byte IPC handle,
 *                        which the host sends to the peers once (e.g. torch.distributed.all_gather_object at training_setup)
 *      gspl_peer_open    maps a peer's buffer from its handle (gspl_peer_close unmaps; gspl_peer_free releases an own buffer)
 *      gspl_peer_put_rows  one launch: rows [row_begin[d], row_begin[d+1]) of `rows` -> dst[d] (16-byte stores; up to 16 destinations)
 *      gspl_peer_signal  one launch behind it: system-scope release fence, then `value` into every flag word
 *      gspl_peer_wait    one launch on the receiver: polls its n_src flag words until all are >= value; gives up after max_polls
 *                        polls and stores 1 + source into *error (a lost peer must not hang the GPU; the host checks the word)
 *    Additive entries (no existing signature changed: the ABI version stays). */
int gspl_peer_alloc(size_t bytes, void** ptr, void* handle_out /* 64 bytes */);
int gspl_peer_open(const void* handle /* 64 bytes */, void** ptr);
int gspl_peer_close(void* ptr);
int gspl_peer_free(void* ptr);
int gspl_peer_put_rows(int n_dst, const float* rows, const int64_t* row_begin /* host, [n_dst + 1] */,
                       void* const* dst /* host, [n_dst] device pointers */, int floats_per_row /* multiple of 4 */, void* stream);
int gspl_peer_signal(int n_dst, void* const* flags /* host, [n_dst] device pointers to 8-byte words */, uint64_t value, void* stream);
int gspl_peer_wait(const uint64_t* flags /* device, [n_src] */, int n_src, uint64_t value, uint64_t max_polls,
                   int32_t* error /* device-visible word: device memory, or PINNED HOST memory the host can read without synchronising */, void* stream);

/* ------------------------------------------------------------------------------------------
 * 7. Mean squared distance to the 3 nearest neighbours ("next" row SURVEY.md §8f rank 1).
 *    Replaces `simple_knn._C.distCUDA2` at its one call site, the initial scales of
 *    `VanillaGaussianModel.setup_from_pcd` (internal/models/vanilla_gaussian.py:122-125):
 *        out[i] = mean over the 3 nearest OTHER points j of |p_i - p_j|^2     (fp32)
 *    points [N,3] f32, out [N] f32.  Exact (uniform grid + expanding shells).  With fewer than three
 *    other points the mean is taken over those that exist (0 for a single point).
 * ---------------------------------------------------------------------------------------- */
size_t gspl_knn_workspace_bytes(int N);
int gspl_knn3_mean_dist2(int N, const float* points, float* out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * 8. Fused photometric loss terms ("next" row SURVEY.md §8f rank 2): mean |x - y| and mean SSIM.
 *    Replaces `l1_loss` + `ssim` (internal/utils/ssim.py:17-63) and the opt-in `fused_ssim` package
 *    (internal/metrics/vanilla_metrics.py:35-39) in  loss = (1-l) L1 + l (1 - SSIM)  (:57-70).
 *    SSIM: 11x11 Gaussian window (sigma 1.5, separable), zero padding, C1 = 0.01^2, C2 = 0.03^2, mean
 *    over all planes * H * W elements.  img1, img2: [planes, H, W] f32 contiguous (planes = batch * channels).
 *      fwd: out_means[2] = (mean |img1-img2|, mean SSIM); dm_* [planes,H,W] (all three or none): the
 *           derivative maps the backward needs (pass NULL for evaluation only).
 *      bwd: v_img1 = weight_l1 * v_l1_mean * d(L1)/d(img1) + weight_ssim * v_ssim_mean * d(SSIM)/d(img1);
 *           v_*_mean are device scalars (NULL = 1); dm_* NULL drops the SSIM term.
 * ---------------------------------------------------------------------------------------- */
size_t gspl_loss_workspace_bytes(int planes, int H, int W);
int gspl_loss_l1_ssim_fwd(int planes, int H, int W, const float* img1, const float* img2,
                          float* out_means, float* dm_dmu1 /*nullable*/, float* dm_ds1, float* dm_ds12,
                          void* workspace, size_t workspace_bytes, void* stream);
/*    The training loss in one go: out_terms[3] = (mean|x-y|, mean SSIM, weight_l1 * L1 + weight_ssim * (1 - SSIM)),
 *    i.e. vanilla_metrics.py:66-68 with weight_l1 = 1 - lambda_dssim, weight_ssim = lambda_dssim.  Its backward is
 *    gspl_loss_l1_ssim_bwd with both upstream pointers = dL/dloss and weights (weight_l1, -weight_ssim). */
int gspl_loss_photometric_fwd(int planes, int H, int W, const float* img1, const float* img2,
                              float weight_l1, float weight_ssim, float* out_terms,
                              float* dm_dmu1 /*nullable*/, float* dm_ds1, float* dm_ds12,
                              void* workspace, size_t workspace_bytes, void* stream);
int gspl_loss_l1_ssim_bwd(int planes, int H, int W, const float* img1, const float* img2,
                          const float* dm_dmu1 /*nullable*/, const float* dm_ds1, const float* dm_ds12,
                          const float* v_l1_mean /*nullable*/, const float* v_ssim_mean /*nullable*/,
                          float weight_l1, float weight_ssim, float* v_img1, void* stream);

/* ------------------------------------------------------------------------------------------
 * 9. Visibility-masked fused Adam over all per-Gaussian tensors ("next" row SURVEY.md §8f rank 3).
 *    Replaces gsplat `SelectiveAdam` (internal/optimizers.py:26-58) and, with visible = NULL and the
 *    bias corrections of step t, the per-property `torch.optim.Adam` steps (vanilla_gaussian.py:266-300).
 *    For every element of every row n with visible[n] != 0 (NULL: all rows):
 *        m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *        p -= (lr / bias_correction1) * m / (sqrt(v) / bias_correction2_sqrt + eps)
 *    (bias_correction1 = 1 - b1^t, bias_correction2_sqrt = sqrt(1 - b2^t); pass 1, 1 for gsplat's
 *    uncorrected update).  Masked rows keep parameter and moments.  All tensors are [N, row_elems]
 *    f32, contiguous, 16-byte aligned; up to GSPL_ADAM_MAX_TENSORS per call, one launch.
 * ---------------------------------------------------------------------------------------- */
enum { GSPL_ADAM_MAX_TENSORS = 16 };
typedef struct gspl_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    float lr;
    int32_t row_elems;
} gspl_adam_tensor;
int gspl_selective_adam(int n_tensors, const gspl_adam_tensor* tensors /* host array */, int N,
                        const uint8_t* visible /*nullable, device [N]*/,
                        float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt,
                        void* stream);
/* The same update with at most `max_blocks` workgroups per tensor (0 = no limit): for a launch that shares the device with the
 * kernels of another stream (optimizers.FusedAdam(deferred=...): the shs_rest update next to the following frame's binning). */
int gspl_selective_adam_limited(int n_tensors, const gspl_adam_tensor* tensors, int N, const uint8_t* visible /*nullable*/,
                                float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt,
                                int max_blocks, void* stream);

/* ------------------------------------------------------------------------------------------
 * 10. Stable LSD radix sort of the binning stage, exported for the parity tests.
 *    The depth sort inside gspl_bin_count (depth keys + splat ids, u32 pairs, first pass counted by the key pass
 *    itself) runs on this sort (csrc/sort.hip: count -> scatter per pass, no inter-workgroup waiting); the u64 keys-only entry
 *    point is the same kernels at the record width of the tile sort (gspl_bin_sort runs them with a last pass that writes
 *    ids + per-tile counts).  Both stand in for the
 *    cub::DeviceRadixSort::SortPairs calls of the reference's native rasterizers (gsplat `isect_tiles`
 *    behind gsplat_v1_renderer.py:524-556, the Inria rasterizer behind vanilla_renderer.py:111): stable,
 *    ascending on key bits [begin_bit, end_bit); at most 32 selected bits (4 passes of <= 8 bits), at most
 *    2^30-1 items.  Buffer 0 holds the input and is overwritten; buffer 1 is scratch of the same size; the
 *    sorted sequence ends in buffer *result_buffer (0 or 1).
 * ---------------------------------------------------------------------------------------- */
size_t gspl_radix_sort_workspace_bytes(int64_t n, int key_bytes /* 4 or 8 */, int begin_bit, int end_bit);
int gspl_radix_sort_pairs_u32(int64_t n, uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1,
                              int begin_bit, int end_bit, int* result_buffer /* host */,
                              void* workspace, size_t workspace_bytes, void* stream);
int gspl_radix_sort_keys_u64(int64_t n, uint64_t* keys0, uint64_t* keys1, int begin_bit, int end_bit,
                             int* result_buffer /* host */, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * 11. Densification statistics of the density controller, one launch.
 *    Replaces the PyTorch lines of `VanillaDensityControllerImpl.update_states` /
 *    `_add_densification_stats` (internal/density_controllers/vanilla_density_controller.py:101-123).
 *    For every Gaussian n with visible[n] != 0 (visible == NULL: radii[n] > 0):
 *        max_radii[n] = max(max_radii[n], radii[n])                     (skipped when max_radii == NULL)
 *        accum[n]    += | (grad[n,0] * scale_x, grad[n,1] * scale_y) |_2
 *        denom[n]    += 1
 *    grad [N, grad_stride] f32 (the first two columns are read: `viewspace_points.grad` or `.absgrad`);
 *    scale: scale_dev (device float[2], e.g. the renderers' `viewspace_points_grad_scale`) when not NULL,
 *    else the two host floats (1, 1 for "no scale"); radii as int32 or float32 (one of them, or neither
 *    when a mask is given and max_radii is NULL); accum, denom, max_radii [N] f32, updated in place.
 * ---------------------------------------------------------------------------------------- */
int gspl_densify_stats(int N, const float* grad, int grad_stride, float scale_x, float scale_y,
                       const float* scale_dev /*nullable*/, const uint8_t* visible /*nullable*/,
                       const int32_t* radii_i32 /*nullable*/, const float* radii_f32 /*nullable*/,
                       float* accum, float* denom, float* max_radii /*nullable*/, void* stream);
/* The same update for the n_views (<= GSPL_STATS_MAX_VIEWS) cameras of ONE Gaussian-sharded step in one launch — the loop over
 * `projection_results_list` of `DistributedVanillaDensityControllerImpl.update_states`
 * (internal/density_controllers/distributed_vanilla_density_controller.py:22-47): grads / visible / radii_i32 are HOST arrays of
 * n_views device pointers ([N, grad_stride] f32, [N] u8 or NULL entries, [N] i32 or NULL entries; `visible` / `radii_i32` themselves may
 * be NULL); the views are applied in array order per Gaussian: the buffers equal those of n_views sequential gspl_densify_stats
 * calls bit for bit.  (ABI 35: round 6.) */
#define GSPL_STATS_MAX_VIEWS 16
int gspl_densify_stats_views(int N, int n_views, const float* const* grads, int grad_stride, float scale_x, float scale_y,
                             const float* scale_dev /*nullable*/, const uint8_t* const* visible /*nullable*/,
                             const int32_t* const* radii_i32 /*nullable*/,
                             float* accum, float* denom, float* max_radii /*nullable*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * 12. Per-splat statistics of a compositing pass (hit-pixel count / rasterize_to_weights).
 *    What LightGaussian pruning (internal/utils/light_gaussian.py:37-50 through
 *    internal/renderers/gsplat_hit_pixel_count_renderer.py:34-44 -> gsplat fork `hit_pixel_count`) and the
 *    Taming-3DGS / GNS scores (internal/density_controllers/taming_3dgs_density_controller.py:429-439 ->
 *    gsplat fork `rasterize_to_weights`) read.  Both kernels are un-vendored; the sums are restated from the
 *    published methods (PARITY UNPINNED).  Traversal and discrete rules of gspl_composite_fwd; for every splat g,
 *    over the pixels p it contributes to (alpha >= 1/255, before the pixel saturates), ADDED to the arrays:
 *        count[g] += 1;  opacity_sum[g] += opacities[g];  alpha_sum[g] += alpha;  visibility_sum[g] += alpha T;
 *        weighted_sum[g] += pixel_weights[p] alpha T;  dist_sum[g] += |p - means2d[g]|
 *    Any output may be NULL; pixel_weights [H,W] f32 is needed for weighted_sum only.  Outputs [N], zeroed by
 *    the caller (several cameras accumulate into the same arrays).
 * ---------------------------------------------------------------------------------------- */
int gspl_composite_scores(int N, int64_t n_isects, int mode,
                          const float* means2d, const float* conics, const float* opacities,
                          int width, int height, int tile_size, int tile_w, int tile_h,
                          const int32_t* offsets, const int32_t* flatten_ids, const float* pixel_weights /*nullable*/,
                          int32_t* count, float* opacity_sum, float* alpha_sum, float* visibility_sum,
                          float* weighted_sum, float* dist_sum, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPL_HIP_H */
