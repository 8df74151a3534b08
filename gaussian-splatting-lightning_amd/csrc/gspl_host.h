// gspl_host.h — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gspl_hip.h"

namespace gspl {

// thread-local last-error text, exposed through gspl_last_error()
void set_error(const char* where, const char* what);

inline int fail_arg(const char* msg) {
    set_error(msg, "invalid argument");
    return GSPL_ERR_INVALID_ARG;
}
inline int fail_ws(const char* msg) {
    set_error(msg, "workspace too small");
    return GSPL_ERR_WORKSPACE;
}
inline int check_launch(const char* where) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(where, hipGetErrorString(e));
        return GSPL_ERR_LAUNCH;
    }
    return GSPL_OK;
}
inline int check_hip(hipError_t e, const char* where) {
    if (e != hipSuccess) {
        set_error(where, hipGetErrorString(e));
        return GSPL_ERR_LAUNCH;
    }
    return GSPL_OK;
}

}  // namespace gspl

// internal launchers shared between translation units (sh.hip -> inria.hip)
namespace gspl {
int sh_fwd_launch(int N, int C, int degree, const float* dirs, const float* origin,
                  const float* dc, int dc_stride, const float* rest, int rest_stride,
                  const uint8_t* mask, const int32_t* mask32, int flags,
                  float* colors, uint8_t* clamped, void* stream, float* jac /* nullable [N,9]: d colour / d unit direction */);
// Adam applied inside the per-Gaussian backward kernels (gspl_rasterize_inria_bwd_adam): moments + hyper-parameters of one parameter
typedef gspl_bwd_adam_tensor ShAdamTargetHost;
struct ShAdamHost { ShAdamTargetHost dc, rest; };
int sh_bwd_launch(int N, int C, int degree, int n_coeffs, const float* dirs, const float* origin,
                  const float* dc, int dc_stride, const float* rest, int rest_stride,
                  const uint8_t* mask, const int32_t* mask32, int flags, const uint8_t* clamped,
                  const float* v_colors, int vc_stride, float* v_dc, float* v_rest, float* v_dirs, void* stream,
                  const float* jac /* nullable: the forward's Jacobian; v_dirs then needs no coefficient read */,
                  const ShAdamHost* adam /* nullable; not NULL: v_dc / v_rest are the PARAMETERS, updated in place, no gradient is written */);
// A table that one kernel clears on behalf of a LATER kernel of the same stream (the tables of a prepared sort): the ~5 us
// radix_zero launch in front of that kernel goes away (profiles/r09_sequence.txt has the two of a frame).  16-byte units.
struct ZeroJob { uint4* p = nullptr; uint32_t n16 = 0; };
// binning.hip -> fused.hip: gspl_bin_count whose scan stores `ticket` into host_counts[2] after the two numbers.
// depth_header_zeroed: the caller's earlier kernel ran bin_depth_header()'s job.  `then_zero`: the last scan kernel runs this job
// (bin_tile_header(): the tables of the emission that follows).
int bin_count_ticket(int N, int mode, const float* means2d, const int32_t* radii, const float* depths, const float* conics, const float* opacities,
                     int tile_size, int tile_w, int tile_h, int32_t* order, int64_t* cum_tiles, int32_t* big_list, void* spans, int64_t* host_counts,
                     void* workspace, size_t workspace_bytes, void* stream, unsigned long long ticket,
                     bool depth_header_zeroed = false, ZeroJob then_zero = ZeroJob());
int bin_depth_header(int N, int n_tiles, void* count_workspace, ZeroJob& job);                       // what bin_count would clear first
int bin_tile_header(int N, int64_t capacity, int n_tiles, void* workspace, ZeroJob& job);            // what gspl_bin_emit would clear first
int bin_emit_impl(int N, int mode, const float* means2d, const int32_t* radii, const float* conics, const float* opacities,
                  const int32_t* order, const int64_t* cum_tiles, const int32_t* big_list, const void* spans,
                  int tile_size, int tile_w, int tile_h, int64_t capacity, void* workspace, size_t workspace_bytes, void* stream,
                  bool tile_header_zeroed);
// the density controller's statistics (gspl_densify_stats) applied by the preprocess backward itself; accum == NULL: not asked for
struct BwdStats { float* accum = nullptr; float* denom = nullptr; float* max_radii = nullptr; };
// inria.hip -> fused.hip: the geometry phase and the preprocess backward with the model's RAW parameters (GSPL_INRIA_RAW_PARAMS)
int inria_geometry_launch(int N, const float* means, const float* scales, const float* quats, const float* cov3d_precomp,
                          const float* viewmatrix, const float* projmatrix, int width, int height, int tile_size,
                          float tanfovx, float tanfovy, float scale_modifier,
                          int32_t* radii, float* means2d, float* depths, float* conics, float* cov3d,
                          const float* raw_opacities /* nullable: activated parameters */, float* opacities_out, hipStream_t s,
                          ZeroJob zero = ZeroJob() /* cleared by the same kernel, for the binning that follows */);
int inria_preprocess_bwd_impl(int N, int degree, int n_coeffs, const float* means, const float* scales, const float* quats,
                              const float* cov3d, const float* shs, const float* shs_rest,
                              const float* viewmatrix, const float* projmatrix, const float* campos,
                              int width, int height, float tanfovx, float tanfovy, float scale_modifier,
                              const int32_t* radii, const uint8_t* clamped,
                              const float* v_means2d, const float* v_conics, const float* v_colors, int grad_stride,
                              float* v_means, float* v_scales, float* v_quats,
                              float* v_cov3d_precomp, float* v_shs, float* v_shs_rest, float* v_colors_precomp,
                              float* v_means2d_ndc, const float* v_opacities_packed, float* v_opacities, const float* sh_jac,
                              const float* opac_act /* nullable: activated parameters */, void* stream,
                              const gspl_bwd_adam_plan* adam = nullptr /* not NULL: v_shs / v_shs_rest / v_scales / v_quats / v_opacities are
                              the PARAMETERS (as means, scales, quats are), updated in place; v_means is scratch [N,3] */,
                              BwdStats stats = BwdStats());
}  // namespace gspl
