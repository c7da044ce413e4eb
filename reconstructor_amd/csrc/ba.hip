// ba.hip -- bundle adjustment on MI355X (gfx950).  (kernels land in the next milestone)
#include "rcn_internal.h"

extern "C" {

void rcn_ba_default_options(int32_t n_cams, rcn_ba_options *o)
{
    if (!o) return;
    o->max_iterations = n_cams < 10 ? 150 : 50;   // BundleAdjuster.cpp:135-142
    o->intrinsics_mode = n_cams < 10 ? 0 : 1;     // :112-121
    o->fix_cam0_pose = 1;                         // :100-101
    o->fix_cam1_translation = 1;                  // :104-105
    o->focal_upper_bound = 1000.0;                // :120-121
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->max_consecutive_invalid_steps = 5;
    o->jacobi_scaling = 1;
}

int rcn_ba_solve(rcn_ctx *ctx, const rcn_ba_problem *, const rcn_ba_options *, rcn_ba_summary *)
{
    if (!ctx) return RCN_ERR_ARG;
    ctx->set_error("rcn_ba_solve: not built yet");
    return RCN_ERR_UNSUPPORTED;
}

}  // extern "C"
