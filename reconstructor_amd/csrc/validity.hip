// validity.hip -- landmark validity sweep behind rcn_landmark_validity (include/rcn.h).  gfx950.
//
// The sweep the reference runs before and after every bundle adjustment on the same observation
// graph (SequentialReconstructor::checkLandmarkValidity, SequentialReconstructor.cpp:869-954):
// per landmark, drop observations whose L1 reprojection error exceeds the threshold or that lie
// behind the camera (with the reference's erase-and-skip loop), then require at least one pair of
// surviving observations whose viewing rays subtend more than the minimum angle.
//
//   V1 k_cam_centres   -R't of every camera, once                                   [trivial]
//   V2 k_landmark_sweep one thread per landmark, tracks walked in place (the erase loop needs no
//                      list: position p of the shrinking list is original index p + erased)  [HBM gathers]
//
// Every operation is a separately rounded IEEE double (contraction off), in the order of the
// oracle (oracle/validity_oracle.c), so the decisions agree with it bit for bit up to the last
// ulp of acos.
#include "rcn_internal.h"

namespace {

struct SweepArgs {
    const double *poses, *intr, *pts, *centres;
    const int32_t *pt_off, *obs_cam, *obs_xy;
    int32_t n_points;
    double max_err, min_angle;
    uint8_t *inlier, *keep;
    int32_t *n_inliers;
};

__global__ void k_cam_centres(const double *__restrict__ poses, int n_cams, double *__restrict__ centres)
{
#pragma clang fp contract(off)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cams) return;
    const double *P = poses + 12 * (size_t)c;
    for (int i = 0; i < 3; ++i)
        centres[3 * (size_t)c + i] = ((-P[i]) * P[3] + (-P[4 + i]) * P[7]) + (-P[8 + i]) * P[11];   // SequentialReconstructor.cpp:820
}

__device__ __forceinline__ bool observation_bad(const SweepArgs &a, const double *X, int o)
{
#pragma clang fp contract(off)
    const int c = a.obs_cam[o];
    const double *P = a.poses + 12 * (size_t)c, *K = a.intr + 6 * (size_t)c;
    double l[3];
    for (int i = 0; i < 3; ++i) l[i] = ((P[4 * i] * X[0] + P[4 * i + 1] * X[1]) + P[4 * i + 2] * X[2]) + P[4 * i + 3];   // :842-848
    double x = l[0] / l[2], y = l[1] / l[2];                                                                          // Camera.h:59-76
    const double radius = x * x + y * y;
    const double distortion = K[4] * radius + (K[5] * radius) * radius;
    x += distortion;
    y += distortion;
    const double u = K[0] * x + K[2], v = K[1] * y + K[3];
    const double resid = fabs(u - (double)a.obs_xy[2 * (size_t)o]) + fabs(v - (double)a.obs_xy[2 * (size_t)o + 1]);   // :852-867
    return resid > a.max_err || l[2] < 0;                                                                             // :886-887
}

__global__ __launch_bounds__(128) void k_landmark_sweep(SweepArgs a)
{
#pragma clang fp contract(off)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    bool inl = false;
    if (j < a.n_points) {
        const int o0 = a.pt_off[j], k = a.pt_off[j + 1] - o0;
        const double X[3] = {a.pts[3 * (size_t)j], a.pts[3 * (size_t)j + 1], a.pts[3 * (size_t)j + 2]};
        inl = true;
        int erased = 0, n = k;
        for (int i = 0; i < k; ++i) a.keep[o0 + i] = 1;
        for (int p = 0; p < n; ++p) {                       // :877-898: erase, then ++ skips the element that slid in
            const int o = o0 + p + erased;
            if (observation_bad(a, X, o)) {
                a.keep[o] = 0;
                ++erased; --n;
                if (n < 2) inl = false;
            }
        }
        bool angle_ok = false;                              // :901-947 (the angle is symmetric in the pair)
        for (int p = 0; p < k && !angle_ok; ++p) {
            if (!a.keep[o0 + p]) continue;
            const double *c1 = a.centres + 3 * (size_t)a.obs_cam[o0 + p];
            const double r1[3] = {X[0] - c1[0], X[1] - c1[1], X[2] - c1[2]};
            const double n1 = sqrt((r1[0] * r1[0] + r1[1] * r1[1]) + r1[2] * r1[2]);
            for (int q = p + 1; q < k; ++q) {
                if (!a.keep[o0 + q]) continue;
                const double *c2 = a.centres + 3 * (size_t)a.obs_cam[o0 + q];
                const double r2[3] = {X[0] - c2[0], X[1] - c2[1], X[2] - c2[2]};
                const double n2 = sqrt((r2[0] * r2[0] + r2[1] * r2[1]) + r2[2] * r2[2]);
                const double dot = (r1[0] * r2[0] + r1[1] * r2[1]) + r1[2] * r2[2];
                const double ang = 180.0 * acos(dot / (n1 * n2)) / 3.1415;      // :831-833
                if (ang > a.min_angle) { angle_ok = true; break; }
            }
        }
        if (!angle_ok) inl = false;
        a.inlier[j] = inl ? 1 : 0;
    }
    const unsigned long long m = __ballot(inl);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(a.n_inliers, (int)__popcll(m));
}

int validate(rcn_ctx *ctx, const rcn_landmark_problem *p, const void *inl, const void *keep)
{
    if (!ctx) return RCN_ERR_ARG;
    if (!p || p->n_cams < 0 || p->n_points < 0 || p->n_obs < 0 || (p->n_points > 0 && (!inl || !p->points || !p->pt_off)) ||
        (p->n_obs > 0 && (!keep || !p->obs_cam || !p->obs_xy)) || (p->n_cams > 0 && (!p->poses34 || !p->intrinsics))) {
        ctx->set_error("rcn_landmark_validity: bad argument");
        return RCN_ERR_ARG;
    }
    return RCN_OK;
}

int launch(rcn_ctx *ctx, const rcn_landmark_problem *dp, double max_err, double min_angle, double *centres,
           uint8_t *inl, uint8_t *keep, int32_t *cnt)
{
    hipStream_t st = ctx->stream;
    RCN_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t), st));
    if (dp->n_cams > 0) k_cam_centres<<<(dp->n_cams + 127) / 128, 128, 0, st>>>(dp->poses34, dp->n_cams, centres);
    if (dp->n_points > 0) {
        SweepArgs a;
        a.poses = dp->poses34; a.intr = dp->intrinsics; a.pts = dp->points; a.centres = centres;
        a.pt_off = dp->pt_off; a.obs_cam = dp->obs_cam; a.obs_xy = dp->obs_xy; a.n_points = dp->n_points;
        a.max_err = max_err; a.min_angle = min_angle; a.inlier = inl; a.keep = keep; a.n_inliers = cnt;
        k_landmark_sweep<<<(dp->n_points + 127) / 128, 128, 0, st>>>(a);
    }
    RCN_HIP(hipGetLastError());
    return RCN_OK;
}

}  // namespace

// Structure check on the host: offsets monotone and inside n_obs, camera indices in range.
static int check_graph(rcn_ctx *ctx, const rcn_landmark_problem *p)
{
    if (p->n_points > 0) {
        if (p->pt_off[0] < 0) { ctx->set_error("rcn_landmark_validity: pt_off[0] < 0"); return RCN_ERR_ARG; }
        for (int j = 0; j < p->n_points; ++j)
            if (p->pt_off[j + 1] < p->pt_off[j]) { ctx->set_error("rcn_landmark_validity: pt_off must be non-decreasing"); return RCN_ERR_ARG; }
        if (p->pt_off[p->n_points] > p->n_obs) { ctx->set_error("rcn_landmark_validity: pt_off exceeds n_obs"); return RCN_ERR_ARG; }
    }
    for (int o = 0; o < p->n_obs; ++o)
        if (p->obs_cam[o] < 0 || p->obs_cam[o] >= p->n_cams) { ctx->set_error("rcn_landmark_validity: obs_cam out of range"); return RCN_ERR_ARG; }
    return RCN_OK;
}

extern "C" int rcn_landmark_validity(rcn_ctx *ctx, const rcn_landmark_problem *p, double max_projection_error,
                                     double min_triangulation_angle, uint8_t *out_inlier, uint8_t *out_keep,
                                     int32_t *out_n_inliers)
{
    int rc = validate(ctx, p, out_inlier, out_keep);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    rc = check_graph(ctx, p);
    if (rc) return rc;
    RCN_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t nc = p->n_cams, np = p->n_points, no = p->n_obs;
    const size_t b_pose = 96 * nc, b_intr = 48 * nc, b_pts = 24 * np, b_cen = 24 * nc, b_off = 4 * (np + 1), b_cam = 4 * no,
                 b_xy = 8 * no, b_inl = np, b_keep = no;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t total = al(b_pose) + al(b_intr) + al(b_pts) + al(b_cen) + al(b_off) + al(b_cam) + al(b_xy) + al(b_inl) + al(b_keep) + 256;
    RCN_HIP(ctx->lm_ws.reserve(total));
    char *base = ctx->lm_ws.as<char>();
    size_t off = 0;
    auto take = [&](size_t b) { char *q = base + off; off += al(b); return q; };
    double *d_pose = (double *)take(b_pose), *d_intr = (double *)take(b_intr), *d_pts = (double *)take(b_pts), *d_cen = (double *)take(b_cen);
    int32_t *d_off = (int32_t *)take(b_off), *d_cam = (int32_t *)take(b_cam), *d_xy = (int32_t *)take(b_xy);
    uint8_t *d_inl = (uint8_t *)take(b_inl), *d_keep = (uint8_t *)take(b_keep);
    int32_t *d_cnt = (int32_t *)take(4);
    auto H2D = [&](void *dst, const void *src, size_t bytes) { return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess; };
    RCN_HIP(H2D(d_pose, p->poses34, b_pose)); RCN_HIP(H2D(d_intr, p->intrinsics, b_intr)); RCN_HIP(H2D(d_pts, p->points, b_pts));
    if (np > 0) RCN_HIP(H2D(d_off, p->pt_off, b_off));
    RCN_HIP(H2D(d_cam, p->obs_cam, b_cam)); RCN_HIP(H2D(d_xy, p->obs_xy, b_xy));
    if (no > 0) RCN_HIP(hipMemsetAsync(d_keep, 0, b_keep, st));   // observations outside every track stay 0
    rcn_landmark_problem dp = *p;
    dp.poses34 = d_pose; dp.intrinsics = d_intr; dp.points = d_pts; dp.pt_off = d_off; dp.obs_cam = d_cam; dp.obs_xy = d_xy;
    rc = launch(ctx, &dp, max_projection_error, min_triangulation_angle, d_cen, d_inl, d_keep, d_cnt);
    if (rc) return rc;
    int32_t cnt = 0;
    if (np > 0) RCN_HIP(hipMemcpyAsync(out_inlier, d_inl, b_inl, hipMemcpyDeviceToHost, st));
    if (no > 0) RCN_HIP(hipMemcpyAsync(out_keep, d_keep, b_keep, hipMemcpyDeviceToHost, st));
    RCN_HIP(hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, st));
    RCN_HIP(hipStreamSynchronize(st));
    if (out_n_inliers) *out_n_inliers = cnt;
    return RCN_OK;
}

extern "C" int rcn_landmark_validity_device(rcn_ctx *ctx, const rcn_landmark_problem *p_dev, double max_projection_error,
                                            double min_triangulation_angle, uint8_t *out_inlier_dev, uint8_t *out_keep_dev,
                                            int32_t *out_n_inliers_dev)
{
    int rc = validate(ctx, p_dev, out_inlier_dev, out_keep_dev);
    if (rc) return rc;
    if (!out_n_inliers_dev) { ctx->set_error("rcn_landmark_validity_device: out_n_inliers_dev is required"); return RCN_ERR_ARG; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(ctx->lm_ws.reserve(24 * (size_t)p_dev->n_cams + 256));
    return launch(ctx, p_dev, max_projection_error, min_triangulation_angle, ctx->lm_ws.as<double>(), out_inlier_dev,
                  out_keep_dev, out_n_inliers_dev);
}
