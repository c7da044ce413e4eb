// ba_session.hip -- device-resident bundle adjustment across the reference's incremental loop.  C ABI: include/rcn.h.
//
// SequentialReconstructor::reconstruct (SequentialReconstructor.cpp:1040-1094) registers one view at a time and, after
// every view, runs   checkLandmarkValidity -> BundleAdjuster().adjust(everything so far) -> checkLandmarkValidity ->
// removeOutlierLandmarks   on a problem that differs from the previous one by one camera, its new landmarks and a few
// hundred observations: N - 2 global solves, each of which the reference re-packs from its containers.  A session
// keeps that problem where the solver works:
//   in HBM, across solves   the landmark coordinates (in and out of every solve, of the validity sweep, compacted on
//                           the device when outliers are removed), the observation arrays in landmark-major order,
//                           the observation-pair lists of the Schur build (rebuilt on the device only when the graph
//                           has changed), the solver workspace of the ctx
//   host mirror             the graph itself (tracks in triangulatedFeatures order: additions append, exactly like
//                           push_back in the reference) and the 12 numbers per camera, which the solver's accept /
//                           reject logic reads anyway
// Appending a view costs the bytes of what is new; nothing is re-packed.  The arithmetic of a session solve is the
// arithmetic of rcn_ba_solve on the same problem (same kernels, same order): results are identical bit for bit.
#include "rcn_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>

namespace {

struct TrackObs { int32_t cam, x, y; };

std::atomic<uint32_t> g_session_ids{1};

// out[k] = in[idx[k]]  (3 doubles per landmark): compaction of the landmark array on the device
__global__ void k_gather_points(const double *__restrict__ in, const int32_t *__restrict__ idx, int n, double *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const size_t s = 3 * (size_t)idx[k];
    out[3 * (size_t)k] = in[s]; out[3 * (size_t)k + 1] = in[s + 1]; out[3 * (size_t)k + 2] = in[s + 2];
}

// a device array that keeps its contents when it grows
struct KeepBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t grow(size_t bytes, size_t used, hipStream_t st)
    {
        if (bytes <= cap) return hipSuccess;
        void *q = nullptr;
        const size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess) return e;
        if (p && used) {
            e = hipMemcpyAsync(q, p, used, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { (void)hipFree(q); return e; }
        }
        if (p) (void)hipFree(p);
        p = q; cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct rcn_ba_session {
    rcn_ctx *ctx = nullptr;
    uint32_t id = 0, version = 1;
    std::vector<double> poses, intr;                 // 6 + 6 per camera, rcn_ba_problem layout
    std::vector<std::vector<TrackObs>> tracks;       // per landmark, triangulatedFeatures order
    int64_t n_obs = 0;
    KeepBuf pts;                                     // n_points x 3 doubles, HBM
    DevBuf uv, ocam, opt, xy, pt_off, poses34, intr_dev, inl, keep, cnt, idx, tmp_pts;
    std::vector<int32_t> h_cam, h_pt;                // flattened graph (host), rebuilt with the device arrays
    std::vector<double> h_uv;
    bool obs_dirty = true;                           // device observation arrays do not reflect `tracks`
    std::vector<uint8_t> last_inlier;                // result of the last validity sweep (for remove_outliers)
    bool have_inlier = false;
};

#define SES_HIP(call)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ctx->set_error(std::string(#call) + ": " + hipGetErrorString(e_));           \
            return RCN_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// flatten the tracks (landmark-major, track order) and refresh the device copies
static int sync_observations(rcn_ba_session *s)
{
    rcn_ctx *ctx = s->ctx;
    if (!s->obs_dirty) return RCN_OK;
    const size_t no = (size_t)s->n_obs, np = s->tracks.size();
    s->h_cam.resize(no); s->h_pt.resize(no); s->h_uv.resize(2 * no);
    std::vector<int32_t> xy(2 * no), off(np + 1, 0);
    size_t o = 0;
    for (size_t j = 0; j < np; ++j) {
        for (const TrackObs &t : s->tracks[j]) {
            s->h_cam[o] = t.cam; s->h_pt[o] = (int32_t)j;
            s->h_uv[2 * o] = (double)t.x; s->h_uv[2 * o + 1] = (double)t.y;      // integer pixel coordinates cast to double (BundleAdjuster.cpp:83-84)
            xy[2 * o] = t.x; xy[2 * o + 1] = t.y;
            ++o;
        }
        off[j + 1] = (int32_t)o;
    }
    hipStream_t st = ctx->stream;
    SES_HIP(s->uv.reserve(std::max<size_t>(1, 2 * no) * sizeof(double)));
    SES_HIP(s->ocam.reserve(std::max<size_t>(1, no) * sizeof(int32_t)));
    SES_HIP(s->opt.reserve(std::max<size_t>(1, no) * sizeof(int32_t)));
    SES_HIP(s->xy.reserve(std::max<size_t>(1, 2 * no) * sizeof(int32_t)));
    SES_HIP(s->pt_off.reserve((np + 1) * sizeof(int32_t)));
    if (no) {
        SES_HIP(hipMemcpyAsync(s->uv.p, s->h_uv.data(), 2 * no * sizeof(double), hipMemcpyHostToDevice, st));
        SES_HIP(hipMemcpyAsync(s->ocam.p, s->h_cam.data(), no * sizeof(int32_t), hipMemcpyHostToDevice, st));
        SES_HIP(hipMemcpyAsync(s->opt.p, s->h_pt.data(), no * sizeof(int32_t), hipMemcpyHostToDevice, st));
        SES_HIP(hipMemcpyAsync(s->xy.p, xy.data(), 2 * no * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    SES_HIP(hipMemcpyAsync(s->pt_off.p, off.data(), (np + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SES_HIP(hipStreamSynchronize(st));       // xy / off are locals
    s->obs_dirty = false;
    return RCN_OK;
}

extern "C" {

int rcn_ba_session_create(rcn_ctx *ctx, rcn_ba_session **out)
{
    if (!ctx || !out) return RCN_ERR_ARG;
    rcn_ba_session *s = new rcn_ba_session();
    s->ctx = ctx;
    s->id = g_session_ids.fetch_add(1);
    *out = s;
    return RCN_OK;
}

void rcn_ba_session_destroy(rcn_ba_session *s)
{
    if (!s) return;
    rcn_ctx *ctx = s->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if ((ctx->ba_pair_token >> 32) == s->id) ctx->ba_pair_token = 0;
        s->pts.release();
        DevBuf *bufs[] = {&s->uv, &s->ocam, &s->opt, &s->xy, &s->pt_off, &s->poses34, &s->intr_dev, &s->inl, &s->keep, &s->cnt, &s->idx, &s->tmp_pts};
        for (DevBuf *b : bufs) b->release();
    }
    delete s;
}

int rcn_ba_session_add_camera(rcn_ba_session *s, const double *pose6, const double *intr6, int32_t *index_out)
{
    if (!s || !pose6 || !intr6) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->ctx->mu);
    if (index_out) *index_out = (int32_t)(s->poses.size() / 6);
    s->poses.insert(s->poses.end(), pose6, pose6 + 6);
    s->intr.insert(s->intr.end(), intr6, intr6 + 6);
    ++s->version;                 // the tangent columns / the reduced system change with the camera count
    return RCN_OK;
}

int rcn_ba_session_cameras(rcn_ba_session *s, double *poses_out, double *intr_out, const double *poses_in, const double *intr_in)
{
    if (!s) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->ctx->mu);
    if (poses_in) std::copy(poses_in, poses_in + s->poses.size(), s->poses.begin());
    if (intr_in) std::copy(intr_in, intr_in + s->intr.size(), s->intr.begin());
    if (poses_out) std::copy(s->poses.begin(), s->poses.end(), poses_out);
    if (intr_out) std::copy(s->intr.begin(), s->intr.end(), intr_out);
    return RCN_OK;
}

int rcn_ba_session_add_points(rcn_ba_session *s, int32_t n, const double *xyz_host, int32_t *first_index_out)
{
    if (!s || n < 0 || (n > 0 && !xyz_host)) return RCN_ERR_ARG;
    rcn_ctx *ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t np = s->tracks.size();
    if (first_index_out) *first_index_out = (int32_t)np;
    if (n == 0) return RCN_OK;
    SES_HIP(hipSetDevice(ctx->device));
    SES_HIP(s->pts.grow((np + n) * 24, np * 24, ctx->stream));
    SES_HIP(hipMemcpyAsync(static_cast<char *>(s->pts.p) + np * 24, xyz_host, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    SES_HIP(hipStreamSynchronize(ctx->stream));      // the host rows are borrowed
    s->tracks.resize(np + n);
    s->obs_dirty = true; s->have_inlier = false;
    ++s->version;
    return RCN_OK;
}

int rcn_ba_session_add_observations(rcn_ba_session *s, int32_t n, const int32_t *pt, const int32_t *cam, const int32_t *xy)
{
    if (!s || n < 0 || (n > 0 && (!pt || !cam || !xy))) return RCN_ERR_ARG;
    rcn_ctx *ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int32_t np = (int32_t)s->tracks.size(), nc = (int32_t)(s->poses.size() / 6);
    for (int i = 0; i < n; ++i)
        if (pt[i] < 0 || pt[i] >= np || cam[i] < 0 || cam[i] >= nc) { ctx->set_error("rcn_ba_session_add_observations: landmark or camera index out of range"); return RCN_ERR_ARG; }
    for (int i = 0; i < n; ++i) s->tracks[pt[i]].push_back(TrackObs{cam[i], xy[2 * i], xy[2 * i + 1]});     // triangulatedFeatures.push_back
    s->n_obs += n;
    if (n) { s->obs_dirty = true; s->have_inlier = false; ++s->version; }
    return RCN_OK;
}

int rcn_ba_session_counts(const rcn_ba_session *s, int32_t *n_cams, int32_t *n_points, int64_t *n_obs)
{
    if (!s) return RCN_ERR_ARG;
    if (n_cams) *n_cams = (int32_t)(s->poses.size() / 6);
    if (n_points) *n_points = (int32_t)s->tracks.size();
    if (n_obs) *n_obs = s->n_obs;
    return RCN_OK;
}

int rcn_ba_session_graph(rcn_ba_session *s, int32_t *pt_out, int32_t *cam_out, int32_t *xy_out)
{
    if (!s) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->ctx->mu);
    size_t o = 0;
    for (size_t j = 0; j < s->tracks.size(); ++j)
        for (const TrackObs &t : s->tracks[j]) {
            if (pt_out) pt_out[o] = (int32_t)j;
            if (cam_out) cam_out[o] = t.cam;
            if (xy_out) { xy_out[2 * o] = t.x; xy_out[2 * o + 1] = t.y; }
            ++o;
        }
    return RCN_OK;
}

int rcn_ba_session_solve(rcn_ba_session *s, const rcn_ba_options *options, rcn_ba_summary *summary)
{
    if (!s || !summary) return RCN_ERR_ARG;
    rcn_ctx *ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int32_t nc = (int32_t)(s->poses.size() / 6), np = (int32_t)s->tracks.size();
    if (nc < 1) { ctx->set_error("rcn_ba_session_solve: no camera"); return RCN_ERR_ARG; }
    SES_HIP(hipSetDevice(ctx->device));
    int rc = sync_observations(s);
    if (rc) return rc;
    rcn_ba_options o;
    if (options) o = *options;
    else rcn_ba_default_options(nc, &o);                      // BundleAdjuster.cpp:99-142 for the current camera count
    rcn_ba_problem pb{nc, np, (int32_t)s->n_obs, 0, s->poses.data(), s->intr.data(), nullptr,
                      s->h_uv.data(), s->h_cam.data(), s->h_pt.data()};
    BaResident res{static_cast<double *>(s->pts.p), s->uv.as<double>(), s->ocam.as<int>(), s->opt.as<int>(),
                   ((uint64_t)s->id << 32) | s->version};
    s->have_inlier = false;
    return rcn_int_ba_solve(ctx, &pb, &o, summary, &res);    // poses / intrinsics come back into the host mirror, points stay in HBM
}

int rcn_ba_session_read_points(rcn_ba_session *s, double *xyz_host)
{
    if (!s || !xyz_host) return RCN_ERR_ARG;
    rcn_ctx *ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t np = s->tracks.size();
    if (!np) return RCN_OK;
    SES_HIP(hipSetDevice(ctx->device));
    SES_HIP(hipMemcpyAsync(xyz_host, s->pts.p, np * 24, hipMemcpyDeviceToHost, ctx->stream));
    SES_HIP(hipStreamSynchronize(ctx->stream));
    return RCN_OK;
}

const double *rcn_ba_session_points_device(rcn_ba_session *s) { return s ? static_cast<const double *>(s->pts.p) : nullptr; }

// checkLandmarkValidity on the session's own arrays: the sweep runs on the device; observations it erases are erased
// from the tracks (the reference erases them from triangulatedFeatures in place, SequentialReconstructor.cpp:877-898)
int rcn_ba_session_validity(rcn_ba_session *s, const double *poses34_host, double max_projection_error, double min_triangulation_angle,
                            uint8_t *inlier_out, int32_t *n_inliers_out, int32_t *n_erased_out)
{
    if (!s || !poses34_host) return RCN_ERR_ARG;
    rcn_ctx *ctx = s->ctx;
    const int32_t nc = (int32_t)(s->poses.size() / 6), np = (int32_t)s->tracks.size();
    const size_t no = (size_t)s->n_obs;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        SES_HIP(hipSetDevice(ctx->device));
        int rc = sync_observations(s);
        if (rc) return rc;
        hipStream_t st = ctx->stream;
        SES_HIP(s->poses34.reserve(std::max(nc, 1) * 12 * sizeof(double)));
        SES_HIP(s->intr_dev.reserve(std::max(nc, 1) * 6 * sizeof(double)));
        SES_HIP(s->inl.reserve(std::max(np, 1)));
        SES_HIP(s->keep.reserve(std::max<size_t>(no, 1)));
        SES_HIP(s->cnt.reserve(16));
        SES_HIP(hipMemcpyAsync(s->poses34.p, poses34_host, (size_t)nc * 12 * sizeof(double), hipMemcpyHostToDevice, st));
        SES_HIP(hipMemcpyAsync(s->intr_dev.p, s->intr.data(), (size_t)nc * 6 * sizeof(double), hipMemcpyHostToDevice, st));
        SES_HIP(hipStreamSynchronize(st));
    }
    if (np == 0) { if (n_inliers_out) *n_inliers_out = 0; if (n_erased_out) *n_erased_out = 0; return RCN_OK; }
    rcn_landmark_problem lp{nc, np, (int32_t)no, 0, s->poses34.as<double>(), s->intr_dev.as<double>(), static_cast<const double *>(s->pts.p),
                            s->pt_off.as<int32_t>(), s->ocam.as<int32_t>(), s->xy.as<int32_t>()};
    int rc = rcn_landmark_validity_device(ctx, &lp, max_projection_error, min_triangulation_angle, s->inl.as<uint8_t>(), s->keep.as<uint8_t>(), s->cnt.as<int32_t>());
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::vector<uint8_t> keep(no);
    s->last_inlier.assign(np, 0);
    int32_t n_in = 0;
    SES_HIP(hipMemcpyAsync(s->last_inlier.data(), s->inl.p, np, hipMemcpyDeviceToHost, ctx->stream));
    if (no) SES_HIP(hipMemcpyAsync(keep.data(), s->keep.p, no, hipMemcpyDeviceToHost, ctx->stream));
    SES_HIP(hipMemcpyAsync(&n_in, s->cnt.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SES_HIP(hipStreamSynchronize(ctx->stream));
    int32_t erased = 0;
    size_t o = 0;
    for (int32_t j = 0; j < np; ++j) {
        std::vector<TrackObs> &t = s->tracks[j];
        size_t w = 0;
        for (size_t k = 0; k < t.size(); ++k, ++o) {
            if (keep[o]) t[w++] = t[k];
            else ++erased;
        }
        t.resize(w);
    }
    if (erased) { s->n_obs -= erased; s->obs_dirty = true; ++s->version; }
    s->have_inlier = true;
    if (inlier_out) std::copy(s->last_inlier.begin(), s->last_inlier.end(), inlier_out);
    if (n_inliers_out) *n_inliers_out = n_in;
    if (n_erased_out) *n_erased_out = erased;
    return RCN_OK;
}

// removeOutlierLandmarks (SequentialReconstructor.cpp:956-976) for the flags of the last sweep: the landmark array is
// compacted on the device, the tracks on the host.  new_index_out (may be NULL): n_points entries, -1 = removed.
int rcn_ba_session_remove_outliers(rcn_ba_session *s, int32_t *new_index_out, int32_t *n_removed_out)
{
    if (!s) return RCN_ERR_ARG;
    rcn_ctx *ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!s->have_inlier) { ctx->set_error("rcn_ba_session_remove_outliers: no validity sweep since the last change"); return RCN_ERR_ARG; }
    const int32_t np = (int32_t)s->tracks.size();
    std::vector<int32_t> keep_idx;
    keep_idx.reserve(np);
    for (int32_t j = 0; j < np; ++j) {
        if (s->last_inlier[j]) { if (new_index_out) new_index_out[j] = (int32_t)keep_idx.size(); keep_idx.push_back(j); }
        else if (new_index_out) new_index_out[j] = -1;
    }
    const int32_t left = (int32_t)keep_idx.size();
    if (n_removed_out) *n_removed_out = np - left;
    s->have_inlier = false;
    if (left == np) return RCN_OK;
    SES_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SES_HIP(s->idx.reserve(std::max(left, 1) * sizeof(int32_t)));
    SES_HIP(s->tmp_pts.reserve(std::max(left, 1) * 24));
    if (left) {
        SES_HIP(hipMemcpyAsync(s->idx.p, keep_idx.data(), (size_t)left * sizeof(int32_t), hipMemcpyHostToDevice, st));
        k_gather_points<<<(left + 255) / 256, 256, 0, st>>>(static_cast<const double *>(s->pts.p), s->idx.as<int32_t>(), left, s->tmp_pts.as<double>());
        SES_HIP(hipGetLastError());
        SES_HIP(hipMemcpyAsync(s->pts.p, s->tmp_pts.p, (size_t)left * 24, hipMemcpyDeviceToDevice, st));
        SES_HIP(hipStreamSynchronize(st));
    }
    int64_t no = 0;
    for (int32_t k = 0; k < left; ++k) {
        if (keep_idx[k] != k) s->tracks[k] = std::move(s->tracks[keep_idx[k]]);
        no += (int64_t)s->tracks[k].size();
    }
    s->tracks.resize(left);
    s->n_obs = no;
    s->obs_dirty = true;
    ++s->version;
    return RCN_OK;
}

}  // extern "C"
