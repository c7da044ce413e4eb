/* oracle/validity_oracle.c -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * CPU restatement of the landmark validity sweep the reference runs before and after every bundle
 * adjustment.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file; the product (reconstructor_amd/) never does.
 *
 * PARITY UNPINNED: the reference holds no fixture for this path and cannot be built here (Eigen and
 * OpenCV headers are absent), so nothing but a reading of its source pins this restatement:
 *   SequentialReconstructor::checkLandmarkValidity      SequentialReconstructor.cpp:869-954
 *   SequentialReconstructor::calcProjectionError        :852-867
 *   SequentialReconstructor::getLandmarkLocalCoords     :839-850
 *   SequentialReconstructor::calcTriangulationAngle     :815-836
 *   PinholeCamera::project                              Camera.h:59-76
 *   thresholds maxProjectionError = 4.0, minTriangulationAngle = 1.0   SequentialReconstructor.h:256-257
 *
 * Quirks restated on purpose:
 *   - the erase loop (:877-898) erases element landFeatId and then increments landFeatId, so the
 *     element that slid into the erased slot is never examined;
 *   - a landmark is marked outlier as soon as an erase leaves fewer than two observations (:893-896),
 *     and also when no ordered pair of the surviving observations subtends more than the minimum
 *     angle (:901-947), which includes every landmark left with fewer than two observations;
 *   - degrees are computed with 3.1415, not pi (:833);
 *   - the distortion term is ADDED to x and y (Camera.h:66-69);
 *   - comparisons with NaN are false: an observation at depth exactly 0 is kept.
 * Arithmetic: IEEE double, every product and sum rounded separately (no contraction; the Makefile
 * passes -ffp-contract=off), sums left to right as Eigen's coefficient-wise 3-vector products do.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define ORC_MAX_TRACK 4096

/* p_c = R p_w + t, pose = 3x4 row-major [R | t]  (:842-848) */
static void local_coords(const double *P, const double *X, double *o)
{
    for (int i = 0; i < 3; ++i) o[i] = ((P[4 * i] * X[0] + P[4 * i + 1] * X[1]) + P[4 * i + 2] * X[2]) + P[4 * i + 3];
}

/* |u - x| + |v - y| with the reference's pinhole model (Camera.h:59-76, :852-867) */
static double projection_error(const double *K, const double *Xl, int fx, int fy)
{
    double x = Xl[0] / Xl[2], y = Xl[1] / Xl[2];
    const double radius = x * x + y * y;
    const double distortion = K[4] * radius + (K[5] * radius) * radius;
    x += distortion;
    y += distortion;
    const double u = K[0] * x + K[2], v = K[1] * y + K[3];
    return fabs(u - (double)fx) + fabs(v - (double)fy);
}

/* angle between the rays from the two camera centres to the landmark, "degrees" (:815-836) */
static double triangulation_angle(const double *P1, const double *P2, const double *X)
{
    double r1[3], r2[3];
    for (int i = 0; i < 3; ++i) {
        const double c1 = ((-P1[i]) * P1[3] + (-P1[4 + i]) * P1[7]) + (-P1[8 + i]) * P1[11];   /* -R' t */
        const double c2 = ((-P2[i]) * P2[3] + (-P2[4 + i]) * P2[7]) + (-P2[8 + i]) * P2[11];
        r1[i] = X[i] - c1;
        r2[i] = X[i] - c2;
    }
    const double dot = (r1[0] * r2[0] + r1[1] * r2[1]) + r1[2] * r2[2];
    const double n1 = sqrt((r1[0] * r1[0] + r1[1] * r1[1]) + r1[2] * r1[2]);
    const double n2 = sqrt((r2[0] * r2[0] + r2[1] * r2[1]) + r2[2] * r2[2]);
    return 180.0 * acos(dot / (n1 * n2)) / 3.1415;
}

/* poses n_cams x 12, intrinsics n_cams x 6 (fx fy cx cy k1 k2), points n_points x 3,
 * observations in track order: obs_cam / obs_xy (integer pixel coordinates), CSR pt_off.
 * out_inlier[n_points] (0/1), out_keep[n_obs] (1 = still in the track after the sweep).
 * Returns the number of inlier landmarks, -1 on a track longer than ORC_MAX_TRACK. */
int orc_landmark_validity(int32_t n_cams, const double *poses, const double *intrinsics,
                          int32_t n_points, const double *points, const int32_t *pt_off,
                          const int32_t *obs_cam, const int32_t *obs_xy,
                          double max_projection_error, double min_triangulation_angle,
                          uint8_t *out_inlier, uint8_t *out_keep)
{
    (void)n_cams;
    int n_in = 0;
    for (int j = 0; j < n_points; ++j) {
        const int o0 = pt_off[j], k = pt_off[j + 1] - o0;
        if (k > ORC_MAX_TRACK) return -1;
        const double *X = points + 3 * (size_t)j;
        int list[ORC_MAX_TRACK], n = k;
        for (int i = 0; i < k; ++i) list[i] = o0 + i;
        int inlier = 1;
        for (int i = 0; i < n; ++i) {              /* :877-898, erase without stepping back */
            const int o = list[i], c = obs_cam[o];
            double Xl[3];
            local_coords(poses + 12 * (size_t)c, X, Xl);
            const double resid = projection_error(intrinsics + 6 * (size_t)c, Xl, obs_xy[2 * o], obs_xy[2 * o + 1]);
            if (resid > max_projection_error || Xl[2] < 0) {
                for (int m = i; m + 1 < n; ++m) list[m] = list[m + 1];
                --n;
                if (n < 2) inlier = 0;
            }
        }
        int angle_ok = 0;
        for (int a = 0; a < n; ++a)                 /* :901-934 */
            for (int b = 0; b < n; ++b) {
                if (a == b) continue;
                const double ang = triangulation_angle(poses + 12 * (size_t)obs_cam[list[a]],
                                                       poses + 12 * (size_t)obs_cam[list[b]], X);
                if (ang > min_triangulation_angle) angle_ok = 1;
            }
        if (!angle_ok) inlier = 0;
        for (int i = 0; i < k; ++i) out_keep[o0 + i] = 0;
        for (int i = 0; i < n; ++i) out_keep[list[i]] = 1;
        out_inlier[j] = (uint8_t)inlier;
        n_in += inlier;
    }
    return n_in;
}

/* single projection error / angle, exposed for spot checks */
double orc_projection_error(const double *pose12, const double *intr6, const double *X, int fx, int fy, double *depth)
{
    double Xl[3];
    local_coords(pose12, X, Xl);
    if (depth) *depth = Xl[2];
    return projection_error(intr6, Xl, fx, fy);
}
double orc_triangulation_angle(const double *pose12_a, const double *pose12_b, const double *X)
{
    return triangulation_angle(pose12_a, pose12_b, X);
}
