/* oracle/fmat_oracle.c -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * CPU restatement of the epipolar filter the reference applies to every image pair's matches right
 * after the matcher:
 *   GeometricFilter::estimateFundamental      GeometricFilter.cpp:39-61   -> cv::findFundamentalMat(pts1, pts2, mask)
 *   hook in the pair loop                     SequentialReconstructor.cpp:237-269 (>= 7 matches, points in
 *                                             ascending query-feature order, only the inlier mask is used)
 *   featuresToCvPoints                        utils.cpp:165-177 (integer pixel coordinates as float)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 *
 * PARITY UNPINNED.  The arithmetic lives in OpenCV (>= 4.2, un-vendored, absent here; no fixture in the
 * reference).  What follows restates, FROM MEMORY of OpenCV 4.x modules/calib3d/src/fundam.cpp and
 * ptsetreg.cpp, the published algorithm behind findFundamentalMat's defaults (FM_RANSAC, threshold 3,
 * confidence 0.99, 1000 iterations): RANSAC over 7-point samples when there are >= 15 points, LMedS for
 * 7..14; cv::RNG (multiply-with-carry, A = 4164903690, state 2^64 - 1 at every call); sample drawing with
 * duplicate rejection; the 7-point solver (two-dimensional null space of the 7x9 design matrix, cubic in
 * the mixing weight, F(3,3) normalised to 1); symmetric squared epipolar distance stored as float; the
 * adaptive iteration count.  Deliberate, documented difference: the null space comes from Gauss-Jordan
 * elimination with complete pivoting instead of OpenCV's SVD -- the hypotheses are the same matrices up
 * to rounding, so masks can differ from OpenCV's only at knife edges.
 * Arithmetic: IEEE double, one rounding per operation (-ffp-contract=off), float where OpenCV stores float.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MODEL_POINTS 7
#define MAX_ITERS 1000
#define MAX_ATTEMPTS 10000

static unsigned rng_next(uint64_t *s)
{
    *s = (uint64_t)(unsigned)*s * 4164903690U + (unsigned)(*s >> 32);
    return (unsigned)*s;
}

/* haveCollinearPoints: is the LAST point on a line through two earlier ones (or too close to one)? */
static int last_point_collinear(const float *m, int count)
{
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = m[2 * j] - m[2 * i], dy1 = m[2 * j + 1] - m[2 * i + 1];
        for (int k = 0; k < j; ++k) {
            const double dx2 = m[2 * k] - m[2 * i], dy2 = m[2 * k + 1] - m[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
        }
    }
    return 0;
}

/* RANSACPointSetRegistrator::getSubset + FMEstimatorCallback::checkSubset: seven distinct indices
 * drawn uniformly (a duplicate is redrawn); the whole sample is redrawn while its last point is
 * collinear with two earlier ones in either image, at most MAX_ATTEMPTS times. */
static int draw_subset(uint64_t *rng, int count, const float *m1, const float *m2, float *s1, float *s2)
{
    for (int attempt = 0; attempt < MAX_ATTEMPTS; ++attempt) {
        int idx[MODEL_POINTS];
        for (int i = 0; i < MODEL_POINTS;) {
            const int v = (int)(rng_next(rng) % (unsigned)count);
            int j = 0;
            for (; j < i; ++j) if (idx[j] == v) break;
            if (j < i) continue;
            idx[i++] = v;
        }
        for (int i = 0; i < MODEL_POINTS; ++i) {
            s1[2 * i] = m1[2 * idx[i]]; s1[2 * i + 1] = m1[2 * idx[i] + 1];
            s2[2 * i] = m2[2 * idx[i]]; s2[2 * i + 1] = m2[2 * idx[i] + 1];
        }
        if (!last_point_collinear(s1, MODEL_POINTS) && !last_point_collinear(s2, MODEL_POINTS)) return 1;
    }
    return 0;
}

/* RANSACUpdateNumIters */
static int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* real roots of c[0] x^3 + c[1] x^2 + c[2] x + c[3] (cv::solveCubic) */
static int solve_cubic(const double *c, double *r)
{
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    double x0 = 0, x1 = 0, x2 = 0;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x0 = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0;
        a1 *= a0; a2 *= a0; a3 *= a0;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        const double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            const double theta = acos(R / sqrt(Qcubed));
            const double sqrtQ = sqrt(Q);
            const double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * 3.14159265358979323846 / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * 3.14159265358979323846 / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) { x0 = -2 * cbrt(R) - a1 / 3; x1 = cbrt(R) - a1 / 3; }
            else { x0 = 2 * cbrt(-R) - a1 / 3; x1 = -cbrt(-R) - a1 / 3; }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            double e;
            d = sqrt(-d);
            e = cbrt(d + fabs(R));
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    r[0] = x0; r[1] = x1; r[2] = x2;
    return n;
}

/* Two vectors spanning the null space of the 7x9 matrix A (row-major, destroyed): Gauss-Jordan with
 * complete pivoting and no row exchanges -- step k takes the largest |entry| among the rows and
 * columns not used yet (first one in row-major order on ties), scales that row and clears the pivot
 * column in every other row.  The two columns never chosen are the free variables: f1 <- first free
 * variable = 1, f2 <- second (any basis spans the same pencil).  Returns 0 when the rank is < 7. */
static int null_space_7x9(double *A, double *f1, double *f2)
{
    int piv_row[7], piv_col[7], row_used[7] = {0}, col_used[9] = {0};
    for (int k = 0; k < 7; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
        for (int r = 0; r < 7; ++r) {
            if (row_used[r]) continue;
            for (int c = 0; c < 9; ++c) {
                if (col_used[c]) continue;
                const double v = fabs(A[9 * r + c]);
                if (v > best) { best = v; pr = r; pc = c; }
            }
        }
        if (pr < 0) return 0;
        row_used[pr] = 1; col_used[pc] = 1; piv_row[k] = pr; piv_col[k] = pc;
        const double inv = 1. / A[9 * pr + pc];
        for (int c = 0; c < 9; ++c) A[9 * pr + c] *= inv;
        for (int r = 0; r < 7; ++r) {
            if (r == pr) continue;
            const double m = A[9 * r + pc];
            if (m == 0) continue;
            for (int c = 0; c < 9; ++c) A[9 * r + c] -= m * A[9 * pr + c];
        }
    }
    int fc[2], nf = 0;
    for (int c = 0; c < 9; ++c) if (!col_used[c]) fc[nf++] = c;
    double *out[2] = {f1, f2};
    for (int b = 0; b < 2; ++b) {
        for (int c = 0; c < 9; ++c) out[b][c] = 0;
        out[b][fc[b]] = 1;
        for (int k = 0; k < 7; ++k) out[b][piv_col[k]] = -A[9 * piv_row[k] + fc[b]];
    }
    return 1;
}

/* run7Point: up to three 3x3 matrices (row-major) from seven correspondences */
static int run_7point(const float *m1, const float *m2, double *fmatrix)
{
    double a[7 * 9], f1[9], f2[9], c[4], r[3];
    for (int i = 0; i < 7; ++i) {
        const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
        double *row = a + 9 * i;
        row[0] = x1 * x0; row[1] = x1 * y0; row[2] = x1;
        row[3] = y1 * x0; row[4] = y1 * y0; row[5] = y1;
        row[6] = x0; row[7] = y0; row[8] = 1;
    }
    if (!null_space_7x9(a, f1, f2)) return 0;
    for (int i = 0; i < 9; ++i) f1[i] -= f2[i];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 -
           f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) + f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7]; t1 = f1[3] * f1[8] - f1[5] * f1[6]; t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 -
           f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) + f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    const int n = solve_cubic(c, r);
    if (n < 1 || n > 3) return n < 0 ? 0 : n > 3 ? 0 : n;
    for (int k = 0; k < n; ++k, fmatrix += 9) {
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; fmatrix[8] = 1.; }
        else fmatrix[8] = 0.;
        for (int i = 0; i < 8; ++i) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

/* FMEstimatorCallback::computeError */
static float epi_error(const double *F, const float *p1, const float *p2)
{
    double a, b, c, d1, d2, s1, s2;
    a = F[0] * p1[0] + F[1] * p1[1] + F[2];
    b = F[3] * p1[0] + F[4] * p1[1] + F[5];
    c = F[6] * p1[0] + F[7] * p1[1] + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = p2[0] * a + p2[1] * b + c;
    a = F[0] * p2[0] + F[3] * p2[1] + F[6];
    b = F[1] * p2[0] + F[4] * p2[1] + F[7];
    c = F[2] * p2[0] + F[5] * p2[1] + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = p1[0] * a + p1[1] * b + c;
    return (float)fmax(d1 * d1 * s1, d2 * d2 * s2);
}

static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* One pair.  xy1 / xy2: n x 2 integer pixel coordinates (matched features in ascending query order).
 * mask[n]: 1 = inlier.  Returns the inlier count; -1 when no model was found (the reference then sees
 * an empty matrix and drops the pair's matches, SequentialReconstructor.cpp:252-255); n < 7 is the
 * caller's business (the reference does not filter such pairs) and returns -2 with an all-ones mask. */
int orc_fmat_filter(const int32_t *xy1, const int32_t *xy2, int32_t n, uint8_t *mask, int32_t *iterations)
{
    const double confidence = 0.99, threshold = 3.0;
    if (iterations) *iterations = 0;
    if (n < MODEL_POINTS) { for (int i = 0; i < n; ++i) mask[i] = 1; return -2; }
    float *m1 = (float *)malloc(sizeof(float) * 4 * (size_t)n), *m2 = m1 + 2 * (size_t)n;
    uint8_t *cur = (uint8_t *)malloc((size_t)n);
    for (int i = 0; i < 2 * n; ++i) { m1[i] = (float)xy1[i]; m2[i] = (float)xy2[i]; }
    uint64_t rng = (uint64_t)-1;
    double best_model[9] = {0};
    int result = -1, it_done = 0;
    if (n == MODEL_POINTS) {                           /* both registrators: the sample is the data set */
        double models[27];
        if (run_7point(m1, m2, models) > 0) { memset(mask, 1, (size_t)n); result = n; }
        it_done = 1;
    } else if (n >= 15) {                              /* RANSACPointSetRegistrator::run */
        const double thr2 = threshold * threshold;
        int niters = MAX_ITERS, max_good = 0;
        for (int iter = 0; iter < niters; ++iter) {
            float s1[14], s2[14];
            double models[27];
            if (!draw_subset(&rng, n, m1, m2, s1, s2)) break;
            const int nm = run_7point(s1, s2, models);
            it_done = iter + 1;
            for (int k = 0; k < nm; ++k) {
                int good = 0;
                for (int i = 0; i < n; ++i) { cur[i] = epi_error(models + 9 * k, m1 + 2 * i, m2 + 2 * i) <= thr2; good += cur[i]; }
                if (good > (max_good > MODEL_POINTS - 1 ? max_good : MODEL_POINTS - 1)) {
                    memcpy(mask, cur, (size_t)n);
                    memcpy(best_model, models + 9 * k, sizeof(best_model));
                    max_good = good;
                    niters = update_num_iters(confidence, (double)(n - good) / n, MODEL_POINTS, niters);
                }
            }
        }
        if (max_good > 0) result = max_good;
    } else {                                           /* LMeDSPointSetRegistrator::run, 8 <= n <= 14 */
        const double outlier_ratio = 0.45;
        int niters = update_num_iters(confidence, outlier_ratio, MODEL_POINTS, MAX_ITERS);
        if (niters < 3) niters = 3;
        double min_median = DBL_MAX;
        for (int iter = 0; iter < niters; ++iter) {
            float s1[14], s2[14], sorted[14];
            double models[27];
            if (!draw_subset(&rng, n, m1, m2, s1, s2)) break;
            const int nm = run_7point(s1, s2, models);
            it_done = iter + 1;
            for (int k = 0; k < nm; ++k) {
                for (int i = 0; i < n; ++i) sorted[i] = epi_error(models + 9 * k, m1 + 2 * i, m2 + 2 * i);
                qsort(sorted, (size_t)n, sizeof(float), cmp_float);
                const double median = n % 2 != 0 ? sorted[n / 2] : (sorted[n / 2 - 1] + sorted[n / 2]) * 0.5;
                if (median < min_median) { min_median = median; memcpy(best_model, models + 9 * k, sizeof(best_model)); }
            }
        }
        if (min_median < DBL_MAX) {
            double sigma = 2.5 * 1.4826 * (1 + 5. / (n - MODEL_POINTS)) * sqrt(min_median);
            sigma = fmax(sigma, 0.001);
            const double thr2 = sigma * sigma;
            int good = 0;
            for (int i = 0; i < n; ++i) { mask[i] = epi_error(best_model, m1 + 2 * i, m2 + 2 * i) <= thr2; good += mask[i]; }
            result = good >= MODEL_POINTS ? good : -1;
        }
    }
    if (result < 0) memset(mask, 0, (size_t)n);
    if (iterations) *iterations = it_done;
    free(m1); free(cur);
    return result;
}

/* CSR batch: pair p owns points off[p] .. off[p+1].  counts[p] as orc_fmat_filter returns. */
void orc_fmat_filter_grid(int32_t n_pairs, const int32_t *off, const int32_t *xy1, const int32_t *xy2,
                          uint8_t *mask, int32_t *counts, int32_t *iterations, int threads)
{
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
    for (int p = 0; p < n_pairs; ++p)
        counts[p] = orc_fmat_filter(xy1 + 2 * (size_t)off[p], xy2 + 2 * (size_t)off[p], off[p + 1] - off[p],
                                    mask + off[p], iterations ? iterations + p : NULL);
}

/* exposed pieces for spot checks */
int orc_fmat_7point(const float *m1, const float *m2, double *fmatrix) { return run_7point(m1, m2, fmatrix); }
float orc_fmat_error(const double *F, const float *p1, const float *p2) { return epi_error(F, p1, p2); }
unsigned orc_rng_next(uint64_t *state) { return rng_next(state); }
int orc_ransac_num_iters(double p, double ep, int model_points, int max_iters) { return update_num_iters(p, ep, model_points, max_iters); }
