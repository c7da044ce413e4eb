/*
 * match_oracle.c -- CPU restatement of the reference's per-pair descriptor matcher.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (reconstructor_amd/) may link,
 * import or call this file.  It is the checker for tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED: the reference (smileyenot983/reconstructor) holds no tests, golden vectors
 * or fixtures for this path, and its arithmetic lives in an un-vendored third-party library
 * (OpenCV >= 4.2, `find_package(OpenCV 4.2 REQUIRED)`, Mapper/CMakeLists.txt:42) that is not
 * present in the build container, so the reference cannot be run here.  The oracle therefore
 * restates the *exact* form of what the reference asks OpenCV for, anchored on the
 * reference's own call sites:
 *
 *   FeatureMatcher.cpp:11-25  featDescToCV      dense row-major K x D fp32 matrix per image
 *   FeatureMatcher.cpp:49     knnMatch(q,t,knn,2) 2 nearest train rows per query row, L2,
 *                                               ascending distance (FLANNBASED approximates
 *                                               this; the oracle is the exact answer)
 *   FeatureMatcher.cpp:55     Lowe ratio        knn[i][0].distance < 0.7f * knn[i][1].distance
 *                                               (ratioThresh is `const float`, FeatureMatcher.h:45)
 *   FeatureMatcher.cpp:58-62  uniqueness        ascending query index, first query to claim a
 *                                               train index keeps it
 *   SequentialReconstructor.cpp:199-279         pair grid: each unordered pair matched once,
 *                                               query = lower image id, train = higher
 *
 * Canonical arithmetic:
 *   d2(q,t)  = fp64 chain, ascending k:  acc = fma(double(q[k]) - double(t[k]), same, acc)
 *   order    = ascending (d2 as fp64, train index)            -- ties go to the lower index
 *   distance = sqrtf((float)d2)                               -- OpenCV returns sqrt of the
 *                                                                squared L2 as float
 *   ratio    = dist0 < ratio * dist1 evaluated in fp32
 *   K2 < 2   = no matches (the reference indexes knn[i][1] unconditionally: UB there)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_JB 32 /* train rows carried together; each keeps its own ascending-k chain */

double orc_l2sq(const float *q, const float *t, int D)
{
    double acc = 0.0;
    for (int k = 0; k < D; ++k) {
        double d = (double)q[k] - (double)t[k];
        acc = fma(d, d, acc);
    }
    return acc;
}

/* transposed copy tT[k][j] so the inner loop over j vectorises without touching the
 * per-(q,t) summation order */
static float *transpose_rows(const float *t, int K, int D, int Kp)
{
    float *tT = (float *)calloc((size_t)D * Kp, sizeof(float));
    for (int j = 0; j < K; ++j)
        for (int k = 0; k < D; ++k)
            tT[(size_t)k * Kp + j] = t[(size_t)j * D + k];
    return tT;
}

/* exact 2-NN of every query row; idx/d2 are K1 x 2; missing neighbours are -1 / +inf */
static void knn2_T(const float *q, int K1, const float *tT, int K2, int Kp, int D,
                   int32_t *idx, double *d2)
{
    for (int i = 0; i < K1; ++i) {
        const float *qi = q + (size_t)i * D;
        double b0 = INFINITY, b1 = INFINITY;
        int32_t i0 = -1, i1 = -1;
        for (int j0 = 0; j0 < K2; j0 += ORC_JB) {
            double acc[ORC_JB];
            for (int l = 0; l < ORC_JB; ++l) acc[l] = 0.0;
            for (int k = 0; k < D; ++k) {
                const double qk = (double)qi[k];
                const float *row = tT + (size_t)k * Kp + j0;
                for (int l = 0; l < ORC_JB; ++l) {
                    double d = qk - (double)row[l];
                    acc[l] = fma(d, d, acc[l]);
                }
            }
            int lim = K2 - j0 < ORC_JB ? K2 - j0 : ORC_JB;
            for (int l = 0; l < lim; ++l) {
                /* ascending j: strict < keeps the lower index on ties */
                double a = acc[l];
                if (a < b0) { b1 = b0; i1 = i0; b0 = a; i0 = j0 + l; }
                else if (a < b1) { b1 = a; i1 = j0 + l; }
            }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        d2[2 * i] = b0;  d2[2 * i + 1] = b1;
    }
}

void orc_knn2(const float *q, int K1, const float *t, int K2, int D, int32_t *idx, double *d2)
{
    int Kp = (K2 + ORC_JB - 1) / ORC_JB * ORC_JB;
    if (Kp == 0) Kp = ORC_JB;
    float *tT = transpose_rows(t, K2, D, Kp);
    knn2_T(q, K1, tT, K2, Kp, D, idx, d2);
    free(tT);
}

/* ratio + uniqueness on a finished 2-NN table (FeatureMatcher.cpp:53-64) */
static int ratio_unique(const int32_t *idx, const double *d2, int K1, int K2, float ratio,
                        int32_t *out)
{
    int count = 0;
    uint8_t *taken = (uint8_t *)calloc(K2 > 0 ? K2 : 1, 1);
    for (int i = 0; i < K1; ++i) {
        out[i] = -1;
        if (K2 < 2) continue;
        float dist0 = sqrtf((float)d2[2 * i]);
        float dist1 = sqrtf((float)d2[2 * i + 1]);
        if (dist0 < ratio * dist1) {
            int32_t tr = idx[2 * i];
            if (!taken[tr]) { taken[tr] = 1; out[i] = tr; ++count; }
        }
    }
    free(taken);
    return count;
}

/* one image pair; out[i] = matched train index of query i or -1; returns the match count */
int orc_match_pair(const float *q, int K1, const float *t, int K2, int D, float ratio,
                   int32_t *out)
{
    if (K1 <= 0) return 0;
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * 2 * K1);
    double *d2 = (double *)malloc(sizeof(double) * 2 * K1);
    orc_knn2(q, K1, t, K2, D, idx, d2);
    int c = ratio_unique(idx, d2, K1, K2, ratio, out);
    free(idx); free(d2);
    return c;
}

/*
 * Pair grid (SequentialReconstructor.cpp:199-279).  desc holds the images back to back,
 * image n = rows [row_off[n], row_off[n+1]) of D floats.  pairs = n_pairs x (query image,
 * train image).  out row stride is out_stride (>= max K1); counts[p] = matches of pair p.
 * The outer pair loop is the parallel one, as in the reference (:202).
 */
void orc_match_grid(const float *desc, const int64_t *row_off, int n_images, int D,
                    const int32_t *pairs, int n_pairs, float ratio,
                    int32_t *out, int64_t out_stride, int32_t *counts, int threads)
{
    float **tTs = (float **)calloc(n_images, sizeof(float *));
    /* transposed copies are built lazily per train image, under a lock */
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int p = 0; p < n_pairs; ++p) {
        int a = pairs[2 * p], b = pairs[2 * p + 1];
        int K1 = (int)(row_off[a + 1] - row_off[a]);
        int K2 = (int)(row_off[b + 1] - row_off[b]);
        int Kp = (K2 + ORC_JB - 1) / ORC_JB * ORC_JB;
        if (Kp == 0) Kp = ORC_JB;
        float *tT;
#ifdef _OPENMP
#pragma omp critical(orc_tt)
#endif
        {
            if (!tTs[b]) tTs[b] = transpose_rows(desc + row_off[b] * D, K2, D, Kp);
            tT = tTs[b];
        }
        int32_t *o = out + (int64_t)p * out_stride;
        if (K1 <= 0) { counts[p] = 0; continue; }
        int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * 2 * K1);
        double *d2 = (double *)malloc(sizeof(double) * 2 * K1);
        knn2_T(desc + row_off[a] * D, K1, tT, K2, Kp, D, idx, d2);
        counts[p] = ratio_unique(idx, d2, K1, K2, ratio, o);
        free(idx); free(d2);
    }
    for (int n = 0; n < n_images; ++n) free(tTs[n]);
    free(tTs);
}
