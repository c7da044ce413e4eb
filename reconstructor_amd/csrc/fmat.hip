// fmat.hip -- epipolar (fundamental-matrix) inlier filter behind rcn_fmat_filter* (include/rcn.h).  gfx950.
//
// What GeometricFilter::estimateFundamental asks cv::findFundamentalMat for (GeometricFilter.cpp:39-61,
// hook SequentialReconstructor.cpp:237-269): the inlier mask of the matches of one image pair under a
// fundamental matrix found by RANSAC over 7-point samples (>= 15 points) or LMedS (8..14 points), with
// OpenCV's defaults (threshold 3 px, confidence 0.99, at most 1000 iterations) and OpenCV's sampling
// sequence (cv::RNG, same seed for every call).  The algorithm is restated in oracle/fmat_oracle.c (see
// its header for what is taken from memory of OpenCV 4.x and the one deliberate difference); this file
// follows the oracle operation by operation with contraction off, so that the masks agree bit for bit.
//
//   F1 k_fmat_filter  one wave per pair, rounds of 32 hypotheses:
//        lane 0        draws the 32 index sets (the RNG stream is sequential); a collinear sample, which
//                      makes the reference redraw, is detected by 32 lanes at once and sends the round
//                      down a slow path that replays the reference's loop literally
//        32 lanes      one 7-point solve each, 7x9 design matrix in registers (Gauss-Jordan with
//                      complete pivoting by selects), cubic, <= 3 matrices
//        64 lanes      score <= 96 matrices against all n points (points resident in LDS as float2)
//        lane 0        replays the sequential accept / shrink-the-iteration-count logic over the round
//      and a final pass that writes the mask of the winning matrix.                      [fp64 VALU]
#include "rcn_internal.h"

#include <cfloat>

namespace {

#define FM_B 32            // hypotheses per round
#define FM_NLDS 1024       // points kept in LDS; larger pairs read them from global memory
#define FM_MAX_ITERS 1000
#define FM_MAX_ATTEMPTS 10000

struct FmatArgs {
    const int32_t *off, *xy1, *xy2;
    int32_t n_pairs;
    uint8_t *mask;
    int32_t *counts, *iters;
    double *F;   // optional: the winning matrix of every pair, 9 doubles (zeros when there is none)
};

__device__ __forceinline__ unsigned rng_next(unsigned long long &s)
{
    s = (unsigned long long)(unsigned)s * 4164903690U + (unsigned)(s >> 32);
    return (unsigned)s;
}

__device__ __forceinline__ int update_num_iters(double p, double ep, int model_points, int max_iters)
{
#pragma clang fp contract(off)
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

__device__ int solve_cubic(const double *c, double *r)
{
#pragma clang fp contract(off)
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

// seven correspondences (s1, s2: 7 x 2 floats, in registers) -> up to three matrices in F (27 doubles,
// LDS).  The 7x9 design matrix lives in registers: every loop is fully unrolled and the pivot row /
// column -- which are data dependent -- are picked with selects, never with indexed addressing.
__device__ int run_7point(const float *s1, const float *s2, double *F)
{
#pragma clang fp contract(off)
    double A[7][9];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double x0 = s1[2 * i], y0 = s1[2 * i + 1], x1 = s2[2 * i], y1 = s2[2 * i + 1];
        A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1;
        A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1;
        A[i][6] = x0; A[i][7] = y0; A[i][8] = 1;
    }
    // null space: Gauss-Jordan, complete pivoting, no row exchanges (oracle: null_space_7x9)
    unsigned row_used = 0, col_used = 0;
    int prk[7], pck[7];
    bool singular = false;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const double v = fabs(A[r][c]);
                const bool ok = !(row_used >> r & 1) && !(col_used >> c & 1) && v > best;
                best = ok ? v : best; pr = ok ? r : pr; pc = ok ? c : pc;
            }
        singular |= pr < 0;
        pr = pr < 0 ? 0 : pr; pc = pc < 0 ? 0 : pc;
        row_used |= 1u << pr; col_used |= 1u << pc;
        prk[k] = pr; pck[k] = pc;
        double prow[9], pval = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            double v = 0;
#pragma unroll
            for (int r = 0; r < 7; ++r) v = r == pr ? A[r][c] : v;
            prow[c] = v;
            pval = c == pc ? v : pval;
        }
        const double inv = 1. / pval;
#pragma unroll
        for (int c = 0; c < 9; ++c) prow[c] *= inv;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            double m = 0;
#pragma unroll
            for (int c = 0; c < 9; ++c) m = c == pc ? A[r][c] : m;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const double upd = A[r][c] - m * prow[c];
                A[r][c] = r == pr ? prow[c] : (m == 0 ? A[r][c] : upd);
            }
        }
    }
    if (singular) return 0;
    // the two free columns, ascending
    int fc0 = -1, fc1 = -1;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const bool fr = !(col_used >> c & 1);
        fc1 = fr && fc0 >= 0 && fc1 < 0 ? c : fc1;
        fc0 = fr && fc0 < 0 ? c : fc0;
    }
    double g1[9], g2[9];   // basis vectors: free variable = 1, pivot variables = -A[piv_row][free col]
#pragma unroll
    for (int c = 0; c < 9; ++c) { g1[c] = c == fc0 ? 1.0 : 0.0; g2[c] = c == fc1 ? 1.0 : 0.0; }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        double v0 = 0, v1 = 0;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                v0 = r == prk[k] && c == fc0 ? A[r][c] : v0;
                v1 = r == prk[k] && c == fc1 ? A[r][c] : v1;
            }
#pragma unroll
        for (int c = 0; c < 9; ++c) { g1[c] = c == pck[k] ? -v0 : g1[c]; g2[c] = c == pck[k] ? -v1 : g2[c]; }
    }
    double f1[9], f2[9], c[4], r[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) { f2[i] = g2[i]; f1[i] = g1[i] - g2[i]; }
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
    if (n < 1 || n > 3) return 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k >= n) break;
        double *fm = F + 9 * k;
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        double f8 = 0.;
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; f8 = 1.; }
        fm[8] = f8;
#pragma unroll
        for (int i = 0; i < 8; ++i) fm[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

__device__ __forceinline__ float epi_error(const double *F, float p1x, float p1y, float p2x, float p2y)
{
#pragma clang fp contract(off)
    double a, b, c, d1, d2, s1, s2;
    a = F[0] * p1x + F[1] * p1y + F[2];
    b = F[3] * p1x + F[4] * p1y + F[5];
    c = F[6] * p1x + F[7] * p1y + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = p2x * a + p2y * b + c;
    a = F[0] * p2x + F[3] * p2y + F[6];
    b = F[1] * p2x + F[4] * p2y + F[7];
    c = F[2] * p2x + F[5] * p2y + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = p1x * a + p1y * b + c;
    return (float)fmax(d1 * d1 * s1, d2 * d2 * s2);
}

// epi_error(...) <= 9.0 without the two divisions whenever the answer is clear: with t = d^2 and
// q = a^2 + b^2 the stored error is fl32(fl(fl(d d) fl(1/q))), within 1e-7 relative of t/q, so
// t <= 8.9999 q on both sides is an inlier and t >= 9.0001 q on either side an outlier; anything
// in between (or a degenerate q) takes the exact path.
__device__ __forceinline__ bool epi_inlier9(const double *F, float p1x, float p1y, float p2x, float p2y)
{
#pragma clang fp contract(off)
    double a, b, c;
    a = F[0] * p1x + F[1] * p1y + F[2];
    b = F[3] * p1x + F[4] * p1y + F[5];
    c = F[6] * p1x + F[7] * p1y + F[8];
    const double q2 = a * a + b * b;
    const double d2 = p2x * a + p2y * b + c, t2 = d2 * d2;
    a = F[0] * p2x + F[3] * p2y + F[6];
    b = F[1] * p2x + F[4] * p2y + F[7];
    c = F[2] * p2x + F[5] * p2y + F[8];
    const double q1 = a * a + b * b;
    const double d1 = p1x * a + p1y * b + c, t1 = d1 * d1;
    const bool sane = q1 > 0 && q2 > 0 && q1 < DBL_MAX && q2 < DBL_MAX;
    if (sane && t1 <= 8.9999 * q1 && t2 <= 8.9999 * q2) return true;
    if (sane && (t1 >= 9.0001 * q1 || t2 >= 9.0001 * q2)) return false;
    const double s1 = 1. / q1, s2 = 1. / q2;
    return (float)fmax(d1 * d1 * s1, d2 * d2 * s2) <= 9.0;
}

__device__ __forceinline__ bool last_point_collinear(const float *m, int count)
{
#pragma clang fp contract(off)
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = m[2 * j] - m[2 * i], dy1 = m[2 * j + 1] - m[2 * i + 1];
        for (int k = 0; k < j; ++k) {
            const double dx2 = m[2 * k] - m[2 * i], dy2 = m[2 * k + 1] - m[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}

#ifdef RCN_FM_PROF   // diagnostic build only (tools/fmat_prof.hip): cycles per phase, summed over pairs
__device__ unsigned long long g_fm_prof[8];
#define FM_T(i) do { const unsigned long long now_ = clock64(); if (lane == 0) atomicAdd(&g_fm_prof[i], now_ - tprev_); tprev_ = clock64(); } while (0)
#define FM_T0() unsigned long long tprev_ = clock64()
#else
#define FM_T(i)
#define FM_T0()
#endif

struct Pts {   // the pair's points: LDS copy when it fits, else the global arrays
    const float2 *l1, *l2;
    const int32_t *g1, *g2;
    __device__ __forceinline__ void get(int i, float &ax, float &ay, float &bx, float &by) const
    {
        if (l1) { const float2 a = l1[i], b = l2[i]; ax = a.x; ay = a.y; bx = b.x; by = b.y; }
        else { ax = (float)g1[2 * i]; ay = (float)g1[2 * i + 1]; bx = (float)g2[2 * i]; by = (float)g2[2 * i + 1]; }
    }
};

// One workgroup of four waves per pair: wave 0 samples, solves and runs the accept logic, all four
// score the round's matrices (matrix m goes to wave m % 4), and five or six pairs share a CU so
// that the serial parts of one overlap with the parallel parts of the others.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_fmat_filter(FmatArgs a)
{
#pragma clang fp contract(off)
    __shared__ float2 P1[FM_NLDS], P2[FM_NLDS];
    __shared__ double sF[FM_B][27], bestF[9], sMed[FM_B * 3];
    __shared__ int sIdx[FM_B][7], sNm[FM_B], sGood[FM_B * 3];
    __shared__ int sCtl[8];   // 0: hypotheses drawn this round, 1: stop, 2: niters, 3: max_good, 4: iterations done, 5: draw failed, 6/7: verdicts
    __shared__ unsigned long long sRng;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int pair = blockIdx.x; pair < a.n_pairs; pair += gridDim.x) {
        const int o0 = a.off[pair], n = a.off[pair + 1] - o0;
        uint8_t *mask = a.mask + o0;
        __syncthreads();
        if (n < 7) {   // not filtered by the reference (SequentialReconstructor.cpp:237)
            for (int i = t; i < n; i += 256) mask[i] = 1;
            if (t == 0) { a.counts[pair] = -2; a.iters[pair] = 0; }
            if (a.F && t < 9) a.F[9 * (size_t)pair + t] = 0.0;
            continue;
        }
        Pts pts;
        pts.g1 = a.xy1 + 2 * (size_t)o0; pts.g2 = a.xy2 + 2 * (size_t)o0;
        pts.l1 = n <= FM_NLDS ? P1 : nullptr; pts.l2 = n <= FM_NLDS ? P2 : nullptr;
        if (n <= FM_NLDS)
            for (int i = t; i < n; i += 256) {
                P1[i] = make_float2((float)pts.g1[2 * i], (float)pts.g1[2 * i + 1]);
                P2[i] = make_float2((float)pts.g2[2 * i], (float)pts.g2[2 * i + 1]);
            }
        const bool ransac = n >= 15;
        FM_T0();
        if (t == 0) {
            sRng = ~0ull;
            sCtl[1] = 0; sCtl[3] = 0; sCtl[4] = 0; sCtl[5] = 0;
            int ni = FM_MAX_ITERS;
            if (!ransac) { ni = update_num_iters(0.99, 0.45, 7, FM_MAX_ITERS); if (ni < 3) ni = 3; }
            if (n == 7) ni = 1;
            sCtl[2] = ni;
        }
        __syncthreads();
        double min_median = DBL_MAX;   // lane 0 only
        bool have_best = false;        // lane 0 only
        for (int base = 0;; base += FM_B) {
            // ---- draw: lane 0 produces the index sets (the RNG stream is sequential); the rare
            // collinear sample, which makes the reference redraw, is detected by all lanes below
            // Every lane runs the draw on wave-uniform values (readfirstlane), so it executes on the
            // scalar unit; x % n is a multiply-high by floor((2^32-1)/n) plus at most two corrections.
            const unsigned long long rng0 = sRng;
            if (w == 0) {
                unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)rng0), hi = __builtin_amdgcn_readfirstlane((unsigned)(rng0 >> 32));
                const unsigned un = (unsigned)__builtin_amdgcn_readfirstlane(n);
                const unsigned M = 0xFFFFFFFFu / un;
                const int niters = __builtin_amdgcn_readfirstlane(sCtl[2]);
                int drawn = 0;
                for (; drawn < FM_B && base + drawn < niters; ++drawn) {
                    int idx[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) idx[j] = j;            // n == 7: the sample is the data set
                    if (un != 7)
                        for (int i = 0; i < 7;) {
                            const unsigned long long st = (unsigned long long)lo * 4164903690U + hi;
                            lo = (unsigned)st; hi = (unsigned)(st >> 32);
                            unsigned v = lo - __umulhi(lo, M) * un;
                            while (v >= un) v -= un;
                            bool dup = false;
#pragma unroll
                            for (int j = 0; j < 7; ++j) dup |= j < i && idx[j] == (int)v;
                            if (dup) continue;
#pragma unroll
                            for (int j = 0; j < 7; ++j) idx[j] = j == i ? (int)v : idx[j];
                            ++i;
                        }
                    if (lane < 7) {
                        int mine = idx[0];
#pragma unroll
                        for (int j = 1; j < 7; ++j) mine = lane == j ? idx[j] : mine;
                        sIdx[drawn][lane] = mine;
                    }
                }
                if (lane == 0) { sRng = (unsigned long long)hi << 32 | lo; sCtl[0] = drawn; }
            }
            __syncthreads();
            FM_T(0);
            int drawn = sCtl[0];
            float s1[14], s2[14];
            bool coll = false;
            if (t < drawn) {
#pragma unroll
                for (int i = 0; i < 7; ++i) pts.get(sIdx[lane][i], s1[2 * i], s1[2 * i + 1], s2[2 * i], s2[2 * i + 1]);
                coll = n != 7 && (last_point_collinear(s1, 7) || last_point_collinear(s2, 7));
            }
            if (t < 64 && __ballot(coll)) sCtl[5] = 2;      // (wave 0) flag the slow path for everybody
            __syncthreads();
            const bool slow = sCtl[5] == 2;
            __syncthreads();                                // everyone has read the flag before lane 0 clears it
            if (slow) {
                // slow path, exactly the reference's loop: redo the round's draws one by one with the
                // collinearity test inside
                if (t == 0) {
                    sCtl[5] = 0;
                    unsigned long long rng = rng0;
                    int d2 = 0;
                    const int niters = sCtl[2];
                    for (; d2 < FM_B && base + d2 < niters; ++d2) {
                        bool ok = false;
                        for (int attempt = 0; attempt < FM_MAX_ATTEMPTS && !ok; ++attempt) {
                            int idx[7];
                            float q1[14], q2[14];
                            for (int i = 0; i < 7;) {
                                const int v = (int)(rng_next(rng) % (unsigned)n);
                                bool dup = false;
#pragma unroll
                                for (int j = 0; j < 7; ++j) dup |= j < i && idx[j] == v;
                                if (dup) continue;
#pragma unroll
                                for (int j = 0; j < 7; ++j) {
                                    if (j == i) { idx[j] = v; pts.get(v, q1[2 * j], q1[2 * j + 1], q2[2 * j], q2[2 * j + 1]); }
                                }
                                ++i;
                            }
                            ok = !last_point_collinear(q1, 7) && !last_point_collinear(q2, 7);
                            if (ok) {
#pragma unroll
                                for (int i = 0; i < 7; ++i) sIdx[d2][i] = idx[i];
                            }
                        }
                        if (!ok) { sCtl[5] = 1; break; }
                    }
                    sRng = rng;
                    sCtl[0] = d2;
                }
                __syncthreads();
                drawn = sCtl[0];
                if (t < drawn) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) pts.get(sIdx[lane][i], s1[2 * i], s1[2 * i + 1], s2[2 * i], s2[2 * i + 1]);
                }
            }
            FM_T(1);
            // ---- solve: one hypothesis per lane, in registers
            if (t < drawn) sNm[t] = run_7point(s1, s2, sF[t]);
            __syncthreads();
            FM_T(2);
            // ---- score
            if (ransac) {
                for (int m = w; m < 3 * drawn; m += 4) {        // each wave walks its share of the models, lanes walk the points
                    const int h = m / 3, k = m - 3 * h;
                    if (k >= sNm[h]) continue;
                    double F[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) F[i] = sF[h][9 * k + i];
                    // A matrix is accepted only if it beats the best count so far (and 6).  The best at
                    // the start of the round is a lower bound of the best at this matrix's turn, so once
                    // even all remaining points could not lift the count above it the matrix is dropped
                    // (its stored count stays <= the bound: never accepted, exactly as if fully counted).
                    const int bound = sCtl[3] > 6 ? sCtl[3] : 6;
                    int good = 0;
                    for (int i0 = 0; i0 < n; i0 += 64) {
                        const int i = i0 + lane;
                        bool in = false;
                        if (i < n) { float ax, ay, bx, by; pts.get(i, ax, ay, bx, by); in = epi_inlier9(F, ax, ay, bx, by); }
                        good += (int)__popcll(__ballot(in));
                        if (good + (n - i0 - 64) <= bound) break;
                    }
                    if (lane == 0) sGood[m] = good;
                }
            } else if (n > 7) {                                 // LMedS, n <= 14: one model per lane
                for (int m = t; m < 3 * drawn; m += 256) {
                    const int h = m / 3, k = m - 3 * h;
                    if (k >= sNm[h]) continue;
                    double F[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) F[i] = sF[h][9 * k + i];
                    float e[14];
#pragma unroll
                    for (int i = 0; i < 14; ++i) e[i] = 3.0e38f;
                    for (int i = 0; i < n; ++i) {               // insertion into the sorted prefix, register resident
                        float ax, ay, bx, by; pts.get(i, ax, ay, bx, by);
                        float v = epi_error(F, ax, ay, bx, by);
#pragma unroll
                        for (int p = 0; p < 14; ++p) { const float lo = fminf(e[p], v), hi = fmaxf(e[p], v); e[p] = lo; v = hi; }
                    }
                    float em1 = 0.f, e0 = 0.f;   // e[n/2 - 1], e[n/2]
#pragma unroll
                    for (int p = 0; p < 14; ++p) { em1 = p == n / 2 - 1 ? e[p] : em1; e0 = p == n / 2 ? e[p] : e0; }
                    sMed[m] = n % 2 != 0 ? (double)e0 : (double)(em1 + e0) * 0.5;
                }
            }
            __syncthreads();
            FM_T(3);
            // ---- accept, in sequence order
            if (t == 0) {
                int niters = sCtl[2], max_good = sCtl[3], done = sCtl[4];
                bool stop = sCtl[5] != 0 || drawn == 0;
                for (int h = 0; h < drawn; ++h) {
                    if (base + h >= niters) { stop = true; break; }
                    done = base + h + 1;
                    for (int k = 0; k < sNm[h]; ++k) {
                        const int m = 3 * h + k;
                        bool take = false;
                        if (n == 7) { take = !have_best; max_good = 7; }
                        else if (ransac) {
                            const int good = sGood[m];
                            if (good > (max_good > 6 ? max_good : 6)) {
                                take = true; max_good = good;
                                niters = update_num_iters(0.99, (double)(n - good) / n, 7, niters);
                            }
                        } else if (sMed[m] < min_median) { take = true; min_median = sMed[m]; }
                        if (take) { have_best = true; for (int i = 0; i < 9; ++i) bestF[i] = sF[h][9 * k + i]; }
                    }
                }
                if (base + drawn >= niters) stop = true;
                sCtl[1] = stop; sCtl[2] = niters; sCtl[3] = max_good; sCtl[4] = done;
                if (stop) {   // final threshold (squared) into sMed[0], verdict into sCtl[6]
                    double thr2 = 9.0;
                    if (!ransac && n > 7 && have_best) {
                        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(min_median);
                        sigma = fmax(sigma, 0.001);
                        thr2 = sigma * sigma;
                    }
                    sMed[0] = thr2; sCtl[6] = have_best ? 1 : 0;
                }
            }
            __syncthreads();
            FM_T(4);
            if (sCtl[1]) break;
        }
        // ---- mask of the winning matrix
        const bool have = sCtl[6] != 0;
        const double thr2 = sMed[0];
        double F[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) F[i] = bestF[i];
        int good = 0;
        for (int i0 = 0; i0 < n; i0 += 256) {
            const int i = i0 + t;
            bool in = false;
            if (i < n && have) {
                if (n == 7) in = true;
                else { float ax, ay, bx, by; pts.get(i, ax, ay, bx, by); in = epi_error(F, ax, ay, bx, by) <= thr2; }
            }
            if (i < n) mask[i] = in ? 1 : 0;
            good += (int)__popcll(__ballot(in));
        }
        if (lane == 0) sGood[w] = good;
        __syncthreads();
        good = sGood[0] + sGood[1] + sGood[2] + sGood[3];
        int verdict = have ? good : -1;
        if (have && !ransac && n > 7 && good < 7) verdict = -1;     // LMedS: fewer than 7 inliers is a failure
        if (t == 0) { a.counts[pair] = verdict; a.iters[pair] = sCtl[4]; }
        if (verdict < 0) for (int i = t; i < n; i += 256) mask[i] = 0;
        if (a.F && t < 9) a.F[9 * (size_t)pair + t] = verdict < 0 ? 0.0 : bestF[t];
        FM_T(5);
    }
}

}  // namespace

static int fmat_launch(rcn_ctx *ctx, int32_t n_pairs, const int32_t *off, const int32_t *xy1, const int32_t *xy2,
                       uint8_t *mask, int32_t *counts, int32_t *iters, double *F)
{
    if (n_pairs <= 0) return RCN_OK;
    FmatArgs a;
    a.off = off; a.xy1 = xy1; a.xy2 = xy2; a.n_pairs = n_pairs; a.mask = mask; a.counts = counts; a.iters = iters; a.F = F;
    const int blocks = std::min<int>(n_pairs, ctx->prop.multiProcessorCount * 24);
    k_fmat_filter<<<blocks, 256, 0, ctx->stream>>>(a);
    RCN_HIP(hipGetLastError());
    return RCN_OK;
}

extern "C" int rcn_fmat_filter_grid(rcn_ctx *ctx, int32_t n_pairs, const int32_t *pair_off, const int32_t *xy1,
                                    const int32_t *xy2, uint8_t *out_mask, int32_t *out_counts, int32_t *out_iterations,
                                    double *out_F)
{
    if (!ctx) return RCN_ERR_ARG;
    if (n_pairs < 0 || (n_pairs > 0 && (!pair_off || !out_counts))) { ctx->set_error("rcn_fmat_filter_grid: bad argument"); return RCN_ERR_ARG; }
    if (n_pairs == 0) return RCN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (pair_off[0] != 0) { ctx->set_error("rcn_fmat_filter_grid: pair_off[0] must be 0"); return RCN_ERR_ARG; }
    for (int p = 0; p < n_pairs; ++p)
        if (pair_off[p + 1] < pair_off[p]) { ctx->set_error("rcn_fmat_filter_grid: pair_off must be non-decreasing"); return RCN_ERR_ARG; }
    const size_t N = (size_t)pair_off[n_pairs];
    if (N > 0 && (!xy1 || !xy2 || !out_mask)) { ctx->set_error("rcn_fmat_filter_grid: bad argument"); return RCN_ERR_ARG; }
    RCN_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_off = 4 * ((size_t)n_pairs + 1), b_xy = 8 * N, b_mask = N, b_cnt = 4 * (size_t)n_pairs;
    const size_t b_F = out_F ? 72 * (size_t)n_pairs : 0;
    RCN_HIP(ctx->fm_ws.reserve(al(b_off) + 2 * al(b_xy) + al(b_mask) + 2 * al(b_cnt) + al(b_F) + 256));
    char *base = ctx->fm_ws.as<char>();
    size_t o = 0;
    auto take = [&](size_t b) { char *q = base + o; o += al(b); return q; };
    int32_t *d_off = (int32_t *)take(b_off), *d_1 = (int32_t *)take(b_xy), *d_2 = (int32_t *)take(b_xy);
    uint8_t *d_mask = (uint8_t *)take(b_mask);
    int32_t *d_cnt = (int32_t *)take(b_cnt), *d_it = (int32_t *)take(b_cnt);
    double *d_F = out_F ? (double *)take(b_F) : nullptr;
    RCN_HIP(hipMemcpyAsync(d_off, pair_off, b_off, hipMemcpyHostToDevice, st));
    if (N > 0) {
        RCN_HIP(hipMemcpyAsync(d_1, xy1, b_xy, hipMemcpyHostToDevice, st));
        RCN_HIP(hipMemcpyAsync(d_2, xy2, b_xy, hipMemcpyHostToDevice, st));
    }
    int rc = fmat_launch(ctx, n_pairs, d_off, d_1, d_2, d_mask, d_cnt, d_it, d_F);
    if (rc) return rc;
    if (N > 0) RCN_HIP(hipMemcpyAsync(out_mask, d_mask, b_mask, hipMemcpyDeviceToHost, st));
    RCN_HIP(hipMemcpyAsync(out_counts, d_cnt, b_cnt, hipMemcpyDeviceToHost, st));
    if (out_iterations) RCN_HIP(hipMemcpyAsync(out_iterations, d_it, b_cnt, hipMemcpyDeviceToHost, st));
    if (out_F) RCN_HIP(hipMemcpyAsync(out_F, d_F, b_F, hipMemcpyDeviceToHost, st));
    RCN_HIP(hipStreamSynchronize(st));
    return RCN_OK;
}

extern "C" int rcn_fmat_filter_grid_device(rcn_ctx *ctx, int32_t n_pairs, const int32_t *pair_off_dev, const int32_t *xy1_dev,
                                           const int32_t *xy2_dev, uint8_t *out_mask_dev, int32_t *out_counts_dev,
                                           int32_t *out_iterations_dev, double *out_F_dev)
{
    if (!ctx) return RCN_ERR_ARG;
    if (n_pairs < 0 || (n_pairs > 0 && (!pair_off_dev || !xy1_dev || !xy2_dev || !out_mask_dev || !out_counts_dev || !out_iterations_dev))) {
        ctx->set_error("rcn_fmat_filter_grid_device: bad argument");
        return RCN_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    return fmat_launch(ctx, n_pairs, pair_off_dev, xy1_dev, xy2_dev, out_mask_dev, out_counts_dev, out_iterations_dev, out_F_dev);
}

extern "C" int rcn_fmat_filter(rcn_ctx *ctx, const int32_t *xy1, const int32_t *xy2, int32_t n, uint8_t *out_mask,
                               int32_t *out_count, double *out_F)
{
    if (!ctx) return RCN_ERR_ARG;
    if (n < 0 || !out_count || (n > 0 && (!xy1 || !xy2 || !out_mask))) { ctx->set_error("rcn_fmat_filter: bad argument"); return RCN_ERR_ARG; }
    const int32_t off[2] = {0, n};
    return rcn_fmat_filter_grid(ctx, 1, off, xy1, xy2, out_mask, out_count, nullptr, out_F);
}
