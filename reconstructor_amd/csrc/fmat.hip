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
//   F1-F5  round-synchronous RANSAC / LMedS, every phase its own launch over ALL pairs (see the
//          comment above FmState): begin -> [solve -> score -> accept] x 32 -> finish.       [fp64 VALU]
#include "rcn_internal.h"

#include <cfloat>
#include <string>

namespace {

#define FM_B 32            // hypotheses per round
#define FM_NLDS 1024       // points kept in LDS; larger pairs read them from global memory
#define FM_MAX_ITERS 1000
#define FM_MAX_ATTEMPTS 10000
#define FM_FUSED_MAX_PAIRS 1024      // up to this many pairs the whole search of a pair runs in one workgroup of one launch (k_fm_pair)

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

// seven correspondences (s1, s2: 7 x 2 floats, in registers) -> up to three matrices in F (27 doubles).
// The 7x9 design matrix of every lane lives in LDS, element-major: element e of lane l at sA[e * 64 + l].
// Pivot rows and columns are data dependent, i.e. different in every lane -- indexed LDS addressing
// takes that in its stride, and with this layout lane l only ever touches banks 2l and 2l + 1
// whatever element it asks for, so the accesses stay conflict-free.
#define FM_AT(r, c) sA[((r) * 9 + (c)) * 64 + lane]
__device__ int run_7point(const float *s1, const float *s2, double *sA, int lane, double *F)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double x0 = s1[2 * i], y0 = s1[2 * i + 1], x1 = s2[2 * i], y1 = s2[2 * i + 1];
        FM_AT(i, 0) = x1 * x0; FM_AT(i, 1) = x1 * y0; FM_AT(i, 2) = x1;
        FM_AT(i, 3) = y1 * x0; FM_AT(i, 4) = y1 * y0; FM_AT(i, 5) = y1;
        FM_AT(i, 6) = x0; FM_AT(i, 7) = y0; FM_AT(i, 8) = 1;
    }
    // null space: Gauss-Jordan, complete pivoting, no row exchanges (oracle: null_space_7x9)
    unsigned row_used = 0, col_used = 0, prs = 0, pcs = 0;   // prs / pcs: 4 bits per step
    bool singular = false;
    for (int k = 0; k < 7; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const double v = fabs(FM_AT(r, c));
                const bool ok = !(row_used >> r & 1) && !(col_used >> c & 1) && v > best;
                best = ok ? v : best; pr = ok ? r : pr; pc = ok ? c : pc;
            }
        singular |= pr < 0;
        pr = pr < 0 ? 0 : pr; pc = pc < 0 ? 0 : pc;
        row_used |= 1u << pr; col_used |= 1u << pc;
        prs |= (unsigned)pr << (4 * k); pcs |= (unsigned)pc << (4 * k);
        double prow[9];
        const double inv = 1. / FM_AT(pr, pc);
#pragma unroll
        for (int c = 0; c < 9; ++c) { prow[c] = FM_AT(pr, c) * inv; FM_AT(pr, c) = prow[c]; }
        for (int r = 0; r < 7; ++r) {
            const double m = FM_AT(r, pc);
            if (r == pr || m == 0) continue;
#pragma unroll
            for (int c = 0; c < 9; ++c) FM_AT(r, c) -= m * prow[c];
        }
    }
    if (singular) return 0;
    int fc0 = -1, fc1 = -1;   // the two free columns, ascending
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const bool fr = !(col_used >> c & 1);
        fc1 = fr && fc0 >= 0 && fc1 < 0 ? c : fc1;
        fc0 = fr && fc0 < 0 ? c : fc0;
    }
    // basis vectors (free variable = 1, pivot variables = -A[pivot row][free column]) built in place in
    // LDS rows 0 and 1 of a scratch area: the matrix rows are dead once their two entries are read
    double v0[7], v1[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { const int pr = prs >> (4 * k) & 15; v0[k] = FM_AT(pr, fc0); v1[k] = FM_AT(pr, fc1); }
#pragma unroll
    for (int c = 0; c < 9; ++c) { FM_AT(0, c) = c == fc0 ? 1.0 : 0.0; FM_AT(1, c) = c == fc1 ? 1.0 : 0.0; }
#pragma unroll
    for (int k = 0; k < 7; ++k) { const int pc = pcs >> (4 * k) & 15; FM_AT(0, pc) = -v0[k]; FM_AT(1, pc) = -v1[k]; }
    double f1[9], f2[9], c[4], r[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) { f2[i] = FM_AT(1, i); f1[i] = FM_AT(0, i) - f2[i]; }
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

// ---------------------------------------------------------------------------------------------
// Round-synchronous form: every pair advances by up to FM_B hypotheses per round, and each phase of
// a round is its own launch over ALL pairs, so that every phase runs with full lanes:
//   k_fm_begin   one wave per pair: RNG seed, iteration budget, first round of samples
//   k_fm_solve   one lane per hypothesis (FM_B lanes = one pair, two pairs per wave): 7-point solve,
//                the design matrix of every lane in LDS (element-major: conflict-free indexed access)
//   k_fm_score   one workgroup per pair: <= 3 FM_B matrices against the pair's points (LDS), with
//                the exact pruning bound; LMedS pairs: one matrix per lane
//   k_fm_accept  one wave per pair: the reference's sequential accept / shrink logic over the round
//                (one lane), then the next round's samples, drawn exactly as the reference draws them
//                (index stream on the scalar unit, collinearity tests of a sample on 30 lanes)
//   k_fm_finish  per pair: mask of the winning matrix, count, iterations
// Rounds are enqueued back to back without host round trips: FM_MAX_ITERS / FM_B of them, each
// launch returning at once when no pair is active any more (a device counter).
struct FmState {            // per pair, structure of arrays
    unsigned long long *rng;
    int *niters, *max_good, *done, *stop, *have, *drawn, *fail, *base;
    double *min_med, *bestF;   // [P], [P][9]
    int *idx;                  // [P][FM_B][7]
    double *F;                 // [P][FM_B][27]
    int *nm;                   // [P][FM_B]
    int *good;                 // [P][3 FM_B]
    double *med;               // [P][3 FM_B]
    int *active;               // pairs still running
};

struct Pts {   // the pair's points: LDS copy when it fits, else the global arrays
    const float2 *l1, *l2;
    const int32_t *g1, *g2;
    __device__ __forceinline__ void get(int i, float &ax, float &ay, float &bx, float &by) const
    {
        if (l1) { const float2 a = l1[i], b = l2[i]; ax = a.x; ay = a.y; bx = b.x; by = b.y; }
        else { ax = (float)g1[2 * i]; ay = (float)g1[2 * i + 1]; bx = (float)g2[2 * i]; by = (float)g2[2 * i + 1]; }
    }
};

// The next round's samples of one pair, drawn by one WAVE exactly as the reference draws them:
// seven distinct indices per sample (a duplicate is redrawn), the whole sample redrawn while its last
// point is collinear with two earlier ones in either image.  The index stream runs on wave-uniform
// values (readfirstlane: scalar unit; x % n is a multiply-high by floor((2^32-1)/n) plus at most two
// corrections); the 2 x 15 collinearity tests of a sample run on 30 lanes at once out of a small LDS
// line.  sp: 28 floats of LDS scratch per wave.
__device__ void draw_round(const FmState &st, int pair, int n, const Pts &pts, int lane, float *sp)
{
#pragma clang fp contract(off)
    const unsigned long long rng0 = st.rng[pair];
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)rng0), hi = __builtin_amdgcn_readfirstlane((unsigned)(rng0 >> 32));
    const unsigned un = (unsigned)__builtin_amdgcn_readfirstlane(n), M = 0xFFFFFFFFu / un;
    const int niters = __builtin_amdgcn_readfirstlane(st.niters[pair]), base = __builtin_amdgcn_readfirstlane(st.base[pair]);
    int drawn = 0;
    bool failed = false;
    // lane l < 15 tests the pair (j, k), k < j < 6, number l; lanes 32..46 the same in the second image
    int tj = 1, tk = 0;
    {
        int l = lane & 31, j = 1;
        while (l >= j && j < 6) { l -= j; ++j; }
        tj = j; tk = l;
    }
    const bool tester = (lane & 31) < 15;
    for (; drawn < FM_B && base + drawn < niters; ++drawn) {
        int myidx = lane;                                   // lanes 0..6 hold the sample (n == 7: the data set itself)
        bool ok = false;
        for (int attempt = 0; attempt < FM_MAX_ATTEMPTS && !ok; ++attempt) {
            if (un != 7)
                for (int i = 0; i < 7;) {
                    const unsigned long long s = (unsigned long long)lo * 4164903690U + hi;
                    lo = (unsigned)s; hi = (unsigned)(s >> 32);
                    unsigned v = lo - __umulhi(lo, M) * un;
                    while (v >= un) v -= un;
                    if (__ballot(lane < i && myidx == (int)v)) continue;     // a duplicate is redrawn
                    myidx = lane == i ? (int)v : myidx;
                    ++i;
                }
            // the sample's points -> LDS line (lanes 0..6), then the tests
            if (lane < 7) {
                float ax, ay, bx, by;
                pts.get(myidx, ax, ay, bx, by);
                sp[2 * lane] = ax; sp[2 * lane + 1] = ay; sp[14 + 2 * lane] = bx; sp[14 + 2 * lane + 1] = by;
            }
            __builtin_amdgcn_wave_barrier();
            bool coll = false;
            if (tester && un != 7) {
                const float *m = sp + (lane >= 32 ? 14 : 0);
                const double dx1 = m[2 * tj] - m[12], dy1 = m[2 * tj + 1] - m[13];
                const double dx2 = m[2 * tk] - m[12], dy2 = m[2 * tk + 1] - m[13];
                coll = fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
            }
            ok = __ballot(coll) == 0;
            __builtin_amdgcn_wave_barrier();
        }
        if (!ok) { failed = true; break; }
        if (lane < 7) st.idx[((size_t)pair * FM_B + drawn) * 7 + lane] = myidx;
    }
    if (lane == 0) {
        st.rng[pair] = (unsigned long long)hi << 32 | lo;
        st.drawn[pair] = drawn;
        st.fail[pair] = failed ? 1 : 0;
    }
}

// the pair's points into LDS (when they fit): every sample costs an LDS read, not a trip to L2
__device__ __forceinline__ Pts load_points(const FmatArgs &a, int o0, int n, float2 *P1, float2 *P2, int t, int nthreads)
{
    Pts pts;
    pts.g1 = a.xy1 + 2 * (size_t)o0; pts.g2 = a.xy2 + 2 * (size_t)o0;
    pts.l1 = n <= FM_NLDS ? P1 : nullptr; pts.l2 = n <= FM_NLDS ? P2 : nullptr;
    if (n <= FM_NLDS)
        for (int i = t; i < n; i += nthreads) {
            P1[i] = make_float2((float)pts.g1[2 * i], (float)pts.g1[2 * i + 1]);
            P2[i] = make_float2((float)pts.g2[2 * i], (float)pts.g2[2 * i + 1]);
        }
    return pts;
}
// the state of a pair before its first round (one lane)
__device__ __forceinline__ void begin_state(const FmState &st, int pair, int n)
{
    st.base[pair] = 0; st.max_good[pair] = 0; st.done[pair] = 0; st.have[pair] = 0; st.fail[pair] = 0; st.drawn[pair] = 0;
    st.min_med[pair] = DBL_MAX;
    st.rng[pair] = ~0ull;
    int ni = FM_MAX_ITERS;
    if (n < 15) { ni = update_num_iters(0.99, 0.45, 7, FM_MAX_ITERS); if (ni < 3) ni = 3; }
    if (n == 7) ni = 1;
    if (n < 7) ni = 0;
    st.niters[pair] = ni; st.stop[pair] = n < 7 ? 1 : 0;
}

// one wave per pair
__global__ __launch_bounds__(64) void k_fm_begin(FmatArgs a, FmState st)
{
    __shared__ float sp[28];
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int o0 = a.off[pair], n = a.off[pair + 1] - o0;
    if (lane == 0) {
        begin_state(st, pair, n);
        if (n >= 7) atomicAdd(st.active, 1);
    }
    if (n < 7) return;
    __syncthreads();
    __shared__ float2 P1[FM_NLDS], P2[FM_NLDS];
    const Pts pts = load_points(a, o0, n, P1, P2, lane, 64);
    __syncthreads();
    draw_round(st, pair, n, pts, lane, sp);
}

// one lane per hypothesis (lane: position inside a 64-lane block of LDS columns, h: hypothesis of the pair)
__device__ __forceinline__ void solve_body(const FmatArgs &a, const FmState &st, int pair, int h, int lane, double *sA)
{
#pragma clang fp contract(off)
    const int o0 = a.off[pair];
    const int32_t *g1 = a.xy1 + 2 * (size_t)o0, *g2 = a.xy2 + 2 * (size_t)o0;
    const int *ix = st.idx + ((size_t)pair * FM_B + h) * 7;
    float s1[14], s2[14];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int v = ix[i];
        s1[2 * i] = (float)g1[2 * v]; s1[2 * i + 1] = (float)g1[2 * v + 1];
        s2[2 * i] = (float)g2[2 * v]; s2[2 * i + 1] = (float)g2[2 * v + 1];
    }
    st.nm[(size_t)pair * FM_B + h] = run_7point(s1, s2, sA, lane, st.F + ((size_t)pair * FM_B + h) * 27);
}
// FM_B lanes = one pair, two pairs per workgroup
__global__ __launch_bounds__(64) void k_fm_solve(FmatArgs a, FmState st)
{
    __shared__ double sA[63 * 64];
    if (*st.active == 0) return;
    const int lane = threadIdx.x, h = lane & (FM_B - 1);
    const int pair = blockIdx.x * (64 / FM_B) + lane / FM_B;
    const bool live = pair < a.n_pairs && !st.stop[pair];
    const int drawn = live ? st.drawn[pair] : 0;
    if (h >= drawn) return;
    solve_body(a, st, pair, h, lane, sA);
}

// one workgroup of 256 threads per pair; sCnt: 3 FM_B ints of LDS (RANSAC's bound of the round)
__device__ __forceinline__ void score_body(const FmState &st, int pair, int n, int drawn, const Pts &pts, int t, int *sCnt)
{
#pragma clang fp contract(off)
    const int lane = t & 63, w = t >> 6;
    const double *Fp = st.F + (size_t)pair * FM_B * 27;
    const int *nm = st.nm + (size_t)pair * FM_B;
    if (n >= 15) {
        // A matrix is accepted only if it beats the best count so far (and 6).  The best at the start
        // of the round is a lower bound of the best at this matrix's turn, so once even all remaining
        // points could not lift the count above it the matrix is dropped (its stored count stays
        // <= the bound: never accepted, exactly as if fully counted).
        // ... and the same holds for the counts of EARLIER matrices of this round: whichever of them
        // are finished when this matrix starts (sCnt, written by the four waves as they go) raise the
        // bound.  Later matrices must not: the reference may stop before it reaches them.
        for (int i = t; i < 3 * FM_B; i += 256) sCnt[i] = -1;
        __syncthreads();
        const int mg = st.max_good[pair], bound0 = mg > 6 ? mg : 6;
        for (int m = w; m < 3 * drawn; m += 4) {
            const int h = m / 3, k = m - 3 * h;
            if (k >= nm[h]) continue;
            int bound = bound0;
            {
                int b = max(lane < m ? __atomic_load_n(&sCnt[lane], __ATOMIC_RELAXED) : -1,
                            lane + 64 < m ? __atomic_load_n(&sCnt[(lane + 64) % (3 * FM_B)], __ATOMIC_RELAXED) : -1);
                for (int o = 32; o; o >>= 1) b = max(b, __shfl_xor(b, o));
                bound = max(bound, b);
            }
            double F[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) F[i] = Fp[27 * h + 9 * k + i];
            int good = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                bool in = false;
                if (i < n) { float ax, ay, bx, by; pts.get(i, ax, ay, bx, by); in = epi_inlier9(F, ax, ay, bx, by); }
                good += (int)__popcll(__ballot(in));
                if (good + (n - i0 - 64) <= bound) break;
            }
            if (lane == 0) { st.good[(size_t)pair * 3 * FM_B + m] = good; __atomic_store_n(&sCnt[m], good, __ATOMIC_RELAXED); }
        }
    } else {                                               // LMedS, 8 <= n <= 14: one matrix per lane
        for (int m = t; m < 3 * drawn; m += 256) {
            const int h = m / 3, k = m - 3 * h;
            if (k >= nm[h]) continue;
            double F[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) F[i] = Fp[27 * h + 9 * k + i];
            float e[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) e[i] = 3.0e38f;
            for (int i = 0; i < n; ++i) {                   // insertion into the sorted prefix, register resident
                float ax, ay, bx, by; pts.get(i, ax, ay, bx, by);
                float v = epi_error(F, ax, ay, bx, by);
#pragma unroll
                for (int p = 0; p < 14; ++p) { const float lo = fminf(e[p], v), hi = fmaxf(e[p], v); e[p] = lo; v = hi; }
            }
            float em1 = 0.f, e0 = 0.f;   // e[n/2 - 1], e[n/2]
#pragma unroll
            for (int p = 0; p < 14; ++p) { em1 = p == n / 2 - 1 ? e[p] : em1; e0 = p == n / 2 ? e[p] : e0; }
            st.med[(size_t)pair * 3 * FM_B + m] = n % 2 != 0 ? (double)e0 : (double)(em1 + e0) * 0.5;
        }
    }
}
__global__ __launch_bounds__(256) void k_fm_score(FmatArgs a, FmState st)
{
    __shared__ float2 P1[FM_NLDS], P2[FM_NLDS];
    __shared__ int sCnt[3 * FM_B];
    if (*st.active == 0) return;
    const int t = threadIdx.x, pair = blockIdx.x;
    if (st.stop[pair]) return;
    const int drawn = st.drawn[pair];
    if (drawn == 0) return;
    const int o0 = a.off[pair], n = a.off[pair + 1] - o0;
    if (n == 7) return;
    const Pts pts = load_points(a, o0, n, P1, P2, t, 256);
    __syncthreads();
    score_body(st, pair, n, drawn, pts, t, sCnt);
}

// the reference's sequential accept / shrink-the-budget logic over a finished round of one pair (ONE lane); returns "stop"
__device__ __forceinline__ bool accept_round(const FmState &st, int pair, int n)
{
#pragma clang fp contract(off)
    const bool ransac = n >= 15;
    const int drawn = st.drawn[pair], base = st.base[pair];
    int niters = st.niters[pair], max_good = st.max_good[pair], done = st.done[pair];
    bool have_best = st.have[pair] != 0;
    double min_median = st.min_med[pair];
    bool stop = st.fail[pair] != 0 || drawn == 0;
    const int *nm = st.nm + (size_t)pair * FM_B, *gd = st.good + (size_t)pair * 3 * FM_B;
    const double *md = st.med + (size_t)pair * 3 * FM_B, *Fp = st.F + (size_t)pair * FM_B * 27;
    for (int h = 0; h < drawn; ++h) {
        if (base + h >= niters) { stop = true; break; }
        done = base + h + 1;
        for (int k = 0; k < nm[h]; ++k) {
            const int m = 3 * h + k;
            bool take = false;
            if (n == 7) { take = !have_best; max_good = 7; }
            else if (ransac) {
                const int good = gd[m];
                if (good > (max_good > 6 ? max_good : 6)) {
                    take = true; max_good = good;
                    niters = update_num_iters(0.99, (double)(n - good) / n, 7, niters);
                }
            } else if (md[m] < min_median) { take = true; min_median = md[m]; }
            if (take) { have_best = true; for (int i = 0; i < 9; ++i) st.bestF[9 * (size_t)pair + i] = Fp[27 * h + 9 * k + i]; }
        }
    }
    if (base + drawn >= niters) stop = true;
    st.niters[pair] = niters; st.max_good[pair] = max_good; st.done[pair] = done; st.have[pair] = have_best ? 1 : 0;
    st.min_med[pair] = min_median; st.base[pair] = base + FM_B;
    if (stop) { st.stop[pair] = 1; st.drawn[pair] = 0; }
    return stop;
}
// one wave per pair: lane 0 replays the accept logic, then the wave draws the next round's samples
__global__ __launch_bounds__(64) void k_fm_accept(FmatArgs a, FmState st)
{
    __shared__ float sp[28];
    __shared__ int sStop;
    if (*st.active == 0) return;
    const int pair = blockIdx.x, lane = threadIdx.x;
    if (st.stop[pair]) return;
    const int o0 = a.off[pair], n = a.off[pair + 1] - o0;
    if (lane == 0) {
        const bool stop = accept_round(st, pair, n);
        if (stop) atomicSub(st.active, 1);
        sStop = stop;
    }
    __syncthreads();
    if (sStop) return;
    __shared__ float2 P1[FM_NLDS], P2[FM_NLDS];
    const Pts pts = load_points(a, o0, n, P1, P2, lane, 64);
    __syncthreads();
    draw_round(st, pair, n, pts, lane, sp);
}

// one workgroup of 256 threads per pair: mask of the winning matrix, verdict, iterations
__device__ __forceinline__ void finish_body(const FmatArgs &a, const FmState &st, int pair, int o0, int n, int t, int *sGood)
{
#pragma clang fp contract(off)
    const int lane = t & 63, w = t >> 6;
    uint8_t *mask = a.mask + o0;
    if (n < 7) {   // not filtered by the reference (SequentialReconstructor.cpp:237)
        for (int i = t; i < n; i += 256) mask[i] = 1;
        if (t == 0) { a.counts[pair] = -2; a.iters[pair] = 0; }
        if (a.F && t < 9) a.F[9 * (size_t)pair + t] = 0.0;
        return;
    }
    const bool ransac = n >= 15, have = st.have[pair] != 0;
    double thr2 = 9.0;
    if (!ransac && n > 7 && have) {
        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(st.min_med[pair]);
        sigma = fmax(sigma, 0.001);
        thr2 = sigma * sigma;
    }
    double F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = st.bestF[9 * (size_t)pair + i];
    const int32_t *g1 = a.xy1 + 2 * (size_t)o0, *g2 = a.xy2 + 2 * (size_t)o0;
    int good = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + t;
        bool in = false;
        if (i < n && have)
            in = n == 7 ? true : epi_error(F, (float)g1[2 * i], (float)g1[2 * i + 1], (float)g2[2 * i], (float)g2[2 * i + 1]) <= thr2;
        if (i < n) mask[i] = in ? 1 : 0;
        good += (int)__popcll(__ballot(in));
    }
    if (lane == 0) sGood[w] = good;
    __syncthreads();
    good = sGood[0] + sGood[1] + sGood[2] + sGood[3];
    int verdict = have ? good : -1;
    if (have && !ransac && n > 7 && good < 7) verdict = -1;     // LMedS: fewer than 7 inliers is a failure
    if (t == 0) { a.counts[pair] = verdict; a.iters[pair] = st.done[pair]; }
    if (verdict < 0) for (int i = t; i < n; i += 256) mask[i] = 0;
    if (a.F && t < 9) a.F[9 * (size_t)pair + t] = verdict < 0 ? 0.0 : st.bestF[9 * (size_t)pair + t];
}
__global__ __launch_bounds__(256) void k_fm_finish(FmatArgs a, FmState st)
{
    __shared__ int sGood[4];
    const int pair = blockIdx.x;
    const int o0 = a.off[pair], n = a.off[pair + 1] - o0;
    finish_body(a, st, pair, o0, n, threadIdx.x, sGood);
}

// The whole search of ONE pair in ONE workgroup (256 threads): begin, then rounds of solve (FM_B lanes) -> score (all four
// waves) -> accept + next draw (wave 0) until the pair stops, then finish -- the same device functions as the
// round-synchronous kernels above, phase by phase behind workgroup barriers, so the masks are the same bits.  One launch
// instead of 98: for the grids of the reference's own size (300 pairs: every launch of the round-synchronous form is
// shorter than its launch overhead) this is what the filter costs; large grids keep the round-synchronous form, whose
// solve phase packs two pairs per wave instead of idling seven of eight lanes (fmat_launch chooses).
__global__ __launch_bounds__(256) void k_fm_pair(FmatArgs a, FmState st)
{
    __shared__ double sA[63 * 64];
    __shared__ float2 P1[FM_NLDS], P2[FM_NLDS];
    __shared__ float sp[28];
    __shared__ int sCnt[3 * FM_B];
    __shared__ int sGood[4];
    __shared__ int sStop;
    const int pair = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int o0 = a.off[pair], n = a.off[pair + 1] - o0;
    if (t == 0) begin_state(st, pair, n);
    Pts pts = load_points(a, o0, n, P1, P2, t, 256);
    __syncthreads();
    if (n >= 7) {
        if (w == 0) draw_round(st, pair, n, pts, lane, sp);
        for (int round = 0; round < (FM_MAX_ITERS + FM_B - 1) / FM_B; ++round) {
            __syncthreads();                                   // the draw's indices (global, same CU) are visible to the workgroup
            const int drawn = st.drawn[pair];
            if (t < drawn) solve_body(a, st, pair, t, t, sA);
            __syncthreads();
            if (drawn > 0 && n != 7) score_body(st, pair, n, drawn, pts, t, sCnt);
            __syncthreads();
            if (t == 0) sStop = accept_round(st, pair, n) ? 1 : 0;
            __syncthreads();
            if (sStop) break;
            if (w == 0) draw_round(st, pair, n, pts, lane, sp);
        }
        __syncthreads();
    }
    finish_body(a, st, pair, o0, n, t, sGood);
}

}  // namespace

// host_poll: the caller synchronises with the host anyway (host-pointer API), so the number of pairs
// still running is read back every four rounds and the remaining launches are skipped once it is 0;
// the device-pointer API enqueues all rounds and never blocks.
static int fmat_launch(rcn_ctx *ctx, int32_t n_pairs, const int32_t *off, const int32_t *xy1, const int32_t *xy2,
                       uint8_t *mask, int32_t *counts, int32_t *iters, double *F, bool host_poll = false)
{
    if (n_pairs <= 0) return RCN_OK;
    FmatArgs a;
    a.off = off; a.xy1 = xy1; a.xy2 = xy2; a.n_pairs = n_pairs; a.mask = mask; a.counts = counts; a.iters = iters; a.F = F;
    const size_t P = (size_t)n_pairs;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t bytes = al(8 * P) + 8 * al(4 * P) + al(8 * P) + al(72 * P) + al(4 * P * FM_B * 7) + al(8 * P * FM_B * 27) +
                         al(4 * P * FM_B) + al(4 * P * 3 * FM_B) + al(8 * P * 3 * FM_B) + 256;
    RCN_HIP(ctx->fm_state.reserve(bytes));
    char *base = ctx->fm_state.as<char>();
    size_t o = 0;
    auto take = [&](size_t b) { char *q = base + o; o += al(b); return q; };
    FmState st;
    st.rng = (unsigned long long *)take(8 * P);
    st.niters = (int *)take(4 * P); st.max_good = (int *)take(4 * P); st.done = (int *)take(4 * P); st.stop = (int *)take(4 * P);
    st.have = (int *)take(4 * P); st.drawn = (int *)take(4 * P); st.fail = (int *)take(4 * P); st.base = (int *)take(4 * P);
    st.min_med = (double *)take(8 * P); st.bestF = (double *)take(72 * P);
    st.idx = (int *)take(4 * P * FM_B * 7); st.F = (double *)take(8 * P * FM_B * 27); st.nm = (int *)take(4 * P * FM_B);
    st.good = (int *)take(4 * P * 3 * FM_B); st.med = (double *)take(8 * P * 3 * FM_B);
    st.active = (int *)take(4);
    hipStream_t s = ctx->stream;
    RCN_HIP(hipMemsetAsync(st.bestF, 0, 72 * P, s));
    if (n_pairs <= FM_FUSED_MAX_PAIRS) {
        // few pairs: one launch, a workgroup walks its pair through every round (k_fm_pair)
        k_fm_pair<<<n_pairs, 256, 0, s>>>(a, st);
        RCN_HIP(hipGetLastError());
        return RCN_OK;
    }
    RCN_HIP(hipMemsetAsync(st.active, 0, 4, s));
    k_fm_begin<<<n_pairs, 64, 0, s>>>(a, st);
    for (int round = 0; round < (FM_MAX_ITERS + FM_B - 1) / FM_B; ++round) {
        k_fm_solve<<<(n_pairs + 64 / FM_B - 1) / (64 / FM_B), 64, 0, s>>>(a, st);
        k_fm_score<<<n_pairs, 256, 0, s>>>(a, st);
        k_fm_accept<<<n_pairs, 64, 0, s>>>(a, st);
        if (host_poll && (round & 3) == 3) {
            int active = 0;
            RCN_HIP(hipMemcpyAsync(&active, st.active, 4, hipMemcpyDeviceToHost, s));
            RCN_HIP(hipStreamSynchronize(s));
            if (active == 0) break;
        }
    }
    k_fm_finish<<<n_pairs, 256, 0, s>>>(a, st);
    RCN_HIP(hipGetLastError());
    return RCN_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused form for a device-resident match table (what rcn_match_grid_device leaves in HBM): the
// pair loop's lines SequentialReconstructor.cpp:237-269 for every pair without a host round trip.
//   k_tf_scan    exclusive scan of the per-pair match counts -> CSR offsets
//   k_tf_fill    per pair: matched features in ascending query order -> coordinates of both sides
//   (the RANSAC / LMedS rounds above)
//   k_tf_apply   per pair: drop the matches the filter rejected (all of them when no model was found),
//                leave pairs with fewer than 7 matches alone; new counts
namespace {

__global__ __launch_bounds__(1024) void k_tf_scan(const int32_t *counts, int n, int32_t *off)
{
    __shared__ int sh[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int carry = 0;
    if (t == 0) off[0] = 0;
    for (int b = 0; b < n; b += 1024) {
        const int i = b + t;
        int v = i < n ? counts[i] : 0;
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v += u; }
        if (lane == 63) sh[w] = v;
        __syncthreads();
        if (t < 16) { int s = sh[t]; for (int o = 1; o < 16; o <<= 1) { const int u = __shfl_up(s, o, 16); if (t >= o) s += u; } sh[t] = s; }
        __syncthreads();
        const int incl = v + (w ? sh[w - 1] : 0) + carry;
        if (i < n) off[i + 1] = incl;
        if (t == 1023) carry_s = incl;
        __syncthreads();
        carry = carry_s;
        __syncthreads();
    }
}

// ordered position of each flagged thread within the workgroup's 256-wide chunk; returns the chunk total
__device__ __forceinline__ int chunk_rank(bool flag, int &rank, int *sh)
{
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) sh[w] = (int)__popcll(m);
    __syncthreads();
    int base = 0, total = 0;
    for (int i = 0; i < 4; ++i) { base += i < w ? sh[i] : 0; total += sh[i]; }
    rank = base + (int)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(256) void k_tf_fill(const PairXY *px, const int32_t *table, int64_t stride, const int32_t *off,
                                                 int32_t *xy1, int32_t *xy2)
{
    __shared__ int sh[4];
    const int pair = blockIdx.x, t = threadIdx.x;
    const PairXY p = px[pair];
    const int32_t *row = table + (size_t)pair * stride;
    int pos = off[pair];
    for (int q0 = 0; q0 < p.Kq; q0 += 256) {
        const int q = q0 + t;
        const int tr = q < p.Kq ? row[q] : -1;
        int rank;
        const int total = chunk_rank(tr >= 0, rank, sh);
        if (tr >= 0) {
            const size_t o = 2 * (size_t)(pos + rank);
            xy1[o] = p.q[2 * q]; xy1[o + 1] = p.q[2 * q + 1];
            xy2[o] = p.t[2 * tr]; xy2[o + 1] = p.t[2 * tr + 1];
        }
        pos += total;
    }
}

__global__ __launch_bounds__(256) void k_tf_apply(const PairXY *px, int32_t *table, int64_t stride, const int32_t *off,
                                                  const uint8_t *mask, const int32_t *verdict, int32_t *counts)
{
    __shared__ int sh[4];
    const int pair = blockIdx.x, t = threadIdx.x;
    const PairXY p = px[pair];
    const int v = verdict[pair];
    if (v == -2) return;                                   // fewer than 7 matches: left as they are (:271-277)
    int32_t *row = table + (size_t)pair * stride;
    int pos = off[pair];
    for (int q0 = 0; q0 < p.Kq; q0 += 256) {
        const int q = q0 + t;
        const int tr = q < p.Kq ? row[q] : -1;
        int rank;
        const int total = chunk_rank(tr >= 0, rank, sh);
        if (tr >= 0 && (v < 0 || !mask[pos + rank])) row[q] = -1;
        pos += total;
    }
    if (t == 0) counts[pair] = v < 0 ? 0 : v;
}

}  // namespace

extern "C" int rcn_coords_upload(rcn_ctx *ctx, int32_t img_id, const int32_t *xy_host, int32_t K)
{
    if (!ctx) return RCN_ERR_ARG;
    if (K < 0 || (K > 0 && !xy_host)) { ctx->set_error("rcn_coords_upload: bad argument"); return RCN_ERR_ARG; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    auto &e = ctx->coords[img_id];
    RCN_HIP(e.first.reserve(std::max<size_t>(8 * (size_t)K, 8)));
    if (K > 0) {
        RCN_HIP(hipMemcpyAsync(e.first.p, xy_host, 8 * (size_t)K, hipMemcpyHostToDevice, ctx->stream));
        RCN_HIP(hipStreamSynchronize(ctx->stream));        // the host rows are borrowed
    }
    e.second = K;
    return RCN_OK;
}

// n images at once: one asynchronous copy per image, ONE host synchronisation for the batch (rcn_desc_upload_batch's counterpart)
extern "C" int rcn_coords_upload_batch(rcn_ctx *ctx, int32_t first_img_id, int32_t n_images, const int32_t *const *xy_host, const int32_t *K)
{
    if (!ctx) return RCN_ERR_ARG;
    if (n_images < 0 || (n_images > 0 && (!xy_host || !K))) { ctx->set_error("rcn_coords_upload_batch: bad argument"); return RCN_ERR_ARG; }
    for (int i = 0; i < n_images; ++i)
        if (K[i] < 0 || (K[i] > 0 && !xy_host[i])) { ctx->set_error("rcn_coords_upload_batch: bad row count or NULL coordinates"); return RCN_ERR_ARG; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < n_images; ++i) {
        auto &e = ctx->coords[first_img_id + i];
        RCN_HIP(e.first.reserve(std::max<size_t>(8 * (size_t)K[i], 8)));
        if (K[i] > 0) RCN_HIP(hipMemcpyAsync(e.first.p, xy_host[i], 8 * (size_t)K[i], hipMemcpyHostToDevice, ctx->stream));
        e.second = K[i];
    }
    RCN_HIP(hipStreamSynchronize(ctx->stream));        // the host rows are borrowed
    return RCN_OK;
}

extern "C" int rcn_coords_clear(rcn_ctx *ctx)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->coords) kv.second.first.release();
    ctx->coords.clear();
    return RCN_OK;
}

extern "C" int rcn_match_table_filter_device(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, int32_t *table_dev,
                                             int64_t stride, int32_t *counts_dev, int32_t *out_status_dev)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return rcn_int_table_filter(ctx, pairs_host, n_pairs, table_dev, stride, counts_dev, out_status_dev);
}

// the same with ctx->mu held by the caller (shard.hip, rcn_match_grid_filtered)
int rcn_int_table_filter(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, int32_t *table_dev,
                         int64_t stride, int32_t *counts_dev, int32_t *out_status_dev)
{
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs_host || !table_dev || !counts_dev)) || stride < 0) {
        ctx->set_error("rcn_match_table_filter_device: bad argument");
        return RCN_ERR_ARG;
    }
    if (n_pairs == 0) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    std::vector<PairXY> &px = ctx->fm_pairs_host;          // kept in the ctx: uploaded asynchronously
    px.resize((size_t)n_pairs);
    size_t cap = 0;
    for (int p = 0; p < n_pairs; ++p) {
        auto q = ctx->coords.find(pairs_host[2 * p]), t = ctx->coords.find(pairs_host[2 * p + 1]);
        if (q == ctx->coords.end() || t == ctx->coords.end()) {
            ctx->set_error("rcn_match_table_filter_device: coordinates of image " + std::to_string(pairs_host[2 * p + (q == ctx->coords.end() ? 0 : 1)]) + " are not resident");
            return RCN_ERR_NOT_FOUND;
        }
        // the table indexes descriptor rows: the coordinate lists must cover exactly those rows
        // (k_tf_fill reads p.t[2 * train row] for any train row the table names)
        auto dq = ctx->images.find(pairs_host[2 * p]), dt = ctx->images.find(pairs_host[2 * p + 1]);
        if ((dq != ctx->images.end() && dq->second.K != q->second.second) || (dt != ctx->images.end() && dt->second.K != t->second.second)) {
            ctx->set_error("rcn_match_table_filter_device: an image's coordinate count differs from its descriptor count");
            return RCN_ERR_ARG;
        }
        if (q->second.second > stride) { ctx->set_error("rcn_match_table_filter_device: stride smaller than a query image's keypoint count"); return RCN_ERR_ARG; }
        px[p].q = q->second.first.as<int32_t>(); px[p].t = t->second.first.as<int32_t>(); px[p].Kq = q->second.second; px[p].pad = 0;
        cap += (size_t)q->second.second;                   // a pair has at most Kq matches
    }
    hipStream_t st = ctx->stream;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t P = (size_t)n_pairs;
    RCN_HIP(ctx->fm_pairs.reserve(al(sizeof(PairXY) * P)));
    RCN_HIP(ctx->fm_csr.reserve(al(4 * (P + 1)) + 2 * al(8 * cap) + al(cap) + 2 * al(4 * P) + 256));
    char *base = ctx->fm_csr.as<char>();
    size_t o = 0;
    auto take = [&](size_t b) { char *q = base + o; o += al(b); return q; };
    int32_t *d_off = (int32_t *)take(4 * (P + 1)), *d_1 = (int32_t *)take(8 * cap), *d_2 = (int32_t *)take(8 * cap);
    uint8_t *d_mask = (uint8_t *)take(cap);
    int32_t *d_ver = out_status_dev ? out_status_dev : (int32_t *)take(4 * P), *d_it = (int32_t *)take(4 * P);
    RCN_HIP(hipMemcpyAsync(ctx->fm_pairs.p, px.data(), sizeof(PairXY) * P, hipMemcpyHostToDevice, st));
    const PairXY *d_px = ctx->fm_pairs.as<PairXY>();
    k_tf_scan<<<1, 1024, 0, st>>>(counts_dev, n_pairs, d_off);
    k_tf_fill<<<n_pairs, 256, 0, st>>>(d_px, table_dev, stride, d_off, d_1, d_2);
    RCN_HIP(hipGetLastError());
    int rc = fmat_launch(ctx, n_pairs, d_off, d_1, d_2, d_mask, d_ver, d_it, nullptr);
    if (rc) return rc;
    k_tf_apply<<<n_pairs, 256, 0, st>>>(d_px, table_dev, stride, d_off, d_mask, d_ver, counts_dev);
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
    int rc = fmat_launch(ctx, n_pairs, d_off, d_1, d_2, d_mask, d_cnt, d_it, d_F, true);
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
