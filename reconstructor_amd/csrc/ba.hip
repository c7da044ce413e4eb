// ba.hip -- bundle adjustment on MI355X (gfx950): kernels + C ABI (rcn_ba_solve).
//
// Replaces, behind include/rcn.h, what BundleAdjuster::adjust hands to Ceres
// (BundleAdjuster.cpp:72-146): Levenberg-Marquardt over camera poses, intrinsics and 3-d
// points with the reprojection residual of BundleAdjuster.h:27-58, point Schur complement
// and a dense Cholesky of the reduced camera system (DENSE_SCHUR, :132).  FP64 throughout.
//
// Kernels:
//   K4 k_ba_eval          residual (+ analytic 2x(10+3) tangent Jacobian) per observation   [HBM]
//   K5 k_ba_point_raw / k_ba_cam_raw   J'J blocks and J'r per point / per camera             [HBM]
//      k_ba_point_solve   V = Vs + D^2/radius, V^-1, per point                               [HBM]
//   K6 k_ba_wy + k_ba_schur_mfma[_wg] + k_ba_schur_diag_mfma   point Schur complement as f64-MFMA
//      gathers over per-block observation-pair lists built once per solve (k_pair_*), written
//      straight into the dense reduced system; bit-reproducible                      [L2 gathers / f64 MFMA]
//      (k_ba_schur + k_ba_S_assemble + k_ba_cam_rhs: the atomic form, RCN_BA_SCHUR_ATOMICS=1)
//   K7 k_chol_diag -> k_gemm_q<0> -> k_gemm_q<1> (latency chain) beside k_gemm_nt_pipe (panels and columns on a second
//      stream, bulk updates on a third): blocked right-looking Cholesky, v_mfma_f64_16x16x4_f64   [f64 MFMA]
//      k_trsv_bwd_chain (k_trsv_bwd per step as its fallback; k_trsv_fwd only when the rhs does not ride through the factorisation)
//   K8 k_ba_backsub, k_ba_model, k_ba_plus, reductions                                       [HBM]
// The LM control flow on the host follows Ceres' TrustRegionMinimizer / LevenbergMarquardt
// strategy step by step (same order of tests as the CPU restatement used for parity).
#include "rcn_internal.h"
#include <utility>
#include "ba_linesearch.h"

#include <cfloat>
#include <chrono>
#include <cmath>

#define NB 128        // Cholesky block size

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Observation rows.  Everything the solver keeps per observation -- 2x10 tangent camera Jacobian, 2x3 point Jacobian,
// residual -- is ONE 224-byte row, so that the per-camera gathers (k_ba_cam_raw) fetch whole rows, while the kernels
// that stream over all observations (k_ba_eval<true> writes the rows, k_ba_model reads them) move a wave's 64 rows as
// one contiguous 14-KiB block with 16-byte accesses and transpose through LDS: a lane-per-observation access to the
// rows themselves puts 64 lanes on 64 different lines per instruction (measured at 0.7 TB/s in round 2).
#define JROW 28          // doubles per row: [0..19] Jc (2 x 10), [20..25] Jp (2 x 3), [26..27] r
#define JCH (JROW / 2)   // 16-byte chunks per row
#define JLD 30           // LDS row stride in doubles: 240 B keeps rows 16-B aligned and ds_*_b128 by row conflict-free
#define WROW 30          // doubles per row of a W / Y plane
#define WCH (WROW / 2)

struct BaDev {
    int nc, np, no, n, npad, mode, use_wy;   // use_wy: unused (the per-observation W / Y table is always built)
    double ub;
    double *poses, *intr, *pts;        // current point
    double *poses2, *intr2, *pts2;     // candidate
    const double *uv;
    const int *ocam, *opt, *pt_off, *cam_obs_off, *cam_obs, *cam_off, *cam_dim, *cols;
    double *J;                         // per observation ONE row of JROW doubles: camera Jacobian 2x10 | point Jacobian 2x3 | residual 2
    double *Uraw, *gcraw, *Vraw, *gpraw; // unscaled J'J / J'r blocks
    double *sc, *sp, *dgc, *dgp;       // Jacobi scale, clamped diag(Js'Js)
    double *Vinv, *gps, *rhs, *S, *L, *Linv, *yc, *stc, *stp, *dlc, *dlp;   // S: reduced system, L: its sub-diagonal Cholesky tiles
    double *partial, *scal;            // reduction scratch, scalars
    double *csplit;                    // [n_cams][split][256] partial per-camera sums when a camera is split over workgroups
    double *WY;                        // two planes of [n_obs][3][10], camera index fastest: scaled W_o = Jc'Jp at WY, Y_o = W_o Vinv at WY + 30 n_obs
    double *tobs;                      // per observation W_o' y_c (3): the back-substitution's per-observation term
    double *crot;                      // per camera CROT doubles: the camera part of the rotation and its derivative (k_ba_cam_rot)
    unsigned *tickets;                 // 2 nc words, zero at rest: the workgroups a camera is split over finish their sums themselves (cam / Schur-diagonal kernels)
    double *crot2;                     // the same for the candidate poses (poses2), written by k_ba_plus; swapped with crot when the step is accepted
    int *flag;
};

// ---------------------------------------------------------------------------------------
// p = R(w) X as ceres::AngleAxisRotatePoint; R and d(RX)/dw = -R [X]x Jr(w)
__device__ __forceinline__ void rotate(const double *w, const double *X, double *p, double *R, double *dpdw, bool jac)
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (th2 > DBL_EPSILON) {
        const double th = sqrt(th2), c = cos(th), s = sin(th);
        const double n[3] = {w[0] / th, w[1] / th, w[2] / th};
        const double cr[3] = {n[1] * X[2] - n[2] * X[1], n[2] * X[0] - n[0] * X[2], n[0] * X[1] - n[1] * X[0]};
        const double tmp = (n[0] * X[0] + n[1] * X[1] + n[2] * X[2]) * (1.0 - c);
        for (int i = 0; i < 3; ++i) p[i] = X[i] * c + cr[i] * s + n[i] * tmp;
        if (!jac) return;
        const double hs = sin(0.5 * th), omc = 2.0 * hs * hs;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[3 * i + j] = omc * n[i] * n[j] + (i == j ? c : 0.0);
        R[1] -= s * n[2]; R[2] += s * n[1];
        R[3] += s * n[2]; R[5] -= s * n[0];
        R[6] -= s * n[1]; R[7] += s * n[0];
        double a, b;
        if (th < 1e-2) {
            a = 0.5 - th2 / 24.0 + th2 * th2 / 720.0;
            b = 1.0 / 6.0 - th2 / 120.0 + th2 * th2 / 5040.0;
        } else {
            a = omc / th2;
            b = (th - s) / (th2 * th);
        }
        const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        double Jr[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Jr[3 * i + j] = -a * K[3 * i + j] + b * (K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j]) + (i == j ? 1.0 : 0.0);
        const double Xx[9] = {0, -X[2], X[1], X[2], 0, -X[0], -X[1], X[0], 0};
        double M[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                M[3 * i + j] = -(R[3 * i] * Xx[j] + R[3 * i + 1] * Xx[3 + j] + R[3 * i + 2] * Xx[6 + j]);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                dpdw[3 * i + j] = M[3 * i] * Jr[j] + M[3 * i + 1] * Jr[3 + j] + M[3 * i + 2] * Jr[6 + j];
    } else {
        p[0] = X[0] + w[1] * X[2] - w[2] * X[1];
        p[1] = X[1] + w[2] * X[0] - w[0] * X[2];
        p[2] = X[2] + w[0] * X[1] - w[1] * X[0];
        if (!jac) return;
        R[0] = 1; R[1] = -w[2]; R[2] = w[1]; R[3] = w[2]; R[4] = 1; R[5] = -w[0]; R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
        dpdw[0] = 0; dpdw[1] = X[2]; dpdw[2] = -X[1]; dpdw[3] = -X[2]; dpdw[4] = 0; dpdw[5] = X[0];
        dpdw[6] = X[1]; dpdw[7] = -X[0]; dpdw[8] = 0;
    }
}

// The part of rotate() that depends on the camera alone -- R(w), Jr(w), cos, sin, the axis: trigonometry, a square root
// and divisions -- tabulated once per camera and evaluation (k_ba_cam_rot), so that the per-observation kernel is left
// with products: CROT doubles per camera, [0..8] R, [9..17] Jr, [18] cos, [19] sin, [20..22] axis, [23] 1 = small angle
// (first-order branch of ceres::AngleAxisRotatePoint), [24..26] w.
#define CROT 28
__device__ __noinline__ void cam_rot_entry(const double *w, double *o);
__global__ void k_ba_cam_rot(const double *__restrict__ poses, int nc, double *__restrict__ tab)
{
    const int cidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (cidx >= nc) return;
    cam_rot_entry(poses + 6 * (size_t)cidx, tab + CROT * (size_t)cidx);
}
// (one camera's entry: also called by k_ba_plus for the pose it has just written -- the table of a candidate that is accepted is then
//  already there when the evaluation with the Jacobian is queued: one launch less per iteration.  NOT inlined: both callers run the
//  same instructions, so a table does not depend on who wrote it -- inlined into k_ba_plus the compiler contracted other products
//  into FMAs than in k_ba_cam_rot, and long far-off solves took other paths than before)
__device__ __noinline__ void cam_rot_entry(const double *w, double *o)
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double R[9], Jr[9], c = 1.0, s = 0.0, n[3] = {0.0, 0.0, 0.0};
    const bool big = th2 > DBL_EPSILON;
    if (big) {
        const double th = sqrt(th2);
        c = cos(th); s = sin(th);
        n[0] = w[0] / th; n[1] = w[1] / th; n[2] = w[2] / th;
        const double hs = sin(0.5 * th), omc = 2.0 * hs * hs;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[3 * i + j] = omc * n[i] * n[j] + (i == j ? c : 0.0);
        R[1] -= s * n[2]; R[2] += s * n[1];
        R[3] += s * n[2]; R[5] -= s * n[0];
        R[6] -= s * n[1]; R[7] += s * n[0];
        double a, b;
        if (th < 1e-2) {
            a = 0.5 - th2 / 24.0 + th2 * th2 / 720.0;
            b = 1.0 / 6.0 - th2 / 120.0 + th2 * th2 / 5040.0;
        } else {
            a = omc / th2;
            b = (th - s) / (th2 * th);
        }
        const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Jr[3 * i + j] = -a * K[3 * i + j] + b * (K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j]) + (i == j ? 1.0 : 0.0);
    } else {
        R[0] = 1; R[1] = -w[2]; R[2] = w[1]; R[3] = w[2]; R[4] = 1; R[5] = -w[0]; R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
        for (int i = 0; i < 9; ++i) Jr[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    for (int i = 0; i < 9; ++i) { o[i] = R[i]; o[9 + i] = Jr[i]; }
    o[18] = c; o[19] = s; o[20] = n[0]; o[21] = n[1]; o[22] = n[2]; o[23] = big ? 0.0 : 1.0;
    o[24] = w[0]; o[25] = w[1]; o[26] = w[2]; o[27] = 0.0;
}

// rotate() with the camera part read from the table: the same expressions in the same order
__device__ __forceinline__ void rotate_tab(const double *__restrict__ cr, const double *X, double *p, double *R, double *dpdw)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = cr[i];
    if (cr[23] == 0.0) {
        const double c = cr[18], s = cr[19];
        const double n[3] = {cr[20], cr[21], cr[22]};
        const double cx[3] = {n[1] * X[2] - n[2] * X[1], n[2] * X[0] - n[0] * X[2], n[0] * X[1] - n[1] * X[0]};
        const double tmp = (n[0] * X[0] + n[1] * X[1] + n[2] * X[2]) * (1.0 - c);
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] = X[i] * c + cx[i] * s + n[i] * tmp;
        const double Xx[9] = {0, -X[2], X[1], X[2], 0, -X[0], -X[1], X[0], 0};
        double M[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                M[3 * i + j] = -(R[3 * i] * Xx[j] + R[3 * i + 1] * Xx[3 + j] + R[3 * i + 2] * Xx[6 + j]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                dpdw[3 * i + j] = M[3 * i] * cr[9 + j] + M[3 * i + 1] * cr[12 + j] + M[3 * i + 2] * cr[15 + j];
    } else {
        const double w[3] = {cr[24], cr[25], cr[26]};
        p[0] = X[0] + w[1] * X[2] - w[2] * X[1];
        p[1] = X[1] + w[2] * X[0] - w[0] * X[2];
        p[2] = X[2] + w[0] * X[1] - w[1] * X[0];
        dpdw[0] = 0; dpdw[1] = X[2]; dpdw[2] = -X[1]; dpdw[3] = -X[2]; dpdw[4] = 0; dpdw[5] = X[0];
        dpdw[6] = X[1]; dpdw[7] = -X[0]; dpdw[8] = 0;
    }
}

__device__ __forceinline__ double block_sum(double v, double *sh)
{
    const int t = threadIdx.x;
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o);
    if ((t & 63) == 0) sh[t >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (t == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    __syncthreads();
    return s;  // valid on thread 0
}

// The sums of a launch, finished BY the launch (round 4; a one-block k_finish_sum launch behind every reduction before: four launches
// of the ~35 of an LM iteration).  Every workgroup delivers its partial sums and takes a ticket; the one that draws the last
// ticket adds the partials up -- in index order, so the result does not depend on who that is -- and writes the scalars.
// The hand-off needs no L2-wide fence: a partial is an agent-coherent (sc1) store, drained (vmcnt(0)) before the ticket is
// taken, and the summing workgroup reads with agent-coherent loads (MI355X_MICROARCH.md, inter-workgroup visibility,
// valid forms).  The ticket word is left at zero for the next launch.
struct Fin {
    double *out;            // NS results (scaled), consecutive
    double *out2;           // optional second home of results 2 and 3 (k_ba_plus: gradient . delta, non-finite count)
    unsigned *ticket;
    const int *flag;        // optional: the factorisation's flag word rides home with the scalars of the step
    double *flag_out;
    double *zero;           // optional: a cell to clear (where the next gradient maximum is collected)
    double scale;
    // optional (round 5): the step's scalars straight into the host's pinned mirror -- scal[0 .. 14) as they stand when this, the
    // step's last reduction, is done, then the sequence number the host spins on.  A copy command plus a stream synchronisation took
    // ~27 us of a small problem's ~110-us iteration; a store over the fabric and a poll take a few.
    double *host;
    const double *scal;
    unsigned long long seq;
};
template <int NS>
__device__ __forceinline__ void finish_sums(const double (&mine)[NS], double *partial, int stride, const Fin &f, double *sh)
{
    __shared__ int s_last;
    const int t = threadIdx.x, nb = (int)gridDim.x;
    if (t == 0) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
            __hip_atomic_store(partial + (size_t)q * stride + blockIdx.x, mine[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned k = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = k == (unsigned)nb - 1u;
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        double v = 0.0;
        for (int i = t; i < nb; i += (int)blockDim.x) v += __hip_atomic_load(partial + (size_t)q * stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double s = block_sum(v, sh);
        if (t == 0) { f.out[q] = f.scale * s; if (f.out2 && q >= 2) f.out2[q - 2] = f.scale * s; }
    }
    if (t == 0) {
        __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f.flag) *f.flag_out = (double)__hip_atomic_load(f.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f.zero) *f.zero = 0.0;
        if (f.host) {
            // (the other cells were written by earlier kernels of the stream, and by this thread just above)
            for (int i = 0; i < 14; ++i) __hip_atomic_store(f.host + i, f.scal[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(f.host + 15), f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// residual of one observation and (JAC) its 2 x 15 Jacobian over [pose 6 | intrinsics 6 | point 3]
template <bool JAC>
__device__ __forceinline__ void obs_residual(const double *ps, const double *in, const double *X, const double *uv,
                                             double &r0, double &r1, double (*J)[15], const double *crot = nullptr)
{
    double p[3], R[9], dpdw[9];
    if (JAC && crot) rotate_tab(crot, X, p, R, dpdw);
    else rotate(ps, X, p, R, dpdw, JAC);
    p[0] += ps[3]; p[1] += ps[4]; p[2] += ps[5];
    const double iz = 1.0 / p[2];
    const double x = p[0] * iz, y = p[1] * iz;
    const double rr = x * x + y * y;
    const double dist = in[4] * rr + in[5] * rr * rr;
    const double xd = x + dist, yd = y + dist;
    r0 = in[0] * xd + in[2] - uv[0];
    r1 = in[1] * yd + in[3] - uv[1];
    if (JAC) {
        const double g = in[4] + 2.0 * in[5] * rr;
        const double a00 = 1.0 + 2.0 * g * x, a01 = 2.0 * g * y, a10 = 2.0 * g * x, a11 = 1.0 + 2.0 * g * y;
        const double b02 = -x * iz, b12 = -y * iz;
        double q[6];
        q[0] = in[0] * (a00 * iz); q[1] = in[0] * (a01 * iz); q[2] = in[0] * (a00 * b02 + a01 * b12);
        q[3] = in[1] * (a10 * iz); q[4] = in[1] * (a11 * iz); q[5] = in[1] * (a10 * b02 + a11 * b12);
        for (int i = 0; i < 2; ++i) {
            const double *qi = q + 3 * i;
            for (int k = 0; k < 3; ++k) {
                J[i][k] = qi[0] * dpdw[k] + qi[1] * dpdw[3 + k] + qi[2] * dpdw[6 + k];
                J[i][3 + k] = qi[k];
                J[i][12 + k] = qi[0] * R[k] + qi[1] * R[3 + k] + qi[2] * R[6 + k];
            }
        }
        J[0][6] = xd; J[0][7] = 0; J[0][8] = 1; J[0][9] = 0; J[0][10] = in[0] * rr; J[0][11] = in[0] * rr * rr;
        J[1][6] = 0; J[1][7] = yd; J[1][8] = 0; J[1][9] = 1; J[1][10] = in[1] * rr; J[1][11] = in[1] * rr * rr;
    }
}

// A wave's rows [row0, row0 + nrows) of a dense [rows][CH * 2] double array <-> its LDS block (row stride JLD), moved
// as 16-byte chunks in memory order: each instruction touches 1 KiB of consecutive addresses.
template <int CH>
__device__ __forceinline__ void rows_to_lds(const double *__restrict__ src, size_t row0, int nrows, double *__restrict__ lds, int lane)
{
    const f64x2 *g = reinterpret_cast<const f64x2 *>(src + row0 * (2 * CH));
    f64x2 v[CH];
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int c = it * 64 + lane;
        v[it] = c < nrows * CH ? g[c] : (f64x2){0.0, 0.0};
    }
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int c = it * 64 + lane, row = c / CH, part = c - row * CH;
        if (c < nrows * CH) *reinterpret_cast<f64x2 *>(lds + row * JLD + 2 * part) = v[it];
    }
}
template <int CH>
__device__ __forceinline__ void lds_to_rows(double *__restrict__ dst, size_t row0, int nrows, const double *__restrict__ lds, int lane)
{
    f64x2 *g = reinterpret_cast<f64x2 *>(dst + row0 * (2 * CH));
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int c = it * 64 + lane, row = c / CH, part = c - row * CH;
        if (c < nrows * CH) g[c] = *reinterpret_cast<const f64x2 *>(lds + row * JLD + 2 * part);
    }
}

// K4: residual and tangent Jacobian per observation; per-block partial of sum r^2.  JAC: every lane builds its row in
// LDS, the wave stores its 64 rows as one contiguous block (HBM-bound: 224 B written per observation).
template <bool JAC>
__global__ __launch_bounds__(JAC ? 128 : 256) void k_ba_eval(BaDev d, const double *poses, const double *intr,
                                                              const double *pts, double *partial, Fin fin)
{
    constexpr int NW = JAC ? 2 : 4;          // JAC: 30 KB of LDS per workgroup, five workgroups per CU
    __shared__ double sh[4];
    __shared__ __attribute__((aligned(16))) double stage[JAC ? NW * 64 * JLD : 2];
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double c2 = 0.0;
    if (o < d.no) {
        const int c = d.ocam[o], j = d.opt[o];
        double ps[6], in[6], X[3];
        for (int i = 0; i < 6; ++i) { ps[i] = poses[6 * c + i]; in[i] = intr[6 * c + i]; }
        for (int i = 0; i < 3; ++i) X[i] = pts[3 * j + i];
        double r0, r1, J[2][15];
        obs_residual<JAC>(ps, in, X, d.uv + 2 * (size_t)o, r0, r1, J, JAC ? d.crot + CROT * (size_t)c : nullptr);
        c2 = r0 * r0 + r1 * r1;
        if (JAC) {
            // tangent columns of this camera, WITHOUT indexing J by a run-time column (that would put J in scratch
            // memory): the host lays them out as the first `npose` pose columns (0, 3 or 6: camera 0 / camera 1 / the
            // others, BundleAdjuster.cpp:100-105) followed, in intrinsics mode 1, by fx fy k1 k2 = columns 6 7 10 11
            const int dc = d.cam_dim[c], nin = d.mode == 1 ? 4 : 0, npose = dc - nin;
            double row[JROW];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double I4[4] = {nin ? J[i][6] : 0.0, nin ? J[i][7] : 0.0, nin ? J[i][10] : 0.0, nin ? J[i][11] : 0.0};
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    const double a6 = k < 6 ? J[i][k] : I4[k - 6 < 4 ? k - 6 : 3];                    // npose == 6
                    const double a3 = k < 3 ? J[i][k] : (k - 3 < 4 ? I4[k - 3 < 4 ? k - 3 : 3] : 0.0);   // npose == 3
                    const double a0 = k < 4 ? I4[k < 4 ? k : 3] : 0.0;                                  // npose == 0
                    row[10 * i + k] = k < dc ? (npose == 6 ? a6 : npose == 3 ? a3 : a0) : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) row[20 + 3 * i + k] = J[i][12 + k];
            }
            row[26] = r0; row[27] = r1;
            f64x2 *my = reinterpret_cast<f64x2 *>(stage + (w * 64 + lane) * JLD);
#pragma unroll
            for (int k = 0; k < JCH; ++k) my[k] = (f64x2){row[2 * k], row[2 * k + 1]};
        }
    }
    if (JAC) {
        __syncthreads();
        const size_t row0 = (size_t)blockIdx.x * blockDim.x + 64 * w;
        const int nrows = (int)min((long long)64, (long long)d.no - (long long)row0);
        if (nrows > 0) lds_to_rows<JCH>(d.J, row0, nrows, stage + w * 64 * JLD, lane);
    }
    const double mine[1] = {block_sum(c2, sh)};
    finish_sums<1>(mine, partial, 0, fin, sh);
}

// Line search (bounds present): directional derivative of the cost at a trial point along the step held in
// d.dlc / d.dlp,  sum_o r_o' (Jc_o dlc + Jp_o dlp)  -- the tangent-space gradient J'r against the direction
// (LineSearchFunction::Evaluate); nothing is stored, the Jacobian of an observation lives in registers.
__global__ __launch_bounds__(256) void k_ba_dirgrad(BaDev d, const double *poses, const double *intr,
                                                     const double *pts, double *partial)
{
    __shared__ double sh[4];
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (o < d.no) {
        const int c = d.ocam[o], j = d.opt[o];
        double ps[6], in[6], X[3];
        for (int i = 0; i < 6; ++i) { ps[i] = poses[6 * c + i]; in[i] = intr[6 * c + i]; }
        for (int i = 0; i < 3; ++i) X[i] = pts[3 * j + i];
        double r[2], J[2][15];
        obs_residual<true>(ps, in, X, d.uv + 2 * (size_t)o, r[0], r[1], J);
        const int dc = d.cam_dim[c], off = d.cam_off[c];
        for (int i = 0; i < 2; ++i) {
            double m = 0.0;
            for (int k = 0; k < 10; ++k) if (k < dc) m += J[i][d.cols[10 * c + k]] * d.dlc[off + k];
            for (int k = 0; k < 3; ++k) m += J[i][12 + k] * d.dlp[3 * (size_t)j + k];
            acc += r[i] * m;
        }
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// max |v| over two arrays (the line search's |direction|_inf); *out zeroed by the caller.  Non-negative doubles
// order like their bit patterns: one integer atomicMax per workgroup, order-independent.
__global__ __launch_bounds__(256) void k_ba_absmax(const double *a, size_t na, const double *b, size_t nb, double *out)
{
    __shared__ double sh[4];
    double m = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * 256) {
        const double v = fabs(i < na ? a[i] : b[i - na]);
        if (v > m || v != v) m = v;
    }
    for (int o = 32; o; o >>= 1) { const double v = __shfl_down(m, o); if (v > m || v != v) m = v; }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) if (sh[i] > m || sh[i] != sh[i]) m = sh[i];
        atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(m));
    }
}

// out[slot] = scale * sum(partial[0..n)) in a fixed order (deterministic)
// (flag -> flag_out: the factorisation's flag word rides home with the scalars of the step instead of in a copy of its own)
__global__ __launch_bounds__(256) void k_finish_sum(const double *partial, int n, double *out, double scale, const int *flag = nullptr, double *flag_out = nullptr)
{
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    const double s = block_sum(v, sh);
    if (threadIdx.x == 0) { *out = scale * s; if (flag) *flag_out = (double)*flag; }
}
// K5: per point  Vraw = sum Jp'Jp (3x3), gpraw = sum Jp'r
__device__ __forceinline__ void point_raw_body(const BaDev &d, int j)
{
    if (j >= d.np) return;
    double V[9] = {0}, g[3] = {0};
    for (int o = d.pt_off[j]; o < d.pt_off[j + 1]; ++o)
        for (int i = 0; i < 2; ++i) {
            const double *q = d.J + JROW * (size_t)o + 20 + 3 * i;       // Jp and r: 64 contiguous bytes of the row
            const double ri = d.J[JROW * (size_t)o + 26 + i];
            for (int a = 0; a < 3; ++a) {
                g[a] += q[a] * ri;
                for (int b = 0; b < 3; ++b) V[3 * a + b] += q[a] * q[b];
            }
        }
    for (int a = 0; a < 9; ++a) d.Vraw[9 * (size_t)j + a] = V[a];
    for (int a = 0; a < 3; ++a) d.gpraw[3 * (size_t)j + a] = g[a];
}
__global__ void k_ba_point_raw(BaDev d)
{
    point_raw_body(d, blockIdx.x * blockDim.x + threadIdx.x);
}

// K5: per camera  Uraw = sum Jc'Jc (10x10), gcraw = sum Jc'r ; one workgroup per camera, eight groups of
// 128 threads: group g stages and sums the 32-observation chunks g, g+8, ... (a camera of the reference's own
// regime sees thousands of observations), the eight partial sums are added in group order.
// With few cameras (the reference's regime) a camera is further split over `split` workgroups so that
// the launch fills the chip; their partial sums go to d.csplit and are added in order by the workgroup that finishes last.
#define CR_GROUPS 8
// (point_blocks > 0: the launch also carries K5's per-point part -- the two read the same rows and depend on nothing of each other --
//  in that many workgroups BEHIND the cameras': one launch less per evaluation with a Jacobian)
__global__ __launch_bounds__(128 * CR_GROUPS) void k_ba_cam_raw(BaDev d, int split, int cam_blocks)
{
    __shared__ __attribute__((aligned(16))) double sh[CR_GROUPS][32 * 22];
    __shared__ double part[CR_GROUPS][110];
    if ((int)blockIdx.x >= cam_blocks) {       // (uniform over the workgroup: no barrier is skipped by part of it)
        point_raw_body(d, ((int)blockIdx.x - cam_blocks) * (128 * CR_GROUPS) + (int)threadIdx.x);
        return;
    }
    const int c = blockIdx.x / split, sidx = blockIdx.x - c * split, g = threadIdx.x >> 7, t = threadIdx.x & 127;
    // thread t < 110 of a group owns one entry: 0..99 of U (a = t/10, b = t%10), 100..109 of g
    double acc = 0.0;
    const int e0 = d.cam_obs_off[c], e1 = d.cam_obs_off[c + 1];
    for (int base0 = e0 + 32 * CR_GROUPS * sidx; base0 < e1; base0 += 32 * CR_GROUPS * split) {      // uniform trip count over the groups
        const int base = base0 + 32 * g;
        // stage 32 observations' (Jc 20 + r 2) rows in LDS
        __syncthreads();
        {   // 32 observations x (Jc 10 chunks + r 1 chunk) of 16 bytes over 128 threads: three per thread, the index
            // loads and then the row loads issued together
            int oo[3], prt[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = t + 128 * q, e = base + i / 11;
                prt[q] = i % 11;
                oo[q] = (i < 32 * 11 && e < e1) ? d.cam_obs[e] : -1;
            }
            f64x2 vv[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                vv[q] = oo[q] < 0 ? (f64x2){0.0, 0.0}
                                  : *reinterpret_cast<const f64x2 *>(d.J + JROW * (size_t)oo[q] + (prt[q] < 10 ? 2 * prt[q] : 26));
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (t + 128 * q < 32 * 11) *reinterpret_cast<f64x2 *>(&sh[g][((t + 128 * q) / 11) * 22 + 2 * prt[q]]) = vv[q];
        }
        __syncthreads();
        if (t < 110) {
            const int a = t < 100 ? t / 10 : t - 100, b = t % 10;
            for (int oi = 0; oi < 32; ++oi) {
                const double *row = sh[g] + oi * 22;
                if (t < 100) acc += row[a] * row[b] + row[10 + a] * row[10 + b];
                else acc += row[a] * row[20] + row[10 + a] * row[21];
            }
        }
    }
    if (t < 110) part[g][t] = acc;
    __syncthreads();
    if (g == 0 && t < 110) {
        double s = 0.0;
        for (int k = 0; k < CR_GROUPS; ++k) s += part[k][t];
        if (split > 1) __hip_atomic_store(d.csplit + ((size_t)c * split + sidx) * 256 + t, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (t < 100) d.Uraw[100 * (size_t)c + t] = s;
        else d.gcraw[10 * (size_t)c + (t - 100)] = s;
    }
    if (split > 1) {
        // the camera's workgroups finish the sum themselves (k_ba_cam_fin, a launch of its own, before): partials delivered as
        // agent-coherent stores and drained, one ticket per workgroup, the one that draws the last adds them in index order --
        // the hand-off of finish_sums, per camera
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(d.tickets + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)split - 1u;
        __syncthreads();
        if (s_last && g == 0 && t < 110) {
            double s = 0.0;
            for (int k = 0; k < split; ++k) s += __hip_atomic_load(d.csplit + ((size_t)c * split + k) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < 100) d.Uraw[100 * (size_t)c + t] = s;
            else d.gcraw[10 * (size_t)c + (t - 100)] = s;
            if (t == 0) __hip_atomic_store(d.tickets + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// Jacobi scaling (initial point) or clamped LM diagonal (scaled Jacobian) from the raw diagonals
__global__ void k_ba_diag(BaDev d, int what, double lo, double hi, int jacobi)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n) {
        // reduced coordinate i -> (camera, k): binary search over cam_off
        int a = 0, b = d.nc - 1;
        while (a < b) { int m = (a + b + 1) >> 1; if (d.cam_off[m] <= i) a = m; else b = m - 1; }
        while (d.cam_dim[a] == 0 || d.cam_off[a] + d.cam_dim[a] <= i) ++a;
        const int k = i - d.cam_off[a];
        const double u = d.Uraw[100 * (size_t)a + 11 * k];
        if (what == 0) d.sc[i] = jacobi ? 1.0 / (1.0 + sqrt(u)) : 1.0;
        else d.dgc[i] = fmin(fmax(u * d.sc[i] * d.sc[i], lo), hi);
    }
    const int j = i;
    if (j < 3 * d.np) {
        const double v = d.Vraw[9 * (size_t)(j / 3) + 4 * (j % 3)];
        if (what == 0) d.sp[j] = jacobi ? 1.0 / (1.0 + sqrt(v)) : 1.0;
        else d.dgp[j] = fmin(fmax(v * d.sp[j] * d.sp[j], lo), hi);
    }
}

// per point: V = scaled Vraw + dgp/radius, V^-1, scaled g_p
__device__ __forceinline__ void point_solve_body(const BaDev &d, int j, double inv_radius)
{
    double V[9];
    const double *s = d.sp + 3 * (size_t)j;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) V[3 * a + b] = d.Vraw[9 * (size_t)j + 3 * a + b] * s[a] * s[b];
    for (int a = 0; a < 3; ++a) V[4 * a] += d.dgp[3 * (size_t)j + a] * inv_radius;
    const double a = V[0], b = V[1], c = V[2], e = V[4], f = V[5], g = V[8];
    const double A = e * g - f * f, B = c * f - b * g, C = b * f - c * e;
    const double det = a * A + b * B + c * C;
    double Vi[9];
    if (!(det > 0.0) || !isfinite(det)) {
        d.flag[0] = 1;
        d.flag[1] = 1;      // (the trailing workgroups of the Schur diagonal launch clear the flag words BEHIND this kernel: they restore [0] from this word)
        for (int i = 0; i < 9; ++i) Vi[i] = 0.0;
    } else {
        const double id = 1.0 / det;
        Vi[0] = A * id; Vi[1] = B * id; Vi[2] = C * id;
        Vi[3] = Vi[1]; Vi[4] = (a * g - c * c) * id; Vi[5] = (b * c - a * f) * id;
        Vi[6] = Vi[2]; Vi[7] = Vi[5]; Vi[8] = (a * e - b * b) * id;
    }
    for (int i = 0; i < 9; ++i) d.Vinv[9 * (size_t)j + i] = Vi[i];
    for (int i = 0; i < 3; ++i) d.gps[3 * (size_t)j + i] = d.gpraw[3 * (size_t)j + i] * s[i];
}
__global__ void k_ba_point_solve(BaDev d, double inv_radius)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < d.np) point_solve_body(d, j, inv_radius);
}

// scaled W_o = (Jc' Jp) (10 x 3) of one observation
__device__ __forceinline__ void load_W(const BaDev &d, int o, int c, int j, double *W)
{
    const double *jc = d.J + JROW * (size_t)o, *jp = jc + 20;
    const int off = d.cam_off[c], dc = d.cam_dim[c];
    for (int a = 0; a < 10; ++a) {
        const double sa = a < dc ? d.sc[off + a] : 0.0;
        for (int b = 0; b < 3; ++b)
            W[3 * a + b] = (jc[a] * jp[b] + jc[10 + a] * jp[3 + b]) * sa * d.sp[3 * (size_t)j + b];
    }
}

// K6: Schur complement.  Contributions are accumulated in a camera-block-major buffer
// Sb[c][c2][10][10] (c2 <= c): one wave owns one observation pair (o,o2) of a point at a time and
// adds its 10x10 block -Y_o W_o2^T with two wave-wide f64 atomic instructions over 800
// contiguous bytes (full-rate shape: MI355X_MICROARCH.md, global float atomics).
__global__ __launch_bounds__(256) void k_ba_schur(BaDev d, double *Sb)
{
    const int j = blockIdx.x;                      // point
    const int o0 = d.pt_off[j], k = d.pt_off[j + 1] - o0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const double *Vi = d.Vinv + 9 * (size_t)j;
    const double sp0 = d.sp[3 * (size_t)j], sp1 = d.sp[3 * (size_t)j + 1], sp2 = d.sp[3 * (size_t)j + 2];
    for (int pr = w; pr < k * k; pr += nw) {
        const int o = o0 + pr / k, o2 = o0 + pr % k;
        const int c = d.ocam[o], c2 = d.ocam[o2];
        if (c2 > c) continue;
        const int dc = d.cam_dim[c], dc2 = d.cam_dim[c2];
        if (dc == 0 || dc2 == 0) continue;
        const int off = d.cam_off[c], off2 = d.cam_off[c2];
        const double *jc = d.J + JROW * (size_t)o, *jp = jc + 20;
        const double *jc2 = d.J + JROW * (size_t)o2, *jp2 = jc2 + 20;
        double *blk = Sb + ((size_t)c * d.nc + c2) * 100;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int e = lane + 64 * rep;
            if (e >= 100) break;
            const int a = e / 10, b = e % 10;
            if (a >= dc || b >= dc2) continue;
            const double sa = d.sc[off + a], sb = d.sc[off2 + b];
            const double ja0 = jc[a] * sa, ja1 = jc[10 + a] * sa, jb0 = jc2[b] * sb, jb1 = jc2[10 + b] * sb;
            // W[a][m] (this row of W_o), W2[b][m]
            const double wa0 = (ja0 * jp[0] + ja1 * jp[3]) * sp0, wa1 = (ja0 * jp[1] + ja1 * jp[4]) * sp1,
                         wa2 = (ja0 * jp[2] + ja1 * jp[5]) * sp2;
            const double wb0 = (jb0 * jp2[0] + jb1 * jp2[3]) * sp0, wb1 = (jb0 * jp2[1] + jb1 * jp2[4]) * sp1,
                         wb2 = (jb0 * jp2[2] + jb1 * jp2[5]) * sp2;
            const double y0 = wa0 * Vi[0] + wa1 * Vi[3] + wa2 * Vi[6];
            const double y1 = wa0 * Vi[1] + wa1 * Vi[4] + wa2 * Vi[7];
            const double y2 = wa0 * Vi[2] + wa1 * Vi[5] + wa2 * Vi[8];
            unsafeAtomicAdd(blk + e, -(y0 * wb0 + y1 * wb1 + y2 * wb2));
        }
    }
}

// ---- gather form of K6 (default): the structure of the observation graph is fixed for a
// solve, so the ordered observation pairs (o, o2) of every camera-pair block (c2 <= c) are
// listed once (count -> scan -> fill -> per-block sort = a fixed order), and every LM iteration
// one wave SUMS its block's contributions in registers and stores the 10x10 block once: no
// atomics, 0.4 GB of plain stores instead of 4.4 GB of atomic traffic, and the reduced system
// -- hence the whole solve -- is bit-reproducible from run to run.
__device__ __forceinline__ bool pair_key(const BaDev &d, int o, int o2, int &key)
{
    if (o == o2) return false;   // an observation with itself: walked from the per-camera list (k_ba_schur_diag_mfma)
    const int c = d.ocam[o], c2 = d.ocam[o2];
    if (c2 > c || d.cam_dim[c] == 0 || d.cam_dim[c2] == 0) return false;
    key = c * d.nc + c2;
    return true;
}

__global__ __launch_bounds__(256) void k_pair_count(BaDev d, int *cnt)
{
    const int j = blockIdx.x, o0 = d.pt_off[j], k = d.pt_off[j + 1] - o0;
    for (int pr = threadIdx.x; pr < k * k; pr += blockDim.x) {
        int key;
        if (pair_key(d, o0 + pr / k, o0 + pr % k, key)) atomicAdd(cnt + key, 1);
    }
}

// exclusive scan of n ints (n ~ nc^2) in three coalesced passes: per-chunk sums (one workgroup
// per 1024-element chunk), scan of the chunk sums by one workgroup, per-chunk scan with offset
__device__ __forceinline__ int block_scan_1024(int v, int *sh)   // inclusive scan over the workgroup's 1024 threads
{
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v += u; }
    if (lane == 63) sh[w] = v;
    __syncthreads();
    if (t < 16) { int s = sh[t]; for (int o = 1; o < 16; o <<= 1) { const int u = __shfl_up(s, o, 16); if (t >= o) s += u; } sh[t] = s; }
    __syncthreads();
    const int base = w ? sh[w - 1] : 0;
    __syncthreads();
    return v + base;
}
__global__ __launch_bounds__(1024) void k_scan_sums(const int *in, int *sums, int n)
{
    __shared__ int sh[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int tot = block_scan_1024(i < n ? in[i] : 0, sh);
    if (threadIdx.x == 1023) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_scan_top(int *sums, int nchunks, int *total)   // nchunks <= 1024*1024
{
    __shared__ int sh[16];
    int carry = 0;
    for (int b = 0; b < nchunks; b += 1024) {
        const int i = b + threadIdx.x;
        const int v = i < nchunks ? sums[i] : 0;
        const int inc = block_scan_1024(v, sh);
        if (i < nchunks) sums[i] = carry + inc - v;      // exclusive
        __shared__ int last;
        if (threadIdx.x == 1023) last = inc;
        __syncthreads();
        carry += last;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(1024) void k_scan_apply(const int *in, const int *sums, int *out, int n)
{
    __shared__ int sh[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    const int inc = block_scan_1024(v, sh);
    if (i < n) out[i] = sums[blockIdx.x] + inc - v;
}

__global__ __launch_bounds__(256) void k_pair_fill(BaDev d, const int *off, int *fill, unsigned long long *list)
{
    const int j = blockIdx.x, o0 = d.pt_off[j], k = d.pt_off[j + 1] - o0;
    for (int pr = threadIdx.x; pr < k * k; pr += blockDim.x) {
        const int o = o0 + pr / k, o2 = o0 + pr % k;
        int key;
        if (pair_key(d, o, o2, key)) list[off[key] + atomicAdd(fill + key, 1)] = ((unsigned long long)o << 32) | (unsigned)o2;
    }
}

// Count, scan and fill in ONE workgroup for the reference's own problem sizes (up to 32 cameras: nc^2 <= 1024 keys; round 5): the six
// launches above (clear, count, three scan passes, fill) were ~25 us of a solve that is ~400 us long, and the host spent as long again
// enqueueing them.  Per-key counters and fill cursors live in LDS; a thread walks the pairs of its landmarks.  The order inside a key
// is whatever the atomics gave -- as in k_pair_fill -- and the two sort kernels behind make it ascending.
__global__ __launch_bounds__(1024) void k_pair_small(BaDev d, int *off, unsigned long long *list, int nkeys, int *long_count)
{
    __shared__ int cnt[1024], cur[1024], sh[16];
    const int t = threadIdx.x;
    cnt[t] = 0; cur[t] = 0;
    if (t == 0) *long_count = 0;          // (the general path's clear covers this word)
    __syncthreads();
    // tp threads per landmark (a power of two): they share out its pairs
    int tp = 1;
    while (2 * tp * d.np <= 1024 && tp < 64) tp *= 2;
    const int sub = t & (tp - 1), jstep = 1024 / tp;
    for (int j = t / tp; j < d.np; j += jstep) {
        const int o0 = d.pt_off[j], k = d.pt_off[j + 1] - o0;
        for (int pr = sub; pr < k * k; pr += tp) {
            int key;
            if (pair_key(d, o0 + pr / k, o0 + pr % k, key)) atomicAdd(&cnt[key], 1);
        }
    }
    __syncthreads();
    const int v = t < nkeys ? cnt[t] : 0;
    const int inc = block_scan_1024(v, sh);
    if (t < nkeys) off[t] = inc - v;
    if (t == 1023) off[nkeys] = inc;
    __syncthreads();
    cnt[t] = inc - v;
    __syncthreads();
    for (int j = t / tp; j < d.np; j += jstep) {
        const int o0 = d.pt_off[j], k = d.pt_off[j + 1] - o0;
        for (int pr = sub; pr < k * k; pr += tp) {
            const int o = o0 + pr / k, o2 = o0 + pr % k;
            int key;
            if (pair_key(d, o, o2, key)) list[cnt[key] + atomicAdd(&cur[key], 1)] = ((unsigned long long)o << 32) | (unsigned)o2;
        }
    }
}

// fixed order inside every block: ascending (o, o2); segments are short except the diagonal blocks
#define PAIR_SORT_SHORT 32
// (long_keys / long_count: the keys whose segments are too long for a thread, listed for k_pair_sort_long -- round 5: that kernel was
//  launched with a workgroup per KEY, a million workgroups at cfg 5 of which none had anything to do: 0.41 ms of every solve that
//  builds its lists)
__global__ void k_pair_sort(const int *off, unsigned long long *list, int nkeys, int *long_keys, int *long_count)
{
    const int key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) return;
    unsigned long long *a = list + off[key];
    const int n = off[key + 1] - off[key];
    if (n > PAIR_SORT_SHORT) { long_keys[atomicAdd(long_count, 1)] = key; return; }           // long segments: k_pair_sort_long
    for (int i = 1; i < n; ++i) {
        const unsigned long long v = a[i];
        int p = i - 1;
        while (p >= 0 && a[p] > v) { a[p + 1] = a[p]; --p; }
        a[p + 1] = v;
    }
}

// Long segments (few cameras sharing thousands of landmarks: the reference's own regime, 3..25 views):
// one workgroup per segment, bitonic sort -- in LDS up to 4096 entries, in global memory beyond.
__global__ __launch_bounds__(256) void k_pair_sort_long(const int *off, unsigned long long *list, const int *long_keys, const int *long_count)
{
    __shared__ unsigned long long sh[4096];
    const int t = threadIdx.x, nlong = *long_count;
    for (int idx = blockIdx.x; idx < nlong; idx += gridDim.x) {      // (uniform over the workgroup)
    const int key = long_keys[idx];
    const int n = off[key + 1] - off[key];
    __syncthreads();                                               // the staging of the last key is free
    unsigned long long *g = list + off[key];
    int P = 64;
    while (P < n) P <<= 1;
    const bool in_lds = n <= 4096;
    if (in_lds) {
        for (int i = t; i < n; i += 256) sh[i] = g[i];
        __syncthreads();
    }
    unsigned long long *a = in_lds ? sh : g;
    // normalised bitonic network: every comparator ascending, the first step of a merge pairs i with its
    // mirror image i ^ (k - 1); entries beyond n count as +infinity and therefore never move
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < n; i += 256) {
                const int l = j == (k >> 1) ? (i ^ (k - 1)) : (i ^ j);
                if (l > i && l < n) {
                    const unsigned long long x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    if (in_lds) for (int i = t; i < n; i += 256) g[i] = sh[i];
    }
}

// ---- MFMA form of the gather (default).  k_ba_wy tabulates, per observation, the scaled
// W_o = Jc' Jp (10 x 3) and Y_o = W_o Vinv (both [m][a], camera coordinate fastest; rows beyond the
// camera's tangent size are zero).  A block (c, c2) is then  - [Y_o1 Y_o2 ...] [W_o1' W_o2' ...]^T,
// a 10 x 3P by 3P x 10 product: ceil(3P/4) v_mfma_f64_16x16x4_f64 steps per wave with two
// 8-byte gathers per lane and step, instead of ~30 loads and ~40 flops per lane and pair.
__global__ __launch_bounds__(256) void k_ba_wy(BaDev d)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int o = (int)(idx / 10), a = (int)(idx - 10 * (long)o);
    if (o >= d.no) return;
    const int c = d.ocam[o], j = d.opt[o];
    double w[3] = {0.0, 0.0, 0.0}, y[3] = {0.0, 0.0, 0.0};
    if (a < d.cam_dim[c]) {
        const double *jc = d.J + JROW * (size_t)o, *jp = jc + 20, *Vi = d.Vinv + 9 * (size_t)j;
        const double sa = d.sc[d.cam_off[c] + a];
        const double ja0 = jc[a] * sa, ja1 = jc[10 + a] * sa;
        for (int m = 0; m < 3; ++m) w[m] = (ja0 * jp[m] + ja1 * jp[3 + m]) * d.sp[3 * (size_t)j + m];
        for (int m = 0; m < 3; ++m) y[m] = w[0] * Vi[m] + w[1] * Vi[3 + m] + w[2] * Vi[6 + m];
    }
    double *outw = d.WY + WROW * (size_t)o, *outy = outw + WROW * (size_t)d.no;
    for (int m = 0; m < 3; ++m) { outw[10 * m + a] = w[m]; outy[10 * m + a] = y[m]; }
}

// acc -= [Y of the listed first observations] [W of the listed second observations]^T for `cnt`
// (<= 64) pairs held one per lane in `pr` (o << 32 | o2)
// RHS: column 10 of the B operand carries the scaled point gradient of the pair's landmark, so
// column 10 of the block comes out as  - sum Y_o gp_j(o)  -- the camera's reduced right-hand side.
// SMB: MFMA steps (of 4 k) per trip of the gather loop -- 4 for the short lists of the one-wave kernel,
// 12 (16 pairs' worth of gathers in flight) where a wave walks hundreds of pairs
template <bool RHS, int SMB>
__device__ __forceinline__ f64x4 schur_mfma_chunk(const double *__restrict__ Wt, const double *__restrict__ Yt, unsigned long long pr, int cnt, int lane, f64x4 acc,
                                                  const int *__restrict__ opt = nullptr, const double *__restrict__ gps = nullptr)
{
    const int i = lane & 15, kk = lane >> 4, K = 3 * cnt;
    for (int k0 = 0; k0 < K; k0 += 4 * SMB) {   // SMB MFMA steps per trip: 2 SMB gathers in flight per lane
        double a[SMB], b[SMB];
#pragma unroll
        for (int u = 0; u < SMB; ++u) {
            const int k = k0 + 4 * u + kk, p = k / 3, m = k - 3 * p;
            const unsigned long long e = __shfl(pr, p & 63);
            a[u] = 0.0; b[u] = 0.0;
            if (k < K && i < 10) {
                a[u] = Yt[WROW * (size_t)(e >> 32) + 10 * m + i];
                b[u] = Wt[WROW * (size_t)(e & 0xFFFFFFFFu) + 10 * m + i];
            }
            if (RHS && k < K && i == 10) b[u] = gps[3 * (size_t)opt[(unsigned)(e & 0xFFFFFFFFu)] + m];
        }
#pragma unroll
        for (int u = 0; u < SMB; ++u)
            if (k0 + 4 * u < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u], b[u], acc, 0, 0, 0);
    }
    return acc;
}

// (Round 5, measured and not kept -- profiles/r05_experiments/stream_kernels_schur_*.txt, 515-541 us per cfg-5 launch as it stands: more
//  gathers in flight per block, SMB = 8 / 12: 557 / 675 us, the extra trips are predicated-off instructions; the operands fetched as
//  whole 240-byte rows, 16 bytes per lane, through LDS: 674 us, same bits; two / three / four blocks per wave with all their gathers in
//  flight at once: 689 / 819 / 883 us.  Neither latency nor the number of load instructions is what bounds it.)
// one wave per strictly-lower block (c, c2 < c), written straight into the dense reduced system
template <int SMB>
__global__ __launch_bounds__(256) void k_ba_schur_mfma(BaDev d, const int *off, const unsigned long long *list)
{
    // (Round 4 tried dealing whole block rows to one XCD -- row c on XCD c % 8, so that camera c's Y rows are fetched into one
    // L2 instead of eight: 547 us against 515 us per cfg-5 launch.  The gathers are 8 bytes per lane out of 240-byte rows; the
    // kernel is bound by the address rate of the gather path, not by what is behind it.  Removed.)
    const int lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nlow = d.nc * (d.nc - 1) / 2;
    if (blk >= nlow) return;
    int c = (int)((sqrt(8.0 * blk + 1.0) + 1.0) * 0.5);
    while (c * (c + 1) / 2 <= blk) ++c;
    while (c * (c - 1) / 2 > blk) --c;
    const int c2 = blk - c * (c - 1) / 2, key = c * d.nc + c2;
    const int dc = d.cam_dim[c], dc2 = d.cam_dim[c2];
    if (dc == 0 || dc2 == 0) return;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int e0 = off[key], e1 = off[key + 1];
    for (int base = e0; base < e1; base += 64) {
        const int cnt = min(64, e1 - base);
        const unsigned long long pr = lane < cnt ? list[base + lane] : 0ull;
        acc = schur_mfma_chunk<false, SMB>(d.WY, d.WY + WROW * (size_t)d.no, pr, cnt, lane, acc);
    }
    // C/D layout: column = lane & 15, row = (lane >> 4) + 4 * reg
    const int col = lane & 15, r0 = lane >> 4;
    double *out = d.S + (size_t)d.cam_off[c] * d.npad + d.cam_off[c2];
    if (col < dc2)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            if (r0 + 4 * reg < dc) out[(size_t)(r0 + 4 * reg) * d.npad + col] = acc[reg];
}

// The same for long pair lists (few cameras sharing thousands of landmarks): one 8-wave workgroup per block,
// wave w takes the 64-pair chunks w, w+8, ...; the eight partial blocks are added in wave order.
__global__ __launch_bounds__(512) void k_ba_schur_mfma_wg(BaDev d, const int *off, const unsigned long long *list)
{
    __shared__ double part[8][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, blk = blockIdx.x;
    int c = (int)((sqrt(8.0 * blk + 1.0) + 1.0) * 0.5);
    while (c * (c + 1) / 2 <= blk) ++c;
    while (c * (c - 1) / 2 > blk) --c;
    const int c2 = blk - c * (c - 1) / 2, key = c * d.nc + c2;
    const int dc = d.cam_dim[c], dc2 = d.cam_dim[c2];
    if (dc == 0 || dc2 == 0) return;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int e0 = off[key], e1 = off[key + 1];
    for (int base = e0 + 64 * w; base < e1; base += 64 * 8) {
        const int cnt = min(64, e1 - base);
        const unsigned long long pr = lane < cnt ? list[base + lane] : 0ull;
        acc = schur_mfma_chunk<false, 12>(d.WY, d.WY + WROW * (size_t)d.no, pr, cnt, lane, acc);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) part[w][((lane >> 4) + 4 * reg) * 16 + (lane & 15)] = acc[reg];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int a = threadIdx.x >> 4, b = threadIdx.x & 15;
        if (a < dc && b < dc2) {
            double v = 0.0;
            for (int k = 0; k < 8; ++k) v += part[k][threadIdx.x];
            d.S[(size_t)(d.cam_off[c] + a) * d.npad + d.cam_off[c2] + b] = v;
        }
    }
}

#define RCN_RHS_BETA 1.0e200
// diagonal blocks: one 16-wave workgroup per camera; wave w takes the observations w, w+16, ... of
// the camera (a pair (o, o)) and the listed pairs of the key (c, c); the 16 partial blocks are
// summed in wave order and  scaled U + D/radius  is added before the store into S; column 10 of
// the same product is the camera's reduced right-hand side.
// (fin: -1 the right-hand side stays in d.rhs; 0 / 1 it goes straight into row n of the padded system -- a launch of its own, k_ba_S_finish,
//  did that until round 5 -- and with 1 the cell in d.rhs takes k_trsv_bwd_chain's "not there yet" pattern)
__device__ __forceinline__ void schur_diag_finish(const BaDev &d, int c, int idx, double v, double inv_radius, int fin = -1)
{
    const int a = idx >> 4, b = idx & 15, dc = d.cam_dim[c], offc = d.cam_off[c];
    if (a < dc && b < dc) {
        v += d.Uraw[100 * (size_t)c + 10 * a + b] * d.sc[offc + a] * d.sc[offc + b];
        if (a == b) v += d.dgc[offc + a] * inv_radius;
        d.S[(size_t)(offc + a) * d.npad + offc + b] = v;
    } else if (a < dc && b == 10) {   // reduced right-hand side: scaled gc - sum_o Y_o gp
        const double r = d.gcraw[10 * (size_t)c + a] * d.sc[offc + a] + v;
        if (fin < 0) d.rhs[offc + a] = r;
        else {
            d.S[(size_t)d.n * d.npad + offc + a] = r;
            if (fin == 1) reinterpret_cast<unsigned long long *>(d.rhs)[offc + a] = 0xFFFFFFFFFFFFFFFFull;
            else d.rhs[offc + a] = r;
        }
    }
}
// (`split` > 1: a camera's observations are spread over that many workgroups, see k_ba_cam_raw; SMB: gather
// depth, 12 in that latency-bound regime, 4 when a thousand cameras keep the chip full anyway)
// (fin >= 0, round 5: the launch also does everything that lies between the Schur build and the factorisation (round 4's k_ba_S_finish) -- the cameras'
//  workgroups put their right-hand sides into row n themselves (schur_diag_finish), the workgroups BEHIND them write the rest of the
//  padded rows and the factorisation's flag word and stream counters: one launch less per LM iteration)
template <int SMB>
__global__ __launch_bounds__(1024) void k_ba_schur_diag_mfma(BaDev d, const int *off, const unsigned long long *list, double inv_radius, int split, int fin = -1)
{
    __shared__ double part[16][256];
    if ((int)blockIdx.x >= d.nc * split) {      // (uniform over the workgroup)
        const size_t idx = (size_t)((int)blockIdx.x - d.nc * split) * 1024 + threadIdx.x, cnt = (size_t)(d.npad - d.n) * d.npad;
        if (idx == 0) { const int ps = d.flag[1]; d.flag[1] = 0; d.flag[0] = ps ? 1 : 0; }
        else if ((idx >= 2 && idx < 8) || (idx >= 12 && idx < 24)) d.flag[idx] = 0;
        if (idx >= cnt) return;
        const int i = d.n + (int)(idx / d.npad), j = (int)(idx % d.npad);
        if (i == d.n && j < d.n) return;        // the cameras' workgroups'
        d.S[(size_t)i * d.npad + j] = i == j ? (i == d.n ? RCN_RHS_BETA : 1.0) : 0.0;
        return;
    }
    const int c = blockIdx.x / split, sidx = blockIdx.x - c * split, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int dc = d.cam_dim[c], key = c * d.nc + c;
    if (dc == 0) return;
    const int W = 16 * split, ww = 16 * sidx + w;          // this wave's rank among all the camera's waves
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int o0 = d.cam_obs_off[c], o1 = d.cam_obs_off[c + 1];
    for (int base = o0 + ww; base < o1; base += W * 64) {
        const int cnt = min(64, (o1 - base + W - 1) / W);
        unsigned long long pr = 0ull;
        if (lane < cnt) { const unsigned o = (unsigned)d.cam_obs[base + W * lane]; pr = ((unsigned long long)o << 32) | o; }
        acc = schur_mfma_chunk<true, SMB>(d.WY, d.WY + WROW * (size_t)d.no, pr, cnt, lane, acc, d.opt, d.gps);
    }
    for (int base = off[key] + ww; base < off[key + 1]; base += W * 64) {   // the same camera seen twice by one landmark
        const int cnt = min(64, (off[key + 1] - base + W - 1) / W);
        const unsigned long long pr = lane < cnt ? list[base + W * lane] : 0ull;
        acc = schur_mfma_chunk<false, SMB>(d.WY, d.WY + WROW * (size_t)d.no, pr, cnt, lane, acc);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) part[w][((lane >> 4) + 4 * reg) * 16 + (lane & 15)] = acc[reg];
    __syncthreads();
    if (threadIdx.x < 256) {
        double v = 0.0;
        for (int k = 0; k < 16; ++k) v += part[k][threadIdx.x];
        if (split > 1) __hip_atomic_store(d.csplit + ((size_t)c * split + sidx) * 256 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else schur_diag_finish(d, c, threadIdx.x, v, inv_radius, fin);
    }
    if (split > 1) {      // as in k_ba_cam_raw: the last of the camera's workgroups adds the partial blocks, in index order (k_ba_schur_diag_fin before)
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(d.tickets + d.nc + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)split - 1u;
        __syncthreads();
        if (s_last && threadIdx.x < 256) {
            double v = 0.0;
            for (int k = 0; k < split; ++k) v += __hip_atomic_load(d.csplit + ((size_t)c * split + k) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            schur_diag_finish(d, c, threadIdx.x, v, inv_radius, fin);
            if (threadIdx.x == 0) __hip_atomic_store(d.tickets + d.nc + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// padded rows of the dense system: identity (the gather form writes every other lower block itself)
__global__ void k_ba_S_pad(BaDev d)
{
    const int i = d.n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.npad) d.S[(size_t)i * d.npad + i] = 1.0;
}

// Forward substitution for free: the reduced right-hand side rides through the factorisation as
// row n of the padded system (a padding row: zero elsewhere) under a huge diagonal entry, so the
// panel kernels leave  y = L^-1 b  in that row of the factor:  [S b; b' beta] = [L 0; y' .][L 0; y' .]'.
__global__ void k_ba_S_rhs_row(BaDev d)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > d.n) return;
    d.S[(size_t)d.n * d.npad + j] = j < d.n ? d.rhs[j] : RCN_RHS_BETA;
}
// y out of row n of the factor: sub-diagonal tiles live in L, the last diagonal tile in S
__global__ void k_ba_y_from_row(BaDev d, int sentinel)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d.npad) return;
    const int last0 = (d.npad / NB - 1) * NB;
    d.yc[j] = j < d.n ? (j < last0 ? d.L : d.S)[(size_t)d.n * d.npad + j] : 0.0;
    if (sentinel) reinterpret_cast<unsigned long long *>(d.rhs)[j] = 0xFFFFFFFFFFFFFFFFull;      // k_trsv_bwd_chain's "not there yet" (the right-hand side itself went into the system's last row)
}

// dense padded reduced system from the block buffer: lower blocks (c2 <= c) of
// S = blockdiag(scaled U + dgc/radius) + Sb ; padded diagonal = 1.  One workgroup per camera row.
__global__ __launch_bounds__(256) void k_ba_S_assemble(BaDev d, const double *Sb, double inv_radius)
{
    const int c = blockIdx.x, t = threadIdx.x;
    if (c >= d.nc) {   // padded rows: identity
        for (int i = d.n + t; i < d.npad; i += 256)
            if (c == d.nc) d.S[(size_t)i * d.npad + i] = 1.0;
        return;
    }
    const int dc = d.cam_dim[c], off = d.cam_off[c];
    if (dc == 0) return;
    for (int idx = t; idx < (c + 1) * 100; idx += 256) {
        const int c2 = idx / 100, e = idx % 100, a = e / 10, b = e % 10;
        const int dc2 = d.cam_dim[c2];
        if (a >= dc || b >= dc2) continue;
        const int off2 = d.cam_off[c2];
        double v = Sb[((size_t)c * d.nc + c2) * 100 + e];
        if (c2 == c) {
            v += d.Uraw[100 * (size_t)c + e] * d.sc[off + a] * d.sc[off + b];
            if (a == b) v += d.dgc[off + a] * inv_radius;
        }
        d.S[(size_t)(off + a) * d.npad + off2 + b] = v;
    }
}

// reduced rhs per camera: scaled gc - sum_o Y_o gp_j(o)   (deterministic order)
__global__ __launch_bounds__(64) void k_ba_cam_rhs(BaDev d)
{
    const int c = blockIdx.x, lane = threadIdx.x;
    const int dc = d.cam_dim[c], off = d.cam_off[c];
    if (dc == 0) return;
    double acc[10] = {0};
    for (int e = d.cam_obs_off[c] + lane; e < d.cam_obs_off[c + 1]; e += 64) {
        const int o = d.cam_obs[e], j = d.opt[o];
        double W[30];
        load_W(d, o, c, j, W);
        const double *Vi = d.Vinv + 9 * (size_t)j, *g = d.gps + 3 * (size_t)j;
        const double t0 = Vi[0] * g[0] + Vi[1] * g[1] + Vi[2] * g[2];
        const double t1 = Vi[3] * g[0] + Vi[4] * g[1] + Vi[5] * g[2];
        const double t2 = Vi[6] * g[0] + Vi[7] * g[1] + Vi[8] * g[2];
        for (int a = 0; a < dc; ++a) acc[a] += W[3 * a] * t0 + W[3 * a + 1] * t1 + W[3 * a + 2] * t2;
    }
    for (int a = 0; a < 10; ++a)
        for (int o = 32; o; o >>= 1) acc[a] += __shfl_down(acc[a], o);
    if (lane == 0)
        for (int a = 0; a < dc; ++a) d.rhs[off + a] = d.gcraw[10 * (size_t)c + a] * d.sc[off + a] - acc[a];
}

// one thread: relaxed poll (an sc1 load) with a 2-second limit (100 MHz wall clock, independent of the shader clock)
#ifdef RCN_DIAG
// Diagnostic build only: a device-side timeline of the factorisation's kernels (tools/chol_device_timeline.py).  rocprofv3's kernel
// trace stretches dependent launches and cross-stream hand-offs by tens of microseconds, which is the very thing to be measured.
// Slot 3 * id: first workgroup entered; + 1: its gate passed; + 2: last workgroup left.  id = 8 * block step + kind.
__device__ unsigned long long *g_tl = nullptr;
#define TL_MARK(id, w) do { if (g_tl && threadIdx.x == 0 && (blockIdx.x == 0 || (w) == 2)) atomicMax(&g_tl[3 * (id) + (w)], (unsigned long long)wall_clock64()); } while (0)
extern "C" int rcn_diag_timeline_set(unsigned long long *dev_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : -1;
}
#else
#define TL_MARK(id, w)
#endif
// The flag word is STICKY: the first event of a factorisation owns it (0 -> code by compare-and-swap), so a gate timeout (3) is
// never overwritten by the non-finite pivot (1) that the kernels behind it may then meet on half-updated tiles -- the host must
// see the 3 to switch schedules -- and a wait that finds the flag already raised returns at once: the result is discarded
// anyway, and every later hand-off of the same factorisation would otherwise burn its own 2 s.
__device__ __forceinline__ void flag_raise(int *flag, int code)
{
    (void)atomicCAS(flag, 0, code);
}
#ifdef RCN_DIAG
__device__ int g_poll[2] = {0, 1};      // {the time-out by the clock (0) or by a count of polls (1), s_sleep(8) per poll}: tools/ only
extern "C" int rcn_diag_set_poll(int mode, int sleeps)
{
    const int h[2] = {mode, sleeps};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_poll), h, sizeof(h)) == hipSuccess ? 0 : -1;
}
#endif
__device__ __forceinline__ void ring_wait(const int *counter, int need, int *flag, int code = 3)
{
#ifdef RCN_DIAG
    const int mode = g_poll[0], sleeps = g_poll[1];
#else
    const int mode = 0, sleeps = 1;
#endif
    const unsigned long long t0 = mode == 0 ? wall_clock64() : 0ull;
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        for (int q = 0; q < sleeps; ++q) __builtin_amdgcn_s_sleep(8);
        if (mode == 0 ? wall_clock64() - t0 > 200000000ull : ++spins > 2000000u) { flag_raise(flag, code); break; }
    }
}

// Cross-stream hand-offs of the factorisation: every stream owns counters that say how far it has come, and a kernel
// that needs another stream's result waits for the counter itself.
//   publish  a kernel's first thread stores the counter of the work that PRECEDES it on its stream: stream order has
//            completed that work and the kernel boundary has released its writes, so the store needs no fence and costs
//            the publishing stream nothing (a trailing signal kernel would cost ~5 us of the chain per step);
//   wait     thread 0 of every workgroup polls (relaxed), then ONE agent-scope acquire, then the workgroup barrier:
//            the consumer recipe of MI355X_MICROARCH.md (inter-workgroup visibility), after which plain loads are safe.
struct Gate { const int *c[6]; int n[6]; int nw; int *pub; int pubval; int *flag; };
__host__ __device__ inline Gate gate_none(int *flag)
{
    Gate g = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0, 0, 0}, 0, nullptr, 0, flag};
    return g;
}
__device__ __forceinline__ void gate_enter(const Gate &g)
{
    if (threadIdx.x == 0) {
        if (g.pub && blockIdx.x == 0) __hip_atomic_store(g.pub, g.pubval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g.nw > 0) ring_wait(g.c[0], g.n[0], g.flag);
        if (g.nw > 1) ring_wait(g.c[1], g.n[1], g.flag);
        if (g.nw > 2) ring_wait(g.c[2], g.n[2], g.flag);
        if (g.nw > 3) ring_wait(g.c[3], g.n[3], g.flag);
        if (g.nw > 4) ring_wait(g.c[4], g.n[4], g.flag);
        if (g.nw > 5) ring_wait(g.c[5], g.n[5], g.flag);
        if (g.nw > 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (g.nw > 0) __syncthreads();
}

__global__ __launch_bounds__(64) void k_ring_gate(Gate g)
{
    gate_enter(g);
}
#ifdef RCN_DIAG
// RCN_CHOL_BREAK=2 (diagnostic build): what a broken hand-off does to the numbers -- the diagonal block behind it holds garbage
__global__ void k_diag_poison(double *S, int ld, int kb)
{
    S[((size_t)kb * NB) * ld + (size_t)kb * NB] = __longlong_as_double(0x7ff8000000000000ll);
}
#endif

// ---------------------------------------------------------------------------------------
// K7: dense Cholesky of the padded reduced system (npad multiple of 128), lower triangle.
// Diagonal block (one workgroup of eight waves, block resident in LDS, 16 workgroup barriers in total):
//   1. blocked factorisation with 16-wide leaves.  A leaf is factored AND inverted by wave 0 alone, on the matrix pipe
//      (leaf_factor below); rows below become A Dinv^T and the trailing square is updated as 16x16
//      v_mfma_f64_16x16x4_f64 tiles.  Lookahead: wave 0 updates the next leaf's diagonal tile first and factors it
//      while waves 1..7 finish the trailing square, so the serial leaf work hides behind the MFMA work;
//   2. the inverse of the factor grows a block row per leaf in the same phase, by waves 1..7, also behind the leaf:
//          Linv[k][j] = -Dinv_k * sum_{m = j .. k-1} L[k][m] Linv[m][j]
//      (round 2 inverted by doubling, 16 -> 32 -> 64 -> 128, AFTER the factorisation: 11 us of the kernel's 58).
//      Leaf inverses sit in place on the diagonal (the factor's own leaf blocks wait transposed above it, diagonal in rd);
//      the other inverse blocks sit in the mirror position above the diagonal, the factor stays below it.
// Writes L^-1 into Linv[kb] (used by the panel GEMM and the triangular solves) as it appears and, on request, L into S.
#define DL 129   // LDS row stride of the diagonal block (doubles): row walks are conflict-free
#define LB 16    // leaf size
#ifdef RCN_STAMP   // diagnostic build only (tools/chol_diag_bench.hip): phase time stamps
__device__ unsigned long long g_stamps[64];
#define STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) g_stamps[i] = clock64(); } while (0)
#else
#define STAMP(i)
#endif
// ring_done / ring_need: in the chain-bound steps of the factorisation (the host decides) this kernel also does the gate's
// job on its way out -- by then the bulk update the first trailing column of this step waits for has long finished, so
// the check is free and the chain loses a 5-us kernel; ring_need < 0: nothing to wait for here.
#define CDW 8          // waves of the diagonal kernel: wave 0 owns the leaves, the others the matrix work between them.  Eight since the
                       // leaf moved to the matrix pipe (136 registers; the vector-ALU leaf of round 2 wanted ~340 and spilled at
                       // eight waves); the doubling steps (2c) deal their tiles to exactly eight waves
// nact: the block's ACTIVE rows -- below them it is the identity padding of the system (the last block of every factorisation; the only
// block of the reference's own problem sizes: 3 cameras are 13 rows of 128).  Only the leaves that hold active rows are factored; the
// rest of the block is its own factor and inverse, and is written as such.
// (the body: k_chol_diag = one block per launch on the chain's stream; k_chol_diag_server = every block of a factorisation in ONE
//  resident workgroup -- round 5, below.  false: a pivot broke down, flag 1 is raised)
__device__ __forceinline__ bool chol_diag_body(double *S, int ld, int kb, double *Linv, int *flag, int store_L, int nact, [[maybe_unused]] int tl)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *L = reinterpret_cast<double *>(smem_raw);  // [128][DL]
    if (threadIdx.x >= 64) __builtin_amdgcn_s_setprio(2);   // wave 0 carries the serial chain: its few MFMAs go before its SIMD neighbour's
    __shared__ double rd[NB + 2];                       // L's diagonal (the leaves hold their inverse in place); [NB] = breakdown flag
    double *misc = rd + NB;                             // (16-byte multiple keeps the dynamic base aligned)
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    double *A = S + ((size_t)kb * NB) * ld + (size_t)kb * NB;
    if (t == 0) misc[0] = 0.0;
    {   // the block, 128 KB, as 16-byte loads ALL in flight before the first is used (round 2 loaded element by element
        // with 256 threads: ~25 us of this kernel's 83 were this loop)
        constexpr int PER = NB * (NB / 2) / (64 * CDW);      // 16-byte chunks per thread
        f64x2 v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = t + 64 * CDW * q, r = i / (NB / 2), c2 = i % (NB / 2);
            v[q] = *reinterpret_cast<const f64x2 *>(A + (size_t)r * ld + 2 * c2);
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = t + 64 * CDW * q, r = i / (NB / 2), c = 2 * (i % (NB / 2));
            L[r * DL + c] = c <= r ? v[q][0] : 0.0;
            L[r * DL + c + 1] = c + 1 <= r ? v[q][1] : 0.0;
        }
    }
    __syncthreads();
    STAMP(0);
    // 1a. 16x16 leaf by wave 0 alone, on the matrix pipe.  The block lives in ONE accumulator tile S (all 256 entries,
    //     kept symmetric) and is eliminated four columns at a time:
    //       - the 4x4 diagonal block is pulled into wave-uniform values (v_readlane) and factored AND inverted there,
    //         ~60 scalar-shaped f64 operations; M = (its factor)^-1, padded with zeros, becomes an MFMA A operand;
    //       - M x S[p] gives the four columns of L for ALL sixteen rows at once, already in operand layout
    //         (lane (row, k)), and S -= P P^T is one more MFMA;
    //       - the identity, carried along as extra rows (T2 = its transpose), undergoes the same column operations and
    //         ends as L^-T: the leaf inverse costs two more MFMAs per step instead of a 136-term substitution.
    //     Entries of S and T2 left of the active columns turn into rounding residue and are never read for a stored
    //     value.  (Round 2's leaf kept one row per lane and did all of this on the vector ALU: ~1200 instructions and
    //     8.6k cycles per leaf, the longest serial stretch of the factorisation's critical chain.)
    double *out = Linv + (size_t)kb * NB * NB;
    unsigned mk[10];      // slot of M[i][k] (i >= k, row-major over the lower triangle) -> all-ones in the lane (i, k) that holds it
    {
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0, slot = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k <= i; ++k, ++slot) mk[slot] = (c == i && g == k) ? 0xFFFFFFFFu : 0u;
    }
    auto leaf_factor = [&](int c0) {
            double *Lb = L + c0 * DL + c0;
            const int c = lane & 15, g = lane >> 4;
            f64x4 Sm, T2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = g + 4 * r;
                Sm[r] = Lb[(i > c ? i : c) * DL + (i > c ? c : i)];
                T2[r] = i == c ? 1.0 : 0.0;
            }
            double Lc[4], Yc[4];
            double last = 0.0;
            const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                auto pick = [&](int k, int j) {      // S[4p+k][4p+j] as a uniform value
                    const int src = 4 * p + j + 16 * k;
                    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(Sm[p]), src),
                                            __builtin_amdgcn_readlane(__double2loint(Sm[p]), src));
                };
                // 1/sqrt(d): v_rsq_f64 is good to 2^-24 (measured over 2^26 arguments), one cubic step
                // y (1 + e/2 + 3 e^2/8), e = 1 - d y^2, leaves |1 - d y^2| <= 2.8e-16 in four dependent operations.
                // A pivot that is not positive and finite makes y a NaN, and the NaN reaches every later pivot.
                auto rsq3 = [&](double d) {
                    const double y = __builtin_amdgcn_rsq(d);
                    const double e = fma(-(d * y), y, 1.0);
                    return fma(y * e, fma(0.375, e, 0.5), y);
                };
                // the A operand of the solve, lane (i = c, k = g) = M[i][k], assembled by mask as each value appears
                // (selects here turn into divergent branches around the scalar chain)
                unsigned mlo = 0, mhi = 0;
                auto put = [&](double v, int slot) { mlo |= (unsigned)__double2loint(v) & mk[slot]; mhi |= (unsigned)__double2hiint(v) & mk[slot]; };
                double d00 = pick(0, 0), d10 = pick(1, 0), d20 = pick(2, 0), d30 = pick(3, 0), d11 = pick(1, 1),
                       d21 = pick(2, 1), d31 = pick(3, 1), d22 = pick(2, 2), d32 = pick(3, 2), d33 = pick(3, 3);
                const double i0 = rsq3(d00);
                put(i0, 0);
                const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
                d11 = fma(-l10, l10, d11); d21 = fma(-l20, l10, d21); d31 = fma(-l30, l10, d31);
                d22 = fma(-l20, l20, d22); d32 = fma(-l30, l20, d32); d33 = fma(-l30, l30, d33);
                const double i1 = rsq3(d11);
                put(i1, 2);
                const double m10 = -i1 * (l10 * i0);
                put(m10, 1);
                const double l21 = d21 * i1, l31 = d31 * i1;
                d22 = fma(-l21, l21, d22); d32 = fma(-l31, l21, d32); d33 = fma(-l31, l31, d33);
                const double i2 = rsq3(d22);
                put(i2, 5);
                const double m21 = -i2 * (l21 * i1), m20 = -i2 * fma(l21, m10, l20 * i0);
                put(m21, 4); put(m20, 3);
                const double l32 = d32 * i2;
                d33 = fma(-l32, l32, d33);
                const double i3 = rsq3(d33);
                const double m32 = -i3 * (l32 * i2), m31 = -i3 * fma(l32, m21, l31 * i1);
                const double m30 = -i3 * fma(l32, m20, fma(l31, m10, l30 * i0));
                put(i3, 9); put(m32, 8); put(m31, 7); put(m30, 6);
                last = i3;
                const double mop = __hiloint2double((int)mhi, (int)mlo);
                const f64x4 X = __builtin_amdgcn_mfma_f64_16x16x4f64(mop, Sm[p], zero4, 0, 0, 0);   // [0]: lane (j, g) = L[j][4p+g]
                if (p < 3) Sm = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[0], X[0], Sm, 0, 0, 0);
                const f64x4 Y = __builtin_amdgcn_mfma_f64_16x16x4f64(mop, T2[p], zero4, 0, 0, 0);   // [0]: lane (e, g) = L^-T[e][4p+g]
                if (p < 3) T2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[0], Y[0], T2, 0, 0, 0);
                Lc[p] = X[0]; Yc[p] = Y[0];
            }
            const bool ok = isfinite(last);
            if (!ok && lane == 0) misc[0] = 1.0;
            // lane (c, g), m = 4p + g: the INVERSE goes in place (entry [m][c], m >= c) -- every later reader of this block
            // (the solves below, the doubling steps, the output) wants the inverse; the factor itself is only ever stored
            // on request and waits transposed above the diagonal ([m][c] = L[c][m], m < c), its diagonal in rd
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int m = 4 * p + g;
                Lb[m * DL + c] = m >= c ? Yc[p] : Lc[p];
                if (m == c) rd[c0 + c] = Lc[p];
            }
            };
    const int NA = LB * ((min(max(nact, 1), NB) + LB - 1) / LB);      // rows of the leaves that hold active rows (a multiple of 16, at least one leaf)
    if (w == 0) leaf_factor(0);
    // the padding: unit diagonal of the factor (rd: what store_L reads) and of the inverse; everything else there is zero already --
    // the block was loaded with zeros above the diagonal, the padding rows are zero left of it, and the inverse's buffer starts as zeros
    for (int i = NA + t; i < NB; i += 64 * CDW) { rd[i] = 1.0; out[(size_t)i * NB + i] = 1.0; }
    __syncthreads();
    STAMP(1);
    f64x4 tcur = {0.0, 0.0, 0.0, 0.0};       // waves 1..7: T_j of the inverse's next block row (2., below)
    for (int c0 = 0; c0 < NA; c0 += LB) {
        if (misc[0] != 0.0) {
            if (t == 0) flag_raise(flag, 1);
            return false;
        }
        const int r0 = c0 + LB, kk = c0 / LB;
        if (r0 >= NA) break;
        // 1b. rows below: X = A * D^-T as 16x16 MFMA tiles (one wave per tile), in place:
        //     X[r][c] = sum_{k<=c} A[r][k] * Dinv[c][k]
        {
            const int leaf = c0 / LB, ntile = (NA - r0) / 16;
            for (int tile = w; tile < ntile; tile += CDW) {
                double *At = L + (r0 + 16 * tile) * DL + c0;
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < LB; kk += 4) {
                    const int k = kk + (lane >> 4), c = lane & 15;
                    const double a = At[(lane & 15) * DL + k];
                    const double b = k <= c ? L[(LB * leaf + c) * DL + LB * leaf + k] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) At[((lane >> 4) + 4 * reg) * DL + (lane & 15)] = acc[reg];
            }
        }
        __syncthreads();
        STAMP(32 + 2 * (c0 / LB));
        // 1c. trailing square -= panel panel^T, lower 16x16 tiles on MFMA: the accumulator starts
        //     as the C tile and the A operand is negated.  LOOKAHEAD: wave 0 takes tile (0,0) -- the next
        //     leaf's diagonal block -- and goes straight on to factor and invert that leaf while waves
        //     1..3 update the rest of the square, so the serial leaf work hides behind the MFMA work.
        {
            const int nt = (NA - r0) / 16, ntile = nt * (nt + 1) / 2;
            for (int tile = w == 0 ? 0 : w; tile < ntile; tile += (w == 0 ? ntile : CDW - 1)) {
                int tr = (int)((sqrtf(8.f * tile + 1.f) - 1.f) * 0.5f);
                while ((tr + 1) * (tr + 2) / 2 <= tile) ++tr;
                while (tr * (tr + 1) / 2 > tile) --tr;
                const int tcn = tile - tr * (tr + 1) / 2;
                double *Ct = L + (r0 + 16 * tr) * DL + r0 + 16 * tcn;
                const double *Pa = L + (r0 + 16 * tr) * DL + c0, *Pb = L + (r0 + 16 * tcn) * DL + c0;
                f64x4 acc;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[reg] = Ct[((lane >> 4) + 4 * reg) * DL + (lane & 15)];
#pragma unroll
                for (int kk = 0; kk < LB; kk += 4) {
                    const int k = kk + (lane >> 4);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pa[(lane & 15) * DL + k], Pb[(lane & 15) * DL + k], acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Ct[((lane >> 4) + 4 * reg) * DL + (lane & 15)] = acc[reg];
            }
            if (w == 0) leaf_factor(r0);
            else if (w - 1 <= kk) {
                // 2. the inverse grows a block row per leaf, behind the leaf work of wave 0:  Linv[k][j] = -Dinv_k T_j  with
                //    T_j = sum_{m = j .. k-1} L[k][m] Linv[m][j].  Wave j + 1 owns tile column j for good: it finishes row kk
                //    (T_j came with it from the last step, in registers: accumulator layout IS the B-operand layout),
                //    keeps the block in the mirror position above the diagonal for its own later use, sends it to HBM, and
                //    builds T_j of row kk + 1 -- whose leaf wave 0 is factoring right now.  Every block it reads is its own
                //    or a finished leaf's, so there is no synchronisation beyond the barriers of the factorisation.
                const int j = w - 1, ci = lane & 15, kq = lane >> 4;
                f64x4 R = {0.0, 0.0, 0.0, 0.0};
                if (j < kk) {
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) {
                        const int k = 4 * sx + kq;
                        const double a = k <= ci ? -L[(c0 + ci) * DL + c0 + k] : 0.0;
                        R = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tcur[sx], R, 0, 0, 0);
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        L[(LB * j + kq + 4 * reg) * DL + c0 + ci] = R[reg];
                        out[(size_t)(c0 + kq + 4 * reg) * NB + LB * j + ci] = R[reg];
                    }
                }
                f64x4 T = {0.0, 0.0, 0.0, 0.0};
                const double *Ar = L + (r0 + ci) * DL;          // row of L[kk + 1][.] this lane feeds as A operand
                {   // m = j: the leaf inverse on the diagonal (lower triangular; above it sits the factor, transposed).  The wave
                    // that starts a tile column (j == kk) is the first to read that leaf's inverse: it also sends it to HBM.
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) {
                        const int k = 4 * sx + kq;
                        const double b = ci <= k ? L[(LB * j + k) * DL + LB * j + ci] : 0.0;
                        if (j == kk && ci <= k) out[(size_t)(c0 + k) * NB + c0 + ci] = b;
                        T = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[LB * j + k], b, T, 0, 0, 0);
                    }
                }
                if (j + 1 < kk) {    // the blocks between, operands of step m + 1 requested before the MFMAs of step m
                    double an[4], bn[4];
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) { an[sx] = Ar[LB * (j + 1) + 4 * sx + kq]; bn[sx] = L[(LB * j + 4 * sx + kq) * DL + LB * (j + 1) + ci]; }
                    for (int m = j + 1; m < kk; ++m) {
                        double ac[4], bc[4];
#pragma unroll
                        for (int sx = 0; sx < 4; ++sx) { ac[sx] = an[sx]; bc[sx] = bn[sx]; }
                        const int mn = m + 1 < kk ? m + 1 : m;
#pragma unroll
                        for (int sx = 0; sx < 4; ++sx) { an[sx] = Ar[LB * mn + 4 * sx + kq]; bn[sx] = L[(LB * j + 4 * sx + kq) * DL + LB * mn + ci]; }
#pragma unroll
                        for (int sx = 0; sx < 4; ++sx) T = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[sx], bc[sx], T, 0, 0, 0);
                    }
                }
                if (j < kk) {
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) T = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[c0 + 4 * sx + kq], R[sx], T, 0, 0, 0);
                }
                tcur = T;
            }
        }
        __syncthreads();
        STAMP(33 + 2 * (c0 / LB));
    }
    STAMP(13);
    // The factor of the diagonal tile itself is read by nobody (panels and triangular solves use its
    // inverse) except for the right-hand-side row inside the LAST tile (k_ba_y_from_row): stored on request.
    if (store_L)
        for (int i = t; i < NB * NB; i += 64 * CDW) {
            const int r = i / NB, c = i % NB;
            if (c <= r) A[(size_t)r * ld + c] = r / LB != c / LB ? L[r * DL + c] : r == c ? rd[r] : L[c * DL + r];
        }
    STAMP(14);
    // the last ACTIVE block row of the inverse: its leaf was the last thing the loop did (tile columns left of it only: with one
    // active leaf there are none)
    if (w >= 1) {
        const int j = w - 1, c0 = NA - LB, ci = lane & 15, kq = lane >> 4;
        f64x4 R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const int k = 4 * sx + kq;
            const double a = k <= ci ? -L[(c0 + ci) * DL + c0 + k] : 0.0;
            R = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tcur[sx], R, 0, 0, 0);
        }
        if (LB * j < c0)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) out[(size_t)(c0 + kq + 4 * reg) * NB + LB * j + ci] = R[reg];
        if (w == 1) {
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                const int k = 4 * sx + kq;
                if (ci <= k) out[(size_t)(c0 + k) * NB + c0 + ci] = L[(c0 + k) * DL + c0 + ci];
            }
        }
    }
    STAMP(16);
    STAMP(17);
    TL_MARK(tl, 2);
    return true;
}
// (x_out != nullptr, systems of ONE block -- the reference's own sizes up to a dozen cameras; round 5: the backward substitution of the
//  block rides in this launch, x = Linv' y with y = row `yrow` of the factor just stored.  The arithmetic is k_trsv_bwd_chain's for its
//  last block row -- four partial sums of 32, combined pairwise -- so the bits are the same; one launch less per LM iteration.)
__global__ __launch_bounds__(64 * CDW) void k_chol_diag(double *S, int ld, int kb, double *Linv, int *flag, int store_L, Gate g, int nact = NB, int tl = 0,
                                                        double *x_out = nullptr, int yrow = 0)
{
    __builtin_amdgcn_s_setprio(3);
    TL_MARK(tl, 0);
    gate_enter(g);
    TL_MARK(tl, 1);
    const bool ok = chol_diag_body(S, ld, kb, Linv, flag, store_L, nact, tl);
    if (!x_out || !ok) return;              // (uniform)
    static_assert(64 * CDW == 512, "the substitution's four groups of 128 threads");
    __shared__ double xk[NB], part[4][NB];
    __syncthreads();                         // the block's inverse and its last row are in memory for every thread of the workgroup
    const int t = threadIdx.x & 127, gq = threadIdx.x >> 7;
    const double *Lk = Linv + (size_t)kb * NB * NB;
    double li[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) li[m] = Lk[(size_t)(32 * gq + m) * NB + t];
    if (gq == 0) {
        const int idx = kb * NB + t;
        xk[t] = idx < yrow ? S[(size_t)yrow * ld + idx] : 0.0;
    }
    __syncthreads();
    double sum = 0.0;
#pragma unroll
    for (int m = 0; m < 32; ++m) sum += li[m] * xk[32 * gq + m];      // zeros above the diagonal
    part[gq][t] = sum;
    __syncthreads();
    if (gq == 0) x_out[(size_t)kb * NB + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}
// The diagonal blocks of a whole factorisation in ONE workgroup that stays resident (round 5).  As a launch per block the 512-thread,
// 132-KB workgroup had to find a CU every step -- and where the panel stream's latency kernels, unmasked since this round, fill the
// CUs that are kept free of bulk work, it waited ~50 us for one at every second step of the right-looking regime (device timeline).
// Here the workgroup keeps its CU: per block it waits for the counters the plan names (thread 0 polls, one agent-scope acquire,
// workgroup barrier -- the consumer recipe), factors the block exactly as k_chol_diag does, drains its stores, and publishes its
// ticket with an agent-scope release (a kernel boundary did that before).  A raised flag (breakdown: 1, a wait that timed out: 3)
// ends the loop; the host then does what it always did.
struct DiagItem { int kb, store_L, ticket, tl, nw, ctr[6], val[6]; };
__global__ __launch_bounds__(64 * CDW) void k_chol_diag_server(double *S, int ld, double *Linv, int *flag, const DiagItem *__restrict__ items, int n_items, int *ctr_base, int own_ctr,
                                                                int n_active)
{
    __builtin_amdgcn_s_setprio(3);
    __shared__ int s_stop;
    for (int it = 0; it < n_items; ++it) {
        const DiagItem item = items[it];
        TL_MARK(item.tl, 0);
        if (threadIdx.x == 0) {
            for (int i = 0; i < 6; ++i)
                if (i < item.nw) ring_wait(ctr_base + item.ctr[i], item.val[i], flag);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_stop = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        }
        __syncthreads();
        if (s_stop) return;
        TL_MARK(item.tl, 1);
        const int nact = min(NB, n_active - item.kb * NB);
        const bool ok = chol_diag_body(S, ld, item.kb, Linv, flag, item.store_L, nact, item.tl);
        if (!ok) return;                        // (uniform: every thread read the same LDS word behind a barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ctr_base + own_ctr, item.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Dense blocked Cholesky, GEMM side.  S holds the reduced system and its trailing updates; the
// panels  P[i,kb] = S[i,kb] Linv_kb^T  (the sub-diagonal tiles of L) are written to a separate
// matrix L, so no kernel ever overwrites an operand another workgroup is still reading.
//
// k_gemm_q<0> (panel)  and  k_gemm_q<1> (first trailing tile column: S[i,kb+1] -= L[i,kb] L[kb+1,kb]^T)
// are the serial chain the next diagonal block waits for, so they are built for latency and for
// running BESIDE the bulk update, whose two resident workgroups per CU hold 128 KB of LDS and
// ~420 of the 512 VGPRs of a SIMD: no LDS, no barrier, at most 96 VGPRs.  One wave owns one 16x16
// output tile (a 128x128 tile = 64 waves) and loads its operands straight into MFMA layout, K in
// two halves of 64: per half 8 + 8 sixteen-byte loads per lane (lane group fk takes k = 8 g + 2 fk
// and + 1 of every 8-wide group g, so a load instruction covers 16 rows x one 64-B line), then 16
// v_mfma_f64_16x16x4_f64.  The four workgroups that share a 32-row A strip carry the same
// (blockIdx & 7), i.e. run on the same XCD and share the strip in its L2.
// k_gemm_nt_pipe (the rest of the trailing update, S[i,j] -= L[i,kb] L[j,kb]^T for kb+1 < j <= i)
// runs beside them on its own stream and is built for throughput.
// first / m: the tiles kb + 1 + first .. kb + 1 + first + m - 1 of the tile column (the critical tile is first = 0, m = 1)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(96)))
void k_gemm_q(double *S, double *L, int ld, int kb, int first, int m, const double *Linv, Gate g, int dj = 1, int tl = 0)
{
    __builtin_amdgcn_s_setprio(3);      // these waves share SIMDs with the bulk update's: their few MFMAs and loads go first
    [[maybe_unused]] const int tl_id = tl;
    TL_MARK(tl_id, 0);
    gate_enter(g);
    TL_MARK(tl_id, 1);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int strip = 8 * (slot >> 2) + xcd, qj = slot & 3;      // strip: 32 rows of the tile column, qj: 32 output columns
    if (strip >= 4 * m) return;
    const int tj = MODE == 0 ? kb : kb + dj;      // MODE 1: the tile column that is updated, S(i, kb + dj) -= L(i, kb) L(kb + dj, kb)'
    const size_t row0 = (size_t)(kb + 1 + first) * NB + 32 * (size_t)strip;
    const double *A = (MODE == 0 ? S : L) + row0 * ld + (size_t)kb * NB;
    const double *B = MODE == 0 ? Linv + (size_t)kb * NB * NB + (size_t)(32 * qj) * NB
                                : L + ((size_t)tj * NB + 32 * qj) * ld + (size_t)kb * NB;
    const int ldb = MODE == 0 ? NB : ld;
    double *C = (MODE == 0 ? L : S) + row0 * ld + (size_t)tj * NB + 32 * qj;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = (w >> 1) * 16, wc = (w & 1) * 16;
    const int fr = lane & 15, fk = lane >> 4;
    const f64x2 *ap = reinterpret_cast<const f64x2 *>(A + (size_t)(wr + fr) * ld + 2 * fk);
    const f64x2 *bp = reinterpret_cast<const f64x2 *>(B + (size_t)(wc + fr) * ldb + 2 * fk);
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg.  The accumulators start as the
    // C tile and the A operand is negated, so C - A B^T comes straight out of the MFMA chain.
    f64x4 acc;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) acc[reg] = MODE == 0 ? 0.0 : C[(size_t)(wr + fk + 4 * reg) * ld + wc + fr];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f64x2 a[8], b[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) { a[g] = ap[4 * (8 * half + g)]; b[g] = bp[4 * (8 * half + g)]; }
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(MODE == 0 ? a[g][h] : -a[g][h], b[g][h], acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) C[(size_t)(wr + fk + 4 * reg) * ld + wc + fr] = acc[reg];
    TL_MARK(tl_id, 2);
}

// The same latency form for the two-level regime's head (round 5): a LIST of tiles (map entry: row << 16 | column), several panels
// per tile, the accumulators kept in registers across them --
//   MODE 1  S(i, j) -= sum_q L(i, kb + q) L(j, kb + q)'              q < npan: the next super-diagonal block, K = 128 g
//   MODE 2  L(i, kb + c) = sum_{m <= c} S(i, kb + m) W[c][m]'         c = the entry's column: the head rows' panel product
// As pipelined launches (one 128 x 128 tile per workgroup, K = 512: 64 stages) these two sat on the chain's critical path for
// ~70 us each alone and ~100 us beside the bulk update; here a tile is 32 workgroups of four 16 x 16 waves, operands straight
// from L2 into MFMA layout, no LDS, at most 96 registers -- they fit beside anything.  Workgroup b: tile b / 16, a 32 x 32 part of it.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(96)))
void k_gemm_qm(double *S, double *L, int ld, int kb, const unsigned *__restrict__ map, int npan, const double *SI, int ldsi, Gate g, int tl)
{
    static_assert(MODE == 1 || MODE == 2, "update / panel product");
    __builtin_amdgcn_s_setprio(3);
    TL_MARK(tl, 0);
    gate_enter(g);
    TL_MARK(tl, 1);
    const unsigned e = map[blockIdx.x >> 4];
    if (e == ~0u) return;
    const int ti = (int)(e >> 16), ecol = (int)(e & 0x3fffu);
    const int sub = blockIdx.x & 15, strip = sub >> 2, qj = sub & 3;      // strip: 32 rows of the tile, qj: 32 output columns
    const int tj = MODE == 1 ? ecol : kb + ecol;
    const int np = MODE == 1 ? npan : ecol + 1;
    const size_t row0 = (size_t)ti * NB + 32 * (size_t)strip;
    const double *A = (MODE == 1 ? L : S) + row0 * ld + (size_t)kb * NB;
    const double *B = MODE == 1 ? L + ((size_t)tj * NB + 32 * qj) * ld + (size_t)kb * NB : SI + ((size_t)ecol * NB + 32 * qj) * ldsi;
    const int ldb = MODE == 1 ? ld : ldsi;
    double *C = (MODE == 1 ? S : L) + row0 * ld + (size_t)tj * NB + 32 * qj;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = (w >> 1) * 16, wc = (w & 1) * 16;
    const int fr = lane & 15, fk = lane >> 4;
    const f64x2 *ap = reinterpret_cast<const f64x2 *>(A + (size_t)(wr + fr) * ld + 2 * fk);
    const f64x2 *bp = reinterpret_cast<const f64x2 *>(B + (size_t)(wc + fr) * ldb + 2 * fk);
    f64x4 acc;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) acc[reg] = MODE == 2 ? 0.0 : C[(size_t)(wr + fk + 4 * reg) * ld + wc + fr];
    for (int q = 0; q < np; ++q) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f64x2 a[8], b[8];
#pragma unroll
            for (int gq = 0; gq < 8; ++gq) { a[gq] = ap[64 * q + 4 * (8 * half + gq)]; b[gq] = bp[64 * q + 4 * (8 * half + gq)]; }
#pragma unroll
            for (int gq = 0; gq < 8; ++gq)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(MODE == 2 ? a[gq][h] : -a[gq][h], b[gq][h], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) C[(size_t)(wr + fk + 4 * reg) * ld + wc + fr] = acc[reg];
    TL_MARK(tl, 2);
}

// Two-level regime (round 5): block row `pos` of W = L_JJ^-1, the inverse of the factor of the super-diagonal block [p, p + g) --
// g x g tiles, lower block-triangular, row-major in SI (row stride ldsi = 128 g):
//     W[pos][pos] = Linv_c,     W[pos][m] = -Linv_c  sum_{r = m .. pos-1} L(c, p + r) W[r][m]     (c = p + pos, m < pos)
// from L W = I.  With it every row below the super-block is ONE product, L(i, J) = S(i, J) W' (k_gemm_nt_pipe, MODE 2), instead
// of g panel products with g (g - 1) / 2 column updates between them.  Workgroup b < 8 pos: tile m = b / 8, its 16-column strip
// b % 8 -- eight waves, wave w the strip's rows 16 w .. 16 w + 15: first the sum (operands straight from L2 into MFMA layout,
// both fed with the same k permutation), through LDS, then the product with Linv_c, whose rows 16 w .. end at column 16 w + 15.
// The last workgroup copies the diagonal tile.  A few microseconds behind the chain's next diagonal block; only the last row
// of a super-block is waited for.
__global__ __launch_bounds__(512) void k_sinv(const double *L, int ld, const double *Linv, double *SI, int ldsi, int p, int pos, Gate g, int tl)
{
    __builtin_amdgcn_s_setprio(3);
    TL_MARK(tl, 0);
    gate_enter(g);
    TL_MARK(tl, 1);
    const int c = p + pos, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const double *Lc = Linv + (size_t)c * NB * NB;
    if ((int)blockIdx.x == 8 * pos) {
        for (int i = t; i < NB * NB / 2; i += 512) {
            const int r = i / (NB / 2), c2 = i % (NB / 2);
            *reinterpret_cast<f64x2 *>(SI + ((size_t)pos * NB + r) * ldsi + (size_t)pos * NB + 2 * c2) = *reinterpret_cast<const f64x2 *>(Lc + (size_t)r * NB + 2 * c2);
        }
        TL_MARK(tl, 2);
        return;
    }
    const int m = blockIdx.x >> 3, strip = blockIdx.x & 7;
    __shared__ double Y[NB][17];
    const int fr = lane & 15, fk = lane >> 4;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int r = m; r < pos; ++r) {
        const double *A = L + ((size_t)c * NB + 16 * w + fr) * ld + (size_t)(p + r) * NB + 2 * fk;
        const double *B = SI + ((size_t)r * NB + 2 * fk) * ldsi + (size_t)m * NB + 16 * strip + fr;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f64x2 a[8];
            double b0[8], b1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k0 = 64 * half + 8 * q;
                a[q] = *reinterpret_cast<const f64x2 *>(A + k0);
                b0[q] = B[(size_t)k0 * ldsi];
                b1[q] = B[(size_t)(k0 + 1) * ldsi];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][0], b0[q], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][1], b1[q], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Y[16 * w + fk + 4 * reg][fr] = acc[reg];
    __syncthreads();
    f64x4 o = {0.0, 0.0, 0.0, 0.0};
    const double *Ar = Lc + (size_t)(16 * w + fr) * NB + 2 * fk;
    for (int k0 = 0; k0 < 16 * (w + 1); k0 += 8) {
        const f64x2 a = *reinterpret_cast<const f64x2 *>(Ar + k0);
        o = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[0], Y[k0 + 2 * fk][fr], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[1], Y[k0 + 2 * fk + 1][fr], o, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) SI[((size_t)pos * NB + 16 * w + fk + 4 * reg) * ldsi + (size_t)m * NB + 16 * strip + fr] = o[reg];
    TL_MARK(tl, 2);
}

// Bulk trailing update, LDS-DMA ring:  S[i,j] -= L[i, kb..] L[j, kb..]^T  over 8-wide k-stages.  Operands reach LDS by LDS-DMA only -- no staging registers --
// through a 4-stage ring, three stages (24 k) ahead of the MFMAs, behind counted vmcnt waits and one
// raw s_barrier per stage.  A stage holds, per operand, 128 rows x 8 doubles as eight 1-KiB
// row groups in PIECE-MAJOR order (slot = piece * 16 + row, 16 B per slot): each DMA lane picks
// the global 16 B that belongs in its linear LDS slot, so a ds_read_b128 of 16 rows x one piece
// is one whole 256-B bank row (conflict-free without padding) and yields the operands of two
// MFMA steps (lane group fk supplies k = 2 fk and 2 fk + 1).
#define GST 4
#define GSTAGE_BYTES (2 * 128 * 8 * 8)   // A + B, 16 KiB
// The bulk kernel (round 3; round 2's k_gemm_nt_ring, same tile, ring and operand layout, waited for LDS inside every stage and read its C tile up front).  What changed is
// where a wave waits.  Measured on the ring form (tools/gemm_nt_bench, tools/mfma_f64_peak): the bare
// v_mfma_f64_16x16x4_f64 loop sustains 77 TFLOP/s on this chip (64 cycles per MFMA at ~2.36 GHz: f64 is not
// clock-limited), the ring form's loop alone 58 (operands + barrier exposed once per 8-k stage) and a K = 128 pass 37:
// a third of a pass is the C tile -- 128 KB read into the accumulators BEFORE the first MFMA, 128 KB stored after the last.
//   * operand fragments are double-buffered in registers: the ds_read_b128s of stage s+1 are issued in front of the 32
//     MFMAs of stage s, so a wave never waits for LDS between two MFMA bursts (only for the stage barrier);
//   * the accumulators start at zero and the C tile is folded in ON THE WAY: the wave's sixteen 16x16 tiles of C are
//     requested two at a time at the even stages 0, 2, .. 14 and added to their accumulators one stage (~2500 cycles)
//     later -- no load of C is waited for, and what is left at the end is the store.  C moves through buffer
//     instructions (one descriptor in SGPRs, one per-lane offset, wave-uniform row / tile offsets as scalar offsets): with
//     128 accumulator and 64 operand registers per lane there is no room for per-tile 64-bit addresses.
// With the stage index known at compile time every wait count below is a literal and the loop has no branch.
#define PIPE_PRIO 0x200      // flag: raise the wave priority (launches on the panel stream: they share SIMDs with the bulk update)
// f(integral_constant<int, BASE + I>) for I = 0 .. : the stage loop with the stage number as a compile-time constant
template <int BASE, int... I, class F> __device__ __forceinline__ void pipe_for_seq(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, BASE + I>{}), ...);
}
// Round 5.  Map entries name ABSOLUTE tiles: row << 16 | class << 14 | column (chol_plan.h); a tile of class 1 / 2 is counted out in
// sig[0] / sig[1] when it is finished (the tiles the next steps read first lead the launch).
// NST = 16 / 32: every stage at compile time (K = 128 / 256).  NST = 0: the ROLLED form for any longer pass -- sixteen compile-time
// stages that carry the C tiles, four-stage trips with running operand pointers, four compile-time stages at the end; the stage
// count is a run-time value, nst_rt (a multiple of 4, at least 24): one instance serves K = 512 and K = 1024 and the ragged
// passes of MODE 2.
// MODE 0  S(i, j) -= L(i, kb ..) L(j, kb ..)'          Out = S, Ain = L
// MODE 1  L(i, kb) = S(i, kb) Linv_kb'                 Out = L, Ain = S, Bm = Linv (row stride 128); no C tile
// MODE 2  L(i, kb + c) = S(i, kb .. kb + c) W[c][.]'    Out = L, Ain = S, Bm = the super-block's inverse W (row stride ldb_arg),
//         c = the entry's column: a pass of 16 (c + 1) stages; no C tile (the two-level regime's panel product)
//         (column 0 as 24 stages, K = 192: the rolled form's shortest pass -- the 64 extra columns meet the zero block W[0][1])
template <int DBG, int NST, int MODE>
__device__ __forceinline__ void pipe_body(double *Out, const double *Ain, int ld, int kb, const unsigned e, int flags, int *sig, const double *Bm, int ldb_arg, int nst_rt, int tl)
{
    constexpr bool rolled = NST == 0;
    constexpr int NSTC = rolled ? 64 : NST;      // what the compile-time stages see: in the rolled form the first sixteen are far from the end and the last four know their distance to it
    static_assert(rolled || (NST % 4 == 0 && NST >= 16 && NST <= 32), "C tiles are folded in during stages 0 .. 15");
    static_assert(MODE != 2 || rolled, "the panel product of the two-level regime has ragged pass lengths");
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    if (flags & PIPE_PRIO) __builtin_amdgcn_s_setprio(2);
    [[maybe_unused]] const int tl_id = tl;
    TL_MARK(tl_id, 0);
    const int ti = (int)(e >> 16), ecol = (int)(e & 0x3fffu), cls = (int)((e >> 14) & 3u);
    const int tj = MODE == 0 ? ecol : MODE == 1 ? kb : kb + ecol;
    const int nst = __builtin_amdgcn_readfirstlane(rolled ? (MODE == 2 ? (ecol == 0 ? 24 : 16 * (ecol + 1)) : nst_rt) : NST);
    const double *A = Ain + ((size_t)ti * NB) * ld + (size_t)kb * NB;
    const double *B = MODE == 0 ? Ain + ((size_t)tj * NB) * ld + (size_t)kb * NB : MODE == 1 ? Bm + (size_t)kb * NB * NB : Bm + ((size_t)ecol * NB) * ldb_arg;
    const int ldb = MODE == 0 ? ld : MODE == 1 ? NB : ldb_arg;
    double *S = Out;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);      // wave-uniform, and the compiler should know: everything derived
                                                               // from it (ring slots, C descriptor, tile offsets) lives in SGPRs
    const int wr = (w >> 1) * 64, wc = (w & 1) * 64;
    const int fr = lane & 15, fk = lane >> 4;
    // the wave's 64 x 64 part of the C tile through a buffer descriptor: per-lane byte offset of its corner element,
    // everything else (tile row / column, register row) is wave-uniform and travels as the scalar offset
    double *Cw = S + ((size_t)ti * NB + wr) * ld + (size_t)tj * NB + wc;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(Cw, 0, (int)(64 * (size_t)ld * 8), 0x00020000);
    const int cvo = (int)(((size_t)fk * ld + fr) * 8);
    const int ld8 = ld * 8;        // one row of the system in bytes (the C part of a wave spans 64 rows: far below 2^31)
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const double *srcA = A + (size_t)(32 * w + fr) * ld + 2 * fk;      // (advanced by the rolled middle of a long pass)
    const double *srcB = B + (size_t)(32 * w + fr) * ldb + 2 * fk;
    // slot: ring slot of the stage (stage number mod GST); k0: its first column relative to where srcA / srcB point (the rolled
    // middle of a long pass advances the two pointers, so that the offset stays an immediate)
    auto issue = [&](int slot, int k0) {
        char *buf = gsm + slot * GSTAGE_BYTES + 2048 * w;
        if (DBG & 4) return;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA + (size_t)(16 * q) * ld + k0),
                                             (__attribute__((address_space(3))) void *)(buf + 1024 * q), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcB + (size_t)(16 * q) * ldb + k0),
                                             (__attribute__((address_space(3))) void *)(buf + 8192 + 1024 * q), 16, 0, 0);
        }
    };
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)gsm;
    const unsigned offA = base + (unsigned)((wr >> 4) * 1024 + fk * 256 + fr * 16);
    const unsigned offB = base + (unsigned)(8192 + (wc >> 4) * 1024 + fk * 256 + fr * 16);
    f64x2 ra[2][4], rb[2][4];
    // wait until at most n of this wave's vector-memory operations (LDS-DMA pieces and C loads, in issue order) are pending
    auto wait_vm = [&](int n) {
        switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // a count this list does not know: wait for everything
        }
    };
    // "=&v": an LDS read writes its destination when the data returns -- it must not share a register with an address
    auto read_stage = [&](int P, int slot) {      // slot: ring slot of the stage that is read (stage number mod GST)
        const unsigned so = (unsigned)(slot * GSTAGE_BYTES);
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                     "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:1024\n\tds_read_b128 %6, %9 offset:2048\n\tds_read_b128 %7, %9 offset:3072"
                     : "=&v"(ra[P][0]), "=&v"(ra[P][1]), "=&v"(ra[P][2]), "=&v"(ra[P][3]), "=&v"(rb[P][0]), "=&v"(rb[P][1]), "=&v"(rb[P][2]), "=&v"(rb[P][3])
                     : "v"(offA + so), "v"(offB + so) : "memory");
    };
    for (int s = 0; s < GST; ++s) issue(s, 8 * s);
    // Stage s sits in register buffer s & 1.  Per stage: make stage s + 1 visible (its DMA pieces have landed for every
    // wave) and refill the ring slot stage s has just left; fold C tile s in; request C tile s + 2; request the operands of
    // stage s + 1; the 32 MFMAs.  The sixteen 16x16 tiles of C are requested ONE at a time, tiles 0 and 1 behind the ring's
    // prologue and tile s + 2 at stage s, and folded in TWO stages after their request (round 3 requested two tiles at the
    // even stages and folded them one stage later: the same sixteen registers, half the time for the load -- and a stage
    // lasts ~1.9 us with two workgroups on the CU, which is what a read from HBM takes under load: the fold waited at every
    // odd stage).
    // Vector-memory operations of a wave in issue order:  D0 D1 D2 D3 C0 C1 | D4 C2 | D5 C3 | ... (D = 4 DMA pieces, C = 4 loads);
    // every wait below counts the operations YOUNGER than the one it needs, which may stay pending.
    typedef int v2i __attribute__((ext_vector_type(2)));
    v2i craw[2][4];
    constexpr int NCT = MODE == 0 ? 16 : 0;            // C tiles
    constexpr bool with_c = NCT > 0 && !(DBG & 1);
    auto c_req = [&](int tile) {
        // inline asm: a load the compiler issues itself it also waits for itself, with vmcnt(0) -- the counter is in-order and
        // it cannot tell the DMA pieces behind the load from the load -- which would drain the ring at every stage
        const int ci = tile >> 2, cj = tile & 3;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=&v"(craw[tile & 1][reg]) : "v"(cvo + 128 * cj), "s"(crs), "s"((16 * ci + 4 * reg) * ld8) : "memory");
    };
    auto fold = [&](int tile) {
        const int ci = tile >> 2, cj = tile & 3, B = tile & 1;
        asm volatile("" : "+v"(craw[B][0]), "+v"(craw[B][1]), "+v"(craw[B][2]), "+v"(craw[B][3]));
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            union { v2i r; double d; } u;
            u.r = craw[B][reg];
            acc[ci][cj][reg] -= u.d;          // the accumulators hold A B' - C: the sign turns at the store
        }
    };
    // C loads among the tiles lo .. hi (those that exist)
    auto n_c = [&](int lo, int hi) { int n = 0; for (int j = lo; j <= hi; ++j) n += (with_c && j >= 0 && j < NCT) ? 1 : 0; return n; };
    if (with_c) { c_req(0); c_req(1); }
    wait_vm(12 + 4 * n_c(0, 1));
    __builtin_amdgcn_s_barrier();
    read_stage(0, 0);
    // One stage.  SC >= 0: the stage number is a compile-time constant (the first sixteen stages, which carry the C tiles, and the
    // last four, whose waits shrink with the ring); SC < 0: a stage of the rolled middle of a long pass (K = 384, 512: round 4),
    // stage number s_rt at run time, parity PAR of its register buffer at compile time, every wait the steady-state literal.
    auto stage = [&](auto sc, auto pos, int s_rt) {
        constexpr int SC = decltype(sc)::value, POS = decltype(pos)::value, P = POS & 1;      // POS: stage number mod GST
        constexpr bool mid = SC < 0;
        const int s = mid ? s_rt : SC;
        // my reads of stage s (requested one stage ago) have returned: the fragments are in their registers and the ring
        // slot is free on my side.  The ONLY LDS wait of the step -- the reads of stage s + 1 requested below stay in flight
        // behind this step's MFMAs (a wait in front of the MFMAs would wait for them too: the counter is in-order)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[P][0]), "+v"(ra[P][1]), "+v"(ra[P][2]), "+v"(ra[P][3]), "+v"(rb[P][0]), "+v"(rb[P][1]), "+v"(rb[P][2]), "+v"(rb[P][3]) :: "memory");
        if (mid || SC + 1 < NSTC) {
            // needs D(s+1).  Younger: D(s+2), D(s+3); the C loads issued behind D(s+1): C0, C1 behind the prologue (all of
            // D0 .. D3 precede them), C(t+2) behind D(t+4) at stage t, i.e. C(s-1) .. C(s+1) for s >= 3
            if constexpr (mid) wait_vm(8);
            else {
                constexpr int c_younger = SC + 1 <= 3 ? (with_c ? (SC + 2 < NCT ? SC + 2 : NCT) : 0) : (with_c ? ((SC - 1 < NCT) + (SC < NCT) + (SC + 1 < NCT)) : 0);
                wait_vm(4 * ((SC + 2 < NSTC) + (SC + 3 < NSTC)) + 4 * c_younger);
            }
            __builtin_amdgcn_s_barrier();
            if constexpr (mid) issue(POS, 8 * POS);                 // into the slot of stage s (the pointers stand at the loop trip's first stage + GST)
            else if constexpr (SC + GST < NSTC) issue(POS, 8 * (SC + GST));
        }
        if constexpr (!mid && with_c && SC < NCT) {
            // needs C(s), requested two stages ago (tiles 0, 1: behind the prologue).  Younger: C(s+1), and every D issued
            // behind C(s): D(s+3) and D(s+4) for s >= 2, D4 and D5 for s = 1, D4 for s = 0 -- those that exist
            constexpr int d_younger = SC >= 2 ? (SC + 3 < NSTC) + (SC + GST < NSTC) : (SC == 1 ? (4 < NSTC) + (5 < NSTC) : (4 < NSTC));
            wait_vm(4 * d_younger + 4 * (SC + 1 < NCT ? 1 : 0));
            fold(SC);
            if (SC + 2 < NCT) c_req(SC + 2);
        }
        if (mid || SC + 1 < NSTC) read_stage(1 - P, (POS + 1) % GST);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[P][i][h], rb[P][j][h], acc[i][j], 0, 0, 0);
    };
    using ic_m1 = std::integral_constant<int, -1>;
    static_assert(GST == 4, "the rolled middle advances one ring turn per trip");
    if constexpr (!rolled) {
        // every stage at compile time (K = 128, 256)
        pipe_for_seq<0>(std::make_integer_sequence<int, NST>{}, [&](auto sc) { stage(sc, std::integral_constant<int, decltype(sc)::value % GST>{}, 0); });
    } else {
        pipe_for_seq<0>(std::make_integer_sequence<int, 16>{}, [&](auto sc) { stage(sc, std::integral_constant<int, decltype(sc)::value % GST>{}, 0); });
        srcA += 8 * (16 + GST); srcB += 8 * (16 + GST);                 // stage 16 issues stage 20
        for (int s4 = 16; s4 < nst - 4; s4 += 4) {
            stage(ic_m1{}, std::integral_constant<int, 0>{}, s4); stage(ic_m1{}, std::integral_constant<int, 1>{}, s4 + 1);
            stage(ic_m1{}, std::integral_constant<int, 2>{}, s4 + 2); stage(ic_m1{}, std::integral_constant<int, 3>{}, s4 + 3);
            srcA += 8 * GST; srcB += 8 * GST;
        }
        pipe_for_seq<NSTC - 4>(std::make_integer_sequence<int, 4>{}, [&](auto sc) { stage(sc, std::integral_constant<int, decltype(sc)::value % GST>{}, 0); });
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (!(DBG & 2) || acc[i][j][reg] == 1.2345e300) {
                    union { v2i r; double d; } u;
                    u.d = MODE != 0 ? acc[i][j][reg] : -acc[i][j][reg];                  // C - A B'  (MODE 1, 2: A B')
                    __builtin_amdgcn_raw_buffer_store_b64(u.r, crs, cvo + 128 * j, (16 * i + 4 * reg) * ld8, 0);
                }
    // The tiles the next steps read first lead the launch and are counted out one by one (class 1 / 2 of the map entry): nobody
    // waits for a whole bulk update except through stream order.
    if (sig && cls) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_fetch_add(sig + (cls - 1), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#ifdef RCN_DIAG
        if (g_tl && t == 0) atomicMax(&g_tl[3 * tl_id + 1], (unsigned long long)wall_clock64());      // a head tile finished
#endif
    }
    TL_MARK(tl_id, 2);
}
template <int DBG, int NST, int MODE = 0>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_pipe(double *Out, const double *Ain, int ld, int kb, const unsigned *__restrict__ map, int flags, int *sig,
                                                          const double *Bm = nullptr, int ldb_arg = 0, int nst_rt = 0, int tl = 0)
{
    const unsigned e = map[blockIdx.x];
    if (e == ~0u) return;
    pipe_body<DBG, NST, MODE>(Out, Ain, ld, kb, e, flags, sig, Bm, ldb_arg, nst_rt, tl);
}
// ONE launch, two kinds of tiles (round 5): the trailing update of super-step J (the first `split` workgroups: MODE 0, rolled) and,
// behind them, the panel product of super-step J + 1 for the rows below its head (MODE 2).  Measured in the device timeline: as a
// launch of its own on another stream that product ran 2-5 times longer beside the bulk update than alone, and the bulk update
// 15 % longer beside it (47 against 53-59 TFLOP/s); here its tiles start where the bulk update's last round leaves slots free --
// they fill the drain -- and the next bulk update follows in stream order, without a gate.
// What a tail tile needs was produced elsewhere: the panel columns it reads by THIS launch's class-2 tiles (they lead the launch, so
// they were dispatched long before a tail workgroup can become resident: the wait cannot hold up what it waits for), the
// super-block's inverse by the chain and its followers on the panel stream, which run a super-step ahead of the bulk stream.
// Thread 0 polls (relaxed, bounded: 2 s -> flag 3 -> the one-stream schedule), one agent-scope acquire, workgroup barrier.
__global__ __launch_bounds__(256, 2) void k_gemm_nt_pipe_tail(double *S, double *L, int ld, int kb0, int nst0, const unsigned *__restrict__ map0, int split, int flags0, int *sig, int tl0,
                                                               int kb2, const unsigned *__restrict__ map2, const double *SI, int ldsi, Gate g2, int tl2)
{
    if ((int)blockIdx.x < split) {
        const unsigned e = map0[blockIdx.x];
        if (e == ~0u) return;
        pipe_body<0, 0, 0>(S, L, ld, kb0, e, flags0, sig, nullptr, 0, nst0, tl0);
    } else {
        const unsigned e = map2[blockIdx.x - split];
        if (e == ~0u) return;
        if (threadIdx.x == 0) {
            for (int i = 0; i < 6; ++i)
                if (i < g2.nw) ring_wait(g2.c[i], g2.n[i], g2.flag);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
#ifdef RCN_DIAG
        if (g_tl && threadIdx.x == 0 && (int)blockIdx.x == split) atomicMax(&g_tl[3 * tl2 + 0], (unsigned long long)wall_clock64());
#endif
        pipe_body<0, 0, 2>(L, S, ld, kb2, e, 0, nullptr, SI, ldsi, 0, tl2);
    }
}

// forward substitution step kb: y_kb = Linv_kb b_kb ; b_i -= L[i,kb] y_kb for i > kb.
// every workgroup recomputes y_kb (16k fma) and updates one 128-row tile.
__global__ __launch_bounds__(128) void k_trsv_fwd(const double *S /* = L: sub-diagonal tiles */, int ld, int kb, const double *Linv, double *b, double *y)
{
    __shared__ double yk[NB], bk[NB];
    const int t = threadIdx.x, i = kb + blockIdx.x;
    bk[t] = b[(size_t)kb * NB + t];
    __syncthreads();
    const double *Li = Linv + (size_t)kb * NB * NB + (size_t)t * NB;
    double s = 0.0;
#pragma unroll 8
    for (int m = 0; m < NB; ++m) s += (m <= t ? Li[m] : 0.0) * bk[m];   // Linv is stored with zeros above the diagonal
    yk[t] = s;
    __syncthreads();
    if (i == kb) { y[(size_t)kb * NB + t] = s; return; }
    const double *row = S + ((size_t)i * NB + t) * ld + (size_t)kb * NB;
    double u = 0.0;
#pragma unroll 16
    for (int m = 0; m < NB; ++m) u += row[m] * yk[m];
    b[(size_t)i * NB + t] -= u;
}

// backward substitution step kb (descending): x_kb = Linv_kb^T y_kb ; y_j -= L[kb,j]^T x_kb for j < kb
// 512 threads per workgroup: four groups of 128 split the 128 rows of each dot product (the 79 launches of a cfg-5
// solve are a serial chain: 12.5 us each with one thread per column walking all 128 rows, ~1 ms per LM iteration)
__global__ __launch_bounds__(512) void k_trsv_bwd(const double *S /* = L: sub-diagonal tiles */, int ld, int kb, const double *Linv, double *y, double *x)
{
    __shared__ double xk[NB], yk[NB], part[4][NB];
    const int t = threadIdx.x & 127, g = threadIdx.x >> 7, j = blockIdx.x;  // j = 0..kb ; j == kb writes x
    if (g == 0) yk[t] = y[(size_t)kb * NB + t];
    __syncthreads();
    const double *Lk = Linv + (size_t)kb * NB * NB;
    double s = 0.0;
#pragma unroll 8
    for (int m = 32 * g; m < 32 * g + 32; ++m) s += Lk[(size_t)m * NB + t] * yk[m];   // zeros above the diagonal
    part[g][t] = s;
    __syncthreads();
    if (g == 0) xk[t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    __syncthreads();
    if (j == kb) { if (g == 0) x[(size_t)kb * NB + t] = xk[t]; return; }
    const double *blk = S + ((size_t)kb * NB) * ld + (size_t)j * NB;  // L[kb, j] tile, rows m, col t
    double u = 0.0;
#pragma unroll 8
    for (int m = 32 * g; m < 32 * g + 32; ++m) u += blk[(size_t)m * ld + t] * xk[m];
    __syncthreads();
    part[g][t] = u;
    __syncthreads();
    if (g == 0) y[(size_t)j * NB + t] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

// The same backward substitution as ONE launch (round 3): 79 dependent launches of ~8 us each were 0.6 ms of a cfg-5 iteration.
// Workgroup j owns block row j: it holds Linv_j in registers from the start, takes every x_i (i > j) as it appears, subtracts
// L(i, j)' x_i from its right-hand side (the tile L(i, j) is already in registers by then), then publishes x_j = Linv_j' y_j.
// The hand-off is the DATA itself: x is preset to a sentinel (all-ones bit pattern, a NaN no arithmetic produces), written with
// agent-coherent stores and polled element by element with agent-coherent loads -- one memory round trip per step instead of
// three (store acknowledged, flag stored, flag seen, data loaded), no L2-wide release / acquire.  The sums are grouped exactly as
// in k_trsv_bwd (four partial sums of 32, combined pairwise, block rows in descending order), so the bits are the same.
// Workgroup j waits only for workgroups dispatched BEFORE it (larger j = smaller block index, and dispatch is in order), so the
// launch cannot deadlock whatever share of it is resident; the host still keeps it to nblk <= half the CUs.  An element that
// does not come within 2 s raises *flag = 4 and the host repeats the substitution with the per-step kernels.
#define TRSV_SENTINEL 0xFFFFFFFFFFFFFFFFull
// (y == nullptr: y is row `yrow` of the factor -- the right-hand side went through the factorisation as the system's last row; its
//  sub-diagonal tiles live in Lm, the last diagonal tile in Sm, entries from yrow on are zero: what k_ba_y_from_row extracts)
__global__ __launch_bounds__(512) void k_trsv_bwd_chain(const double *Lm, int ld, int nblk, const double *Linv, const double *y, double *x, int *flag,
                                                        const double *Sm = nullptr, int yrow = 0)
{
    __shared__ double xk[NB], part[4][NB];
    const int t = threadIdx.x & 127, g = threadIdx.x >> 7;
    const int j = nblk - 1 - (int)blockIdx.x;             // the head of the chain is dispatched first
    double li[32], cur[32], nxt[32];
    {
        const double *Lk = Linv + (size_t)j * NB * NB;
#pragma unroll
        for (int m = 0; m < 32; ++m) li[m] = Lk[(size_t)(32 * g + m) * NB + t];
    }
    double yj = 0.0;
    if (g == 0) {
        const int idx = j * NB + t;
        yj = y ? y[idx] : (idx < yrow ? (j < nblk - 1 ? Lm : Sm)[(size_t)yrow * ld + idx] : 0.0);
    }
    auto fetch = [&](int i, double (&dst)[32]) {
        const double *blk = Lm + ((size_t)i * NB) * ld + (size_t)j * NB;      // L[i, j] tile: rows m, column t
#pragma unroll
        for (int m = 0; m < 32; ++m) dst[m] = blk[(size_t)(32 * g + m) * ld + t];
    };
    if (nblk - 1 > j) fetch(nblk - 1, nxt);
    unsigned long long *xb = reinterpret_cast<unsigned long long *>(x);
    for (int i = nblk - 1; i > j; --i) {
#pragma unroll
        for (int m = 0; m < 32; ++m) cur[m] = nxt[m];
        if (i - 1 > j) fetch(i - 1, nxt);
        if (g == 0) {
            unsigned long long v = __hip_atomic_load(xb + (size_t)i * NB + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == TRSV_SENTINEL) {
                const unsigned long long t0 = wall_clock64();
                do {
                    __builtin_amdgcn_s_sleep(2);
                    v = __hip_atomic_load(xb + (size_t)i * NB + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (wall_clock64() - t0 > 200000000ull || __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { flag_raise(flag, 4); break; }
                } while (v == TRSV_SENTINEL);
            }
            xk[t] = __longlong_as_double((long long)v);
        }
        __syncthreads();
        double u = 0.0;
#pragma unroll
        for (int m = 0; m < 32; ++m) u += cur[m] * xk[32 * g + m];
        part[g][t] = u;
        __syncthreads();
        if (g == 0) yj -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    }
    __syncthreads();
    if (g == 0) xk[t] = yj;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < 32; ++m) s += li[m] * xk[32 * g + m];      // zeros above the diagonal
    part[g][t] = s;
    __syncthreads();
    if (g == 0) {
        const double xv = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        __hip_atomic_store(xb + (size_t)j * NB + t, (unsigned long long)__double_as_longlong(xv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------
// K8: point back-substitution, scaled step = -y.  Two passes: per observation  t_o = W_o' y_c  out of the W plane (a
// wave's 64 rows of 240 B staged through LDS as one contiguous block), then per point  -Vinv (g_p - sum_o t_o)  over the
// point's observations in order.
__global__ __launch_bounds__(128) void k_ba_backsub_obs(BaDev d)
{
    __shared__ __attribute__((aligned(16))) double stage[2 * 64 * JLD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t row0 = (size_t)blockIdx.x * blockDim.x + 64 * w;
    const int nrows = (int)min((long long)64, (long long)d.no - (long long)row0);
    double *mine = stage + w * 64 * JLD;
    if (nrows > 0) rows_to_lds<WCH>(d.WY, row0, nrows, mine, lane);
    __syncthreads();
    if (lane >= nrows) return;
    const size_t o = row0 + lane;
    const int c = d.ocam[o], dc = d.cam_dim[c];
    const double *y = d.yc + d.cam_off[c];
    const f64x2 *Wr = reinterpret_cast<const f64x2 *>(mine + lane * JLD);
    double W[WROW];
#pragma unroll
    for (int k = 0; k < WCH; ++k) { const f64x2 v = Wr[k]; W[2 * k] = v[0]; W[2 * k + 1] = v[1]; }
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int a = 0; a < 10; ++a)
        if (a < dc) { const double ya = y[a]; t0 += W[a] * ya; t1 += W[10 + a] * ya; t2 += W[20 + a] * ya; }
    d.tobs[3 * o] = t0; d.tobs[3 * o + 1] = t1; d.tobs[3 * o + 2] = t2;
}
__global__ void k_ba_backsub(BaDev d)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < d.n) d.stc[j] = -d.yc[j];
    if (j >= d.np) return;
    double tt[3] = {d.gps[3 * (size_t)j], d.gps[3 * (size_t)j + 1], d.gps[3 * (size_t)j + 2]};
    for (int o = d.pt_off[j]; o < d.pt_off[j + 1]; ++o) {
        tt[0] -= d.tobs[3 * (size_t)o]; tt[1] -= d.tobs[3 * (size_t)o + 1]; tt[2] -= d.tobs[3 * (size_t)o + 2];
    }
    const double *Vi = d.Vinv + 9 * (size_t)j;
    for (int a = 0; a < 3; ++a) d.stp[3 * (size_t)j + a] = -(Vi[3 * a] * tt[0] + Vi[3 * a + 1] * tt[1] + Vi[3 * a + 2] * tt[2]);
}

// model: sum_o m (r + m/2), m = Js step  (partial per block); observation rows staged through LDS like k_ba_eval writes them
__global__ __launch_bounds__(128) void k_ba_model(BaDev d, double *partial, Fin fin)
{
    __shared__ double sh[4];
    __shared__ __attribute__((aligned(16))) double stage[2 * 64 * JLD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t row0 = (size_t)blockIdx.x * blockDim.x + 64 * w;
    const int nrows = (int)min((long long)64, (long long)d.no - (long long)row0);
    double *mine = stage + w * 64 * JLD;
    if (nrows > 0) rows_to_lds<JCH>(d.J, row0, nrows, mine, lane);
    __syncthreads();
    double acc = 0.0;
    if (lane < nrows) {
        const size_t o = row0 + lane;
        const int c = d.ocam[o], j = d.opt[o], off = d.cam_off[c], dc = d.cam_dim[c];
        const f64x2 *Jr = reinterpret_cast<const f64x2 *>(mine + lane * JLD);
        double row[JROW];
#pragma unroll
        for (int k = 0; k < JCH; ++k) { const f64x2 v = Jr[k]; row[2 * k] = v[0]; row[2 * k + 1] = v[1]; }
        double sc[10], sp[3];
#pragma unroll
        for (int k = 0; k < 10; ++k) sc[k] = k < dc ? d.sc[off + k] * d.stc[off + k] : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) sp[k] = d.sp[3 * (size_t)j + k] * d.stp[3 * (size_t)j + k];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double m = 0.0;
#pragma unroll
            for (int k = 0; k < 10; ++k) if (k < dc) m += row[10 * i + k] * sc[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) m += row[20 + 3 * i + k] * sp[k];
            acc += m * (row[26 + i] + 0.5 * m);
        }
    }
    const double tot[1] = {block_sum(acc, sh)};
    finish_sums<1>(tot, partial, 0, fin, sh);
}

// delta = alpha * step .* scale ; candidate = clamp(x + delta) ; partials of |dx|^2, |x|^2,
// gradient.delta and the finite check.  which: 0 -> also writes dlc/dlp (alpha = 1 first time)
__global__ __launch_bounds__(256) void k_ba_plus(BaDev d, double alpha, double *partial, int nblocks, Fin fin)
{
    __shared__ double sh[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double dn = 0.0, xn = 0.0, g0 = 0.0, bad = 0.0;
    if (i < d.nc) {   // one thread per camera: 12 ambient parameters
        const int c = i, off = d.cam_off[c], dc = d.cam_dim[c];
        double ps[6], in[6];
        // |x| over the reduced program only, as Ceres' minimizer (x_norm_): a constant block (camera 0's pose, the
        // intrinsics when they are all held) or one no residual uses is not part of the state; a block with a
        // subset manifold counts with all six coordinates
        const bool used = d.cam_obs_off[c + 1] > d.cam_obs_off[c];
        const bool pose_in = used && dc > 0 && d.cols[10 * c] < 6, intr_in = used && d.mode == 1;
        for (int k = 0; k < 6; ++k) {
            ps[k] = d.poses[6 * c + k]; in[k] = d.intr[6 * c + k];
            if (pose_in) xn += ps[k] * ps[k];
            if (intr_in) xn += in[k] * in[k];
        }
        double ps2[6], in2[6];
        for (int k = 0; k < 6; ++k) { ps2[k] = ps[k]; in2[k] = in[k]; }
        // (every input of the loop below is loaded BEFORE its first store -- round 5: with the loads inside the loop the compiler had to
        //  keep each behind the store to dlc of the step before (the arrays may alias for all it knows), ten dependent round trips to L2
        //  in a kernel of a few hundred threads: 14 of its 16 us on the reference's problem sizes)
        double stc_[10], sc_[10], gcr_[10];
        int col_[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const bool on = k < dc;
            stc_[k] = on ? d.stc[off + k] : 0.0;
            sc_[k] = on ? d.sc[off + k] : 0.0;
            gcr_[k] = on ? d.gcraw[10 * (size_t)c + k] : 0.0;
            col_[k] = on ? d.cols[10 * c + k] : 0;
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            if (k >= dc) break;
            const double dl = alpha * stc_[k] * sc_[k];
            d.dlc[off + k] = dl;
            if (!isfinite(dl)) bad = 1.0;
            const int col = col_[k];
            // unscaled gradient of this coordinate = gcraw
            g0 += gcr_[k] * dl;
            if (col < 6) ps2[col] += dl;
            else {
                in2[col - 6] += dl;
                if (d.mode == 1 && col - 6 < 2 && in2[col - 6] > d.ub) in2[col - 6] = d.ub;
            }
        }
        for (int k = 0; k < 6; ++k) {
            d.poses2[6 * c + k] = ps2[k]; d.intr2[6 * c + k] = in2[k];
            dn += (ps2[k] - ps[k]) * (ps2[k] - ps[k]) + (in2[k] - in[k]) * (in2[k] - in[k]);
        }
        cam_rot_entry(ps2, d.crot2 + CROT * (size_t)c);
    }
    if (i < d.np) {
        const bool used = d.pt_off[i + 1] > d.pt_off[i];
        double stp_[3], sp_[3], x_[3], gp_[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const size_t e = 3 * (size_t)i + k; stp_[k] = d.stp[e]; sp_[k] = d.sp[e]; x_[k] = d.pts[e]; gp_[k] = d.gpraw[e]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t e = 3 * (size_t)i + k;
            const double dl = alpha * stp_[k] * sp_[k];
            d.dlp[e] = dl;
            if (!isfinite(dl)) bad = 1.0;
            const double x = x_[k];
            g0 += gp_[k] * dl;
            d.pts2[e] = x + dl;
            dn += dl * dl;
            if (used) xn += x * x;
        }
    }
    double mine[4];
    mine[0] = block_sum(dn, sh); mine[1] = block_sum(xn, sh); mine[2] = block_sum(g0, sh); mine[3] = block_sum(bad, sh);
    finish_sums<4>(mine, partial, nblocks, fin, sh);
}

// projected gradient max-norm of the unscaled gradient (gcraw / gpraw); *out zeroed by the caller.
// max is order-independent and non-negative doubles order like their bit patterns, so one
// integer atomicMax per workgroup keeps the result deterministic.
// (with_diag: the clamped LM diagonal of k_ba_diag(what = 1) in the same launch -- after an accepted step both are due, from the same raw
//  blocks: one launch less per iteration, the same arithmetic)
__device__ __forceinline__ void ba_lm_diag_entry(const BaDev &d, int i, double lo, double hi)
{
    if (i < d.n) {
        int a = 0, b = d.nc - 1;
        while (a < b) { int m = (a + b + 1) >> 1; if (d.cam_off[m] <= i) a = m; else b = m - 1; }
        while (d.cam_dim[a] == 0 || d.cam_off[a] + d.cam_dim[a] <= i) ++a;
        const int k = i - d.cam_off[a];
        const double u = d.Uraw[100 * (size_t)a + 11 * k];
        d.dgc[i] = fmin(fmax(u * d.sc[i] * d.sc[i], lo), hi);
    }
    if (i < 3 * d.np) {
        const double v = d.Vraw[9 * (size_t)(i / 3) + 4 * (i % 3)];
        d.dgp[i] = fmin(fmax(v * d.sp[i] * d.sp[i], lo), hi);
    }
}
// (with_diag = 2, round 5: and the damped 3 x 3 blocks of k_ba_point_solve at trust-region radius 1 / inv_radius -- a landmark's three
//  diagonal entries and its block by ONE thread, so nothing crosses threads: the step that follows an accepted one loses a launch)
__global__ __launch_bounds__(256) void k_ba_gradmax(BaDev d, double *out, int with_diag = 0, double lo = 0.0, double hi = 0.0, double inv_radius = 0.0)
{
    __shared__ double sh[4];
    const int t0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (with_diag == 2) {
        for (int i = t0; i < d.n; i += stride) ba_lm_diag_entry(d, i, lo, hi);          // (i < n <= 3 np is not given: the camera part on its own)
        for (int j = t0; j < d.np; j += stride) {
            for (int a = 0; a < 3; ++a) {
                const int i = 3 * j + a;
                const double v = d.Vraw[9 * (size_t)j + 4 * a];
                d.dgp[i] = fmin(fmax(v * d.sp[i] * d.sp[i], lo), hi);
            }
            point_solve_body(d, j, inv_radius);
        }
    }
    else if (with_diag)
        for (int i = t0; i < d.n || i < 3 * d.np; i += stride) ba_lm_diag_entry(d, i, lo, hi);
    double m = 0.0;
    for (int i = t0; i < d.nc; i += stride)
        for (int k = 0; k < d.cam_dim[i]; ++k) {
            double g = d.gcraw[10 * (size_t)i + k];
            const int col = d.cols[10 * i + k];
            if (d.mode == 1 && (col == 6 || col == 7)) {
                const double x = d.intr[6 * i + col - 6];
                double xn = x - g;
                if (xn > d.ub) xn = d.ub;
                g = x - xn;
            }
            m = fmax(m, fabs(g));
        }
    for (int i = t0; i < 3 * d.np; i += stride) m = fmax(m, fabs(d.gpraw[i]));
    for (int o = 32; o; o >>= 1) m = fmax(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
        atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(m));
    }
}

// =========================================================================================
// host side
// =========================================================================================
// The factorisation's schedule for nblk blocks (chol_plan.h): built once per shape and parameter set, its tile maps uploaded once.
static int ensure_chol_plan(rcn_ctx *ctx, int nblk)
{
    chol::Params prm;
    prm.nblk = nblk; prm.tl_g = ctx->chol_tl_g; prm.tl_min = ctx->chol_tl_min; prm.pair = ctx->chol_group >= 2 ? 1 : 0;
    prm.pair_min = ctx->chol_pair_min; prm.pipe_min = ctx->chol_pipe_min; prm.pg_stream = ctx->chol_pg_stream; prm.fuse_tail = ctx->chol_fuse_tail; prm.head_small = ctx->chol_head_small; prm.tl_serial = ctx->chol_tl_serial; prm.window = ctx->chol_window; prm.diag_server = ctx->chol_diag_server; prm.bulk_behind = ctx->chol_bulk_behind; prm.carve_rows = ctx->chol_carve_rows;
    const chol::Params &o = ctx->chol_plan.prm_asked;
    if (ctx->chol_plan_valid && o.nblk == prm.nblk && o.tl_g == prm.tl_g && o.tl_min == prm.tl_min && o.pair == prm.pair && o.pair_min == prm.pair_min &&
        o.pipe_min == prm.pipe_min && o.pg_stream == prm.pg_stream && o.fuse_tail == prm.fuse_tail && o.head_small == prm.head_small && o.tl_serial == prm.tl_serial && o.window == prm.window && o.diag_server == prm.diag_server && o.bulk_behind == prm.bulk_behind && o.carve_rows == prm.carve_rows) return RCN_OK;
    ctx->chol_plan_valid = false;
    ctx->chol_plan = chol::make_plan(prm);
    ctx->chol_plan.prm_asked = prm;
    RCN_HIP(hipStreamSynchronize(ctx->stream));          // nobody may still read the old maps
    const size_t nm = std::max<size_t>(ctx->chol_plan.maps.size(), 1);
    RCN_HIP(ctx->bulk_map.reserve(nm * sizeof(unsigned)));
    if (!ctx->chol_plan.maps.empty()) RCN_HIP(hipMemcpy(ctx->bulk_map.p, ctx->chol_plan.maps.data(), ctx->chol_plan.maps.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    if (prm.pg_stream && !ctx->panel2_stream) {      // a fourth stream for the first super-step's panel product (the shipping plan has none)
        if (ctx->bulk_cu_mask.empty() || hipExtStreamCreateWithCUMask(&ctx->panel2_stream, (uint32_t)ctx->bulk_cu_mask.size(), ctx->bulk_cu_mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            RCN_HIP(hipStreamCreateWithFlags(&ctx->panel2_stream, hipStreamNonBlocking));
        }
    }
    if (prm.diag_server && !ctx->diag_stream) {      // (tools/ only: the product's plans have no resident workgroup)
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        const char *dsp = nullptr;
#ifdef RCN_DIAG
        dsp = std::getenv("RCN_DIAG_STREAM_PRIO");      // 0: normal priority, 2: a CU mask of all CUs
#endif
        if (dsp && std::atoi(dsp) == 0) hi = 0;
        if (dsp && std::atoi(dsp) == 2) {      // a stream with a CU mask of all CUs: a hardware queue of its own?
            const int ncu = ctx->prop.multiProcessorCount;
            std::vector<uint32_t> full((ncu + 31) / 32, 0xFFFFFFFFu);
            if (ncu % 32) full.back() = (1u << (ncu % 32)) - 1u;
            RCN_HIP(hipExtStreamCreateWithCUMask(&ctx->diag_stream, (uint32_t)full.size(), full.data()));
        } else
        RCN_HIP(hipStreamCreateWithPriority(&ctx->diag_stream, hipStreamNonBlocking, hi));
    }
    {   // the resident diagonal workgroup's list: block, whether the factor itself is stored (last block), ticket, waits
        std::vector<DiagItem> items;
        for (const chol::Op &op : ctx->chol_plan.ops) {
            if (op.stream != chol::ST_E || op.kind != chol::DIAG) continue;
            DiagItem it;
            memset(&it, 0, sizeof(it));
            it.kb = op.kb; it.store_L = op.kb == nblk - 1; it.ticket = op.ticket; it.tl = op.tl; it.nw = op.nw;
            for (int i = 0; i < op.nw; ++i) { it.ctr[i] = op.w[i].ctr; it.val[i] = op.w[i].val; }
            items.push_back(it);
        }
        RCN_HIP(ctx->diag_items.reserve(std::max<size_t>(items.size(), 1) * sizeof(DiagItem)));
        if (!items.empty()) RCN_HIP(hipMemcpy(ctx->diag_items.p, items.data(), items.size() * sizeof(DiagItem), hipMemcpyHostToDevice));
    }
    ctx->chol_plan_valid = true;
    return RCN_OK;
}

namespace {

struct Ws {   // growable device workspace out of ctx->ba_ws
    rcn_ctx *ctx;
    int next = 0;
    hipError_t err = hipSuccess;
    template <class T> T *get(size_t count)
    {
        if (next >= (int)(sizeof(ctx->ba_ws) / sizeof(ctx->ba_ws[0]))) { err = hipErrorOutOfMemory; return nullptr; }
        DevBuf &b = ctx->ba_ws[next++];
        hipError_t e = b.reserve(std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) err = e;
        return b.as<T>();
    }
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

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

int rcn_ba_factor_plan(int32_t n_blocks, const int32_t *params, int32_t *ops, int64_t ops_cap, uint32_t *maps, int64_t maps_cap, int64_t *n_ops, int64_t *n_maps)
{
    if (n_blocks < 1 || n_blocks > 16383 || !n_ops || !n_maps) return RCN_ERR_ARG;
    chol::Params prm;
    {
        rcn_ctx defaults;      // (never created on a device: only the schedule's parameters are read)
        prm.tl_g = defaults.chol_tl_g; prm.tl_min = defaults.chol_tl_min; prm.pair = defaults.chol_group >= 2; prm.pair_min = defaults.chol_pair_min; prm.pipe_min = defaults.chol_pipe_min;
        prm.pg_stream = defaults.chol_pg_stream; prm.fuse_tail = defaults.chol_fuse_tail; prm.head_small = defaults.chol_head_small; prm.tl_serial = defaults.chol_tl_serial; prm.window = defaults.chol_window; prm.diag_server = defaults.chol_diag_server; prm.bulk_behind = defaults.chol_bulk_behind; prm.carve_rows = defaults.chol_carve_rows;
    }
    prm.nblk = n_blocks;
    if (params) { prm.tl_g = params[0]; prm.tl_min = params[1]; prm.pair = params[2]; prm.pair_min = params[3]; prm.pipe_min = params[4]; prm.pg_stream = params[5]; prm.fuse_tail = params[6]; prm.head_small = params[7]; prm.tl_serial = params[8]; prm.window = params[9]; prm.diag_server = params[10]; prm.bulk_behind = params[11]; prm.carve_rows = params[12]; }
    if (prm.tl_g < 0 || prm.tl_g == 1 || prm.tl_g > 16 || prm.pipe_min < 1) return RCN_ERR_ARG;
    const chol::Plan pl = chol::make_plan(prm);
    *n_ops = (int64_t)pl.ops.size();
    *n_maps = (int64_t)pl.maps.size();
    if ((int64_t)pl.ops.size() > ops_cap || (int64_t)pl.maps.size() > maps_cap || (!ops && !pl.ops.empty()) || (!maps && !pl.maps.empty())) return RCN_ERR_ARG;
    for (size_t i = 0; i < pl.ops.size(); ++i) {
        const chol::Op &o = pl.ops[i];
        int32_t *w = ops + RCN_PLAN_OP_WORDS * i;
        const int32_t v[RCN_PLAN_OP_WORDS] = {o.kind, o.stream, o.ticket, o.kb, o.first, o.m, o.dj, o.nst, o.map_off, o.map_n, o.g, o.pos, o.nw,
                                              o.w[0].ctr, o.w[0].val, o.w[1].ctr, o.w[1].val, o.w[2].ctr, o.w[2].val, o.w[3].ctr, o.w[3].val, o.w[4].ctr, o.w[4].val, o.w[5].ctr, o.w[5].val,
                                              o.tl, o.awaited, o.fuse_with, o.small};
        memcpy(w, v, sizeof(v));
    }
    if (!pl.maps.empty()) memcpy(maps, pl.maps.data(), pl.maps.size() * sizeof(uint32_t));
    return RCN_OK;
}

int rcn_ba_solve(rcn_ctx *ctx, const rcn_ba_problem *pb, const rcn_ba_options *opt, rcn_ba_summary *sum)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return rcn_int_ba_solve(ctx, pb, opt, sum, nullptr);
}

}  // extern "C"

// The solve proper (ctx->mu held).  res == nullptr: everything comes from / goes back to the host arrays of the
// problem.  res != nullptr (rcn_ba_session, ba_session.hip): the points and the observation arrays are already in HBM
// and stay there -- pb->points may be NULL, pb->obs_* are the session's host mirror (structure only) -- and the pair
// lists of the Schur build are reused when the token says the graph has not changed since they were built.
int rcn_int_ba_solve(rcn_ctx *ctx, const rcn_ba_problem *pb, const rcn_ba_options *opt, rcn_ba_summary *sum, const BaResident *res)
{
    if (!pb || !opt || !sum || pb->n_cams <= 0 || pb->n_points < 0 || pb->n_obs < 0 || !pb->poses ||
        !pb->intrinsics || (pb->n_points > 0 && !pb->points && !res) ||
        (pb->n_obs > 0 && (!pb->obs_uv || !pb->obs_cam || !pb->obs_pt))) {
        ctx->set_error("rcn_ba_solve: bad argument");
        return RCN_ERR_ARG;
    }
    memset(sum, 0, sizeof(*sum));
    const int nc = pb->n_cams, np = pb->n_points, no = pb->n_obs;
    RCN_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    // ---- host-side structure (BundleAdjuster.cpp:99-129 constraints -> tangent columns)
    std::vector<int> pt_off(np + 1, 0), cam_off(nc + 1, 0), cam_dim(nc, 0), cols(10 * (size_t)nc, 0),
        cam_obs_off(nc + 1, 0), cam_obs(std::max(no, 1));
    for (int o = 0; o < no; ++o) {
        const int j = pb->obs_pt[o], c = pb->obs_cam[o];
        if (j < 0 || j >= np || c < 0 || c >= nc || (o && j < pb->obs_pt[o - 1])) {
            ctx->set_error("rcn_ba_solve: observations must be landmark-major with valid indices");
            return RCN_ERR_ARG;
        }
        pt_off[j + 1]++;
        cam_obs_off[c + 1]++;
    }
    for (int j = 0; j < np; ++j) pt_off[j + 1] += pt_off[j];
    for (int c = 0; c < nc; ++c) cam_obs_off[c + 1] += cam_obs_off[c];
    {
        std::vector<int> fill(nc, 0);
        for (int o = 0; o < no; ++o) { const int c = pb->obs_cam[o]; cam_obs[cam_obs_off[c] + fill[c]++] = o; }
    }
    int n = 0, kmax = 0;
    for (int j = 0; j < np; ++j) kmax = std::max(kmax, pt_off[j + 1] - pt_off[j]);
    for (int c = 0; c < nc; ++c) {
        int dcm = 0;
        if (!(c == 0 && opt->fix_cam0_pose)) {
            const int npose = (c == 1 && opt->fix_cam1_translation) ? 3 : 6;
            for (int k = 0; k < npose; ++k) cols[10 * c + dcm++] = k;
        }
        if (opt->intrinsics_mode == 1) { cols[10 * c + dcm++] = 6; cols[10 * c + dcm++] = 7; cols[10 * c + dcm++] = 10; cols[10 * c + dcm++] = 11; }
        cam_off[c] = n; cam_dim[c] = dcm; n += dcm;
    }
    cam_off[nc] = n;
    const int npad = std::max(NB, (n + NB - 1) / NB * NB), nblk = npad / NB;
    sum->reduced_dim = n;

    // ---- device workspace
    Ws ws{ctx};
    BaDev d;
    memset(&d, 0, sizeof(d));
    d.nc = nc; d.np = np; d.no = no; d.n = n; d.npad = npad; d.mode = opt->intrinsics_mode; d.ub = opt->focal_upper_bound;
    d.poses = ws.get<double>(6 * (size_t)nc); d.intr = ws.get<double>(6 * (size_t)nc); d.pts = ws.get<double>(3 * (size_t)np);
    d.poses2 = ws.get<double>(6 * (size_t)nc); d.intr2 = ws.get<double>(6 * (size_t)nc); d.pts2 = ws.get<double>(3 * (size_t)np);
    double *uv = ws.get<double>(2 * (size_t)no);
    int *ints = ws.get<int>((size_t)no * 3 + np + 1 + 2 * (nc + 1) + nc + 10 * (size_t)nc + 24);      // + flag words [0..7], reduction tickets [8..11], the factorisation's counters [12..23]
    int *p_ocam = ints, *p_opt = p_ocam + no, *p_camobs = p_opt + no, *p_ptoff = p_camobs + no,
        *p_camobsoff = p_ptoff + np + 1, *p_camoff = p_camobsoff + nc + 1, *p_camdim = p_camoff + nc + 1,
        *p_cols = p_camdim + nc, *p_flag = p_cols + 10 * (size_t)nc;
    d.J = ws.get<double>(JROW * (size_t)no + 2); d.tobs = ws.get<double>(3 * (size_t)no);
    d.crot = ws.get<double>(CROT * (size_t)nc);
    d.crot2 = ws.get<double>(CROT * (size_t)nc);
    d.tickets = ws.get<unsigned>(2 * (size_t)std::max(nc, 1));
    d.Uraw = ws.get<double>(100 * (size_t)nc); d.gcraw = ws.get<double>(10 * (size_t)nc);
    d.Vraw = ws.get<double>(9 * (size_t)np); d.gpraw = ws.get<double>(3 * (size_t)np);
    const size_t nvec = (size_t)npad + 3 * (size_t)np + 16;
    double *vecs = ws.get<double>(12 * nvec);
    d.sc = vecs; d.sp = vecs + nvec; d.dgc = vecs + 2 * nvec; d.dgp = vecs + 3 * nvec;
    d.rhs = vecs + 4 * nvec; d.yc = vecs + 5 * nvec; d.stc = vecs + 6 * nvec; d.stp = vecs + 7 * nvec;
    d.dlc = vecs + 8 * nvec; d.dlp = vecs + 9 * nvec; d.gps = vecs + 10 * nvec;
    d.Vinv = ws.get<double>(9 * (size_t)np);
    d.S = ws.get<double>((size_t)npad * npad);
    d.L = ws.get<double>((size_t)npad * npad);
    d.Linv = ws.get<double>((size_t)nblk * NB * NB);
    const size_t si_elems = (size_t)(NB * std::max(ctx->chol_tl_g, 1)) * (NB * std::max(ctx->chol_tl_g, 1));
    double *SI = ws.get<double>(2 * si_elems);      // a super-block's inverse (two-level regime of the factorisation), two buffers alternating by super-step
    double *Sb = ws.get<double>(100 * (size_t)nc * nc);
    // gather lists of the Schur build (RCN_BA_SCHUR_ATOMICS=1 falls back to the atomic form)
    const bool gather = !ctx->ba_atomics;
    d.use_wy = 1;
    size_t npairs_lower = 0;
    for (int j = 0; j < np; ++j) { const size_t k = pt_off[j + 1] - pt_off[j]; npairs_lower += k * k; }   // upper bound
    const int nkeys = nc * nc;
    int *pk = ws.get<int>(gather ? 3 * (size_t)nkeys + 4 + (nkeys + 1023) / 1024 : 4);
    int *pk_cnt = pk, *pk_off = pk + nkeys + 1, *pk_fill = pk + 2 * nkeys + 2, *pk_sums = pk + 3 * (size_t)nkeys + 4;
    unsigned long long *pk_list = ws.get<unsigned long long>(gather ? std::max<size_t>(npairs_lower, 1) : 1);
    d.WY = ws.get<double>(2 * WROW * (size_t)std::max(no, 1));        // W and Y planes: the Schur gathers and the back-substitution read them
    // few cameras: split every camera over several workgroups so that the per-camera kernels fill the chip
    // (round 5: and only as far as a camera's observations call for it -- a workgroup of the camera kernels takes 256 observations per trip
    //  (k_ba_cam_raw) / 1024 (k_ba_schur_diag_mfma); splitting the reference's own problem sizes, a few hundred observations per camera,
    //  over 16 workgroups bought nothing and cost the ticket hand-off: 14 -> ~7 us per launch at 6 cameras)
    int max_cam_obs = 0;
    for (int c = 0; c < nc; ++c) max_cam_obs = std::max(max_cam_obs, cam_obs_off[c + 1] - cam_obs_off[c]);
    const int csplit = nc >= 128 ? 1 : std::max(1, std::min(std::min(16, 512 / std::max(nc, 1)), (max_cam_obs + 767) / 768));
    d.csplit = ws.get<double>((size_t)std::max(nc, 1) * csplit * 256);
    const int eb = (no + 255) / 256, ebj = (no + 127) / 128, pbk = (std::max(nc, np) + 255) / 256;
    d.partial = ws.get<double>(4 * (size_t)std::max(std::max(ebj, pbk), 1) + 16);
    d.scal = ws.get<double>(32);
    if (ws.err != hipSuccess) { ctx->set_error(std::string("rcn_ba_solve: workspace: ") + hipGetErrorString(ws.err)); return RCN_ERR_HIP; }
    d.uv = uv; d.ocam = p_ocam; d.opt = p_opt; d.cam_obs = p_camobs; d.pt_off = p_ptoff; d.cam_obs_off = p_camobsoff;
    d.cam_off = p_camoff; d.cam_dim = p_camdim; d.cols = p_cols; d.flag = p_flag;

    auto H2D = [&](void *dst, const void *src, size_t bytes) { return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess; };
    RCN_HIP(H2D(d.poses, pb->poses, sizeof(double) * 6 * nc));
    // TrustRegionMinimizer::IterationZero: start from the projection onto the box
    std::vector<double> intr0(pb->intrinsics, pb->intrinsics + 6 * (size_t)nc);
    if (opt->intrinsics_mode == 1)
        for (int c = 0; c < nc; ++c)
            for (int k = 0; k < 2; ++k)
                if (intr0[6 * c + k] > opt->focal_upper_bound) { intr0[6 * c + k] = opt->focal_upper_bound; sum->bound_projections++; }
    RCN_HIP(H2D(d.intr, intr0.data(), sizeof(double) * 6 * nc));
    if (res) {
        if (np > 0) RCN_HIP(hipMemcpyAsync(d.pts, res->pts, sizeof(double) * 3 * np, hipMemcpyDeviceToDevice, st));
        uv = const_cast<double *>(res->uv); p_ocam = const_cast<int *>(res->ocam); p_opt = const_cast<int *>(res->opt);
        d.uv = uv; d.ocam = p_ocam; d.opt = p_opt;
    } else {
        RCN_HIP(H2D(d.pts, pb->points, sizeof(double) * 3 * np));
        RCN_HIP(H2D(uv, pb->obs_uv, sizeof(double) * 2 * no));
        RCN_HIP(H2D(p_ocam, pb->obs_cam, sizeof(int) * no));
        RCN_HIP(H2D(p_opt, pb->obs_pt, sizeof(int) * no));
    }
    {   // the six structure arrays lie back to back on the device (p_camobs .. p_cols): one copy instead of six
        std::vector<int> stage((size_t)(p_flag - p_camobs));
        int *o = stage.data();
        auto put = [&](const std::vector<int> &v, size_t cnt) { if (cnt) memcpy(o, v.data(), cnt * sizeof(int)); o += cnt; };
        put(cam_obs, (size_t)no); put(pt_off, (size_t)np + 1); put(cam_obs_off, (size_t)nc + 1); put(cam_off, (size_t)nc + 1); put(cam_dim, (size_t)nc); put(cols, 10 * (size_t)nc);
        RCN_HIP(H2D(p_camobs, stage.data(), stage.size() * sizeof(int)));
        RCN_HIP(hipStreamSynchronize(st));      // `stage` leaves scope
    }
    RCN_HIP(hipMemsetAsync(p_flag, 0, 24 * sizeof(int), st));
    RCN_HIP(hipMemsetAsync(d.tickets, 0, 2 * (size_t)std::max(nc, 1) * sizeof(unsigned), st));
    RCN_HIP(hipMemsetAsync(vecs, 0, sizeof(double) * 12 * nvec, st));
    RCN_HIP(hipMemsetAsync(d.S, 0, sizeof(double) * (size_t)npad * npad, st));   // upper part / padding never rewritten
    RCN_HIP(hipMemsetAsync(d.Linv, 0, sizeof(double) * (size_t)nblk * NB * NB, st));   // upper triangles of the tile inverses stay zero
    if (nblk > 2) RCN_HIP(hipMemsetAsync(SI, 0, sizeof(double) * 2 * si_elems, st));   // blocks above a super-block inverse's diagonal stay zero (a column-0 pass reads 64 columns of one)
    RCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_diag), hipFuncAttributeMaxDynamicSharedMemorySize, NB * DL * 8));
    RCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_diag_server), hipFuncAttributeMaxDynamicSharedMemorySize, NB * DL * 8));
    RCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES));
    RCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES));
    RCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES));
    RCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<0, 16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES));
    RCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<0, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES));
    RCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe_tail), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES));
    { int rcm = ensure_chol_plan(ctx, nblk); if (rcm) return rcm; }
    uint64_t plain_token = 0;
    if (!res && no > 0) {
        const bool same = ctx->ba_graph_nc == nc && ctx->ba_graph_np == np && ctx->ba_graph_cam.size() == (size_t)no &&
                          memcmp(ctx->ba_graph_cam.data(), pb->obs_cam, sizeof(int) * (size_t)no) == 0 &&
                          memcmp(ctx->ba_graph_pt.data(), pb->obs_pt, sizeof(int) * (size_t)no) == 0;
        if (!same) {
            ctx->ba_graph_cam.assign(pb->obs_cam, pb->obs_cam + no);
            ctx->ba_graph_pt.assign(pb->obs_pt, pb->obs_pt + no);
            ctx->ba_graph_nc = nc; ctx->ba_graph_np = np;
            ctx->ba_graph_serial++;
        }
        plain_token = (1ull << 63) | ctx->ba_graph_serial;      // (a session's token carries its id in the upper half: never this large)
    }
    RCN_HIP(hipStreamSynchronize(st));   // host vectors go out of use; timing starts with inputs resident
    const double t_start = now_s();        // the pair lists of the Schur build are part of the solve (only the pack is not)
    // Whose lists these are: a session names its graph version (res->pair_token); a plain rcn_ba_solve is recognised by its observation
    // graph itself -- the arrays of the last plain solve are kept on the host and compared element by element (no hash: nothing to
    // collide), outside the timed region like the validation loop above.  The reference's loop never solves one graph twice
    // (SequentialReconstructor.cpp:1040-1094 adds a view before every adjust); a caller that re-solves after changing only the
    // estimates (another start, other options) skips the 0.75 ms the lists cost at 1M observations.
    const uint64_t pair_token = res ? res->pair_token : plain_token;
    // (the lists also depend on WHICH cameras have no free parameter -- pair_key drops their pairs -- i.e. on the options: ADVICE r4.  The
    //  per-camera dimensions the lists were built for are kept beside the token and compared; a session's token does not cover options either)
    const bool pairs_cached = pair_token != 0 && ctx->ba_pair_token == pair_token && ctx->ba_pair_camdim == cam_dim;
    // The lists are built on the panel stream, BESIDE the first evaluation (which needs nothing of them; both are eight launches or so, and
    // on the reference's problem sizes a solve is a few hundred launch-bound microseconds): the ctx stream waits for them in front of the loop.
    const bool pairs_build = gather && np > 0 && !pairs_cached;
    // (queued BEHIND the first evaluation's launches, below: on these sizes the host's enqueue rate is what the device waits for, and the
    //  evaluation is what the host waits for first)
    auto build_pairs = [&]() -> int {
        hipStream_t sp = ctx->panel_stream;          // idle: the stream was synchronised above and every factorisation joins its streams
        int *const long_keys = pk_cnt, *const long_count = pk_fill + nkeys;      // (the counts are dead once the lists are filled; a spare word behind the fill cursors)
        if (ctx->ba_pair_small && nkeys <= 1024 && npairs_lower <= 16384) k_pair_small<<<1, 1024, 0, sp>>>(d, pk_off, pk_list, nkeys, long_count);      // (the smallest of the reference's own sizes: one launch for six; beyond, one workgroup walks the pairs slower than six launches)
        else {
            RCN_HIP(hipMemsetAsync(pk, 0, sizeof(int) * (3 * (size_t)nkeys + 4 + (nkeys + 1023) / 1024), sp));
            const int thr = std::min(256, std::max(64, (kmax * kmax + 63) / 64 * 64));
            k_pair_count<<<np, thr, 0, sp>>>(d, pk_cnt);
            const int nchunks = (nkeys + 1023) / 1024;
            k_scan_sums<<<nchunks, 1024, 0, sp>>>(pk_cnt, pk_sums, nkeys);
            k_scan_top<<<1, 1024, 0, sp>>>(pk_sums, nchunks, pk_off + nkeys);
            k_scan_apply<<<nchunks, 1024, 0, sp>>>(pk_cnt, pk_sums, pk_off, nkeys);
            k_pair_fill<<<np, thr, 0, sp>>>(d, pk_off, pk_fill, pk_list);
        }
        k_pair_sort<<<(nkeys + 127) / 128, 128, 0, sp>>>(pk_off, pk_list, nkeys, long_keys, long_count);
        k_pair_sort_long<<<std::min(nkeys, 2048), 256, 0, sp>>>(pk_off, pk_list, long_keys, long_count);
        RCN_HIP(hipGetLastError());
        RCN_HIP(hipEventRecord(ctx->ba_ev[7], sp));
        return RCN_OK;
    };
    ctx->ba_pair_token = (gather && np > 0) ? pair_token : 0;
    if (!pairs_cached) ctx->ba_pair_camdim = cam_dim;
    sum->pair_lists_reused = pairs_cached ? 1 : 0;

    // phase times (summary.schur_seconds ...: HIP events on the stream) only where a phase outlasts the six event records of an
    // iteration: a 25-camera solve is ~35 launches of a few microseconds each, and every record is one more
    const bool phase_times = no >= 50000;
    double hs[32];
    // The cost at an ACCEPTED point (scal[0], written by the evaluation that also refreshes the Jacobian there) is not waited for: the
    // host needs it only when the next step's scalars come home, and every read below starts at scal[0] -- so it rides with whichever
    // read comes next (round 4, second session: one host round trip per iteration instead of two; the round trip and the idle stream
    // behind it are ~35 us, an eighth of an iteration of the reference's own problem sizes).
    double cost = 0.0;
    bool cost_pending = false;
    int cost_iter = 0;
    // (the pinned mirror serves the small problems, where the round trip is a quarter of an iteration; the large ones time their phases
    //  with events, which want the stream synchronised)
    double *const mirror = (!phase_times && no > 0) ? ctx->ba_host_scal : nullptr;
    auto read_scal = [&](int cnt, bool mirrored = false) -> hipError_t {
        hipError_t e = hipSuccess;
        if (mirrored && mirror) {
            const unsigned long long want = ctx->ba_host_seq;
            const unsigned long long *seqp = reinterpret_cast<const unsigned long long *>(mirror + 15);
            const double t0 = now_s();
            unsigned spins = 0;
            while (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != want) {
                if ((++spins & 0x3FFu) == 0) {
                    if (hipStreamQuery(st) == hipSuccess) break;                 // the stream has drained: the store is there or will never come
                    if (now_s() - t0 > 10.0) break;
                }
            }
            if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == want) for (int i = 0; i < cnt; ++i) hs[i] = mirror[i];
            else {      // not expected: the ordinary way, which also reports what went wrong
                e = hipMemcpyAsync(hs, d.scal, sizeof(double) * cnt, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
            }
        } else {
            e = hipMemcpyAsync(hs, d.scal, sizeof(double) * cnt, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(st);
        }
        if (e == hipSuccess && cost_pending) {
            cost = hs[0];
            if (cost_iter < 160) sum->cost_trace[cost_iter] = cost;
            cost_pending = false;
        }
        return e;
    };
    // cost at (poses,intr,pts) -> scal[slot]; JAC also refreshes the observation rows and the raw blocks
    bool jac_pending = false;
    // tickets of the in-kernel reductions (finish_sums): d.flag + 8 .. 11, zero at rest
    auto fin = [&](double *out, double scale, int which, bool with_flag, double *zero, double *out2 = nullptr, bool home = false) {
        Fin f{out, out2, reinterpret_cast<unsigned *>(d.flag + 8 + which), with_flag ? d.flag : nullptr, d.scal + 13, zero, scale, nullptr, d.scal, 0ull};
        if ((with_flag || home) && mirror) { f.host = mirror; f.seq = ++ctx->ba_host_seq; }      // the step's last reduction (the candidate's cost), the start's cost: the scalars go home by themselves
        return f;
    };
    auto eval = [&](bool jac, const double *ps, const double *in, const double *x, int slot, bool with_flag = false, bool rot_ready = false, bool home = false) -> hipError_t {
        if (no > 0) {
            if (jac) {
                // the streaming kernel of the solve, timed on its own (summary.jacobian_seconds): 224 B written per observation
                if (jac_pending && phase_times) {       // the previous pair of events has long completed (a read of the scalars came after it)
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, ctx->ba_tev[4], ctx->ba_tev[5]) == hipSuccess) { sum->jacobian_seconds += 1e-3 * ms; sum->jacobian_evals++; }
                    else (void)hipGetLastError();
                }
                if (!rot_ready) k_ba_cam_rot<<<(nc + 127) / 128, 128, 0, st>>>(ps, nc, d.crot);      // (an accepted candidate brings its table: k_ba_plus)
                if (phase_times) (void)hipEventRecord(ctx->ba_tev[4], st);
                k_ba_eval<true><<<ebj, 128, 0, st>>>(d, ps, in, x, d.partial, fin(d.scal + slot, 0.5, 0, false, d.scal + 12, nullptr, home));
                if (phase_times) (void)hipEventRecord(ctx->ba_tev[5], st);
                jac_pending = true;
            } else k_ba_eval<false><<<eb, 256, 0, st>>>(d, ps, in, x, d.partial, fin(d.scal + slot, 0.5, 0, with_flag, nullptr));
        } else {
            k_finish_sum<<<1, 256, 0, st>>>(d.partial, 0, d.scal + slot, 0.5, with_flag ? d.flag : nullptr, d.scal + 13);      // no observations: the sums are zero
            if (jac) (void)hipMemsetAsync(d.scal + 12, 0, sizeof(double), st);
        }
        if (jac) {
            const int pblocks = (np + 128 * CR_GROUPS - 1) / (128 * CR_GROUPS);
            k_ba_cam_raw<<<nc * csplit + pblocks, 128 * CR_GROUPS, 0, st>>>(d, csplit, nc * csplit);
        }
        return hipGetLastError();
    };

    RCN_HIP(eval(true, d.poses, d.intr, d.pts, 0, false, false, true));      // (its cost goes home by itself)
    const int dgrid = (std::max(n, 3 * np) + 255) / 256;
    k_ba_diag<<<std::max(dgrid, 1), 256, 0, st>>>(d, 0, 0.0, 0.0, opt->jacobi_scaling);
    RCN_HIP(hipGetLastError());
    if (pairs_build) { const int rcp = build_pairs(); if (rcp) return rcp; }
    RCN_HIP(read_scal(1, true));
    cost = hs[0];
    sum->initial_cost = cost;
    sum->initial_rms_px = std::sqrt(2.0 * cost / std::max(no, 1));
    sum->cost_trace[0] = cost;

    bool pairs_awaited = !pairs_build;      // (the first Schur build waits for the lists -- not the gradient, the landmark blocks and W / Y in front of it)
    double radius = opt->initial_trust_region_radius, decrease = 2.0;
    bool reuse_diag = false, need_gradient = true, grad_pending = false, diag_fresh = false, points_fresh = false;
    int invalid_run = 0, termination = 0, iter = 0;
    for (;;) {
        if (need_gradient) {
            // max |gradient| at the new point.  Its test comes BEFORE the next step in Ceres' loop; here the kernel is queued
            // and the value rides home with the scalars of the step that follows (one host synchronisation per iteration
            // instead of two: a quarter of a small problem's iteration).  If it was below the tolerance after all, that step
            // is dropped unseen -- it has only written candidate buffers -- and the loop ends where Ceres' would have.
            // (its cell, scal[12], was cleared by the evaluation at this point: k_ba_eval<true>, finish_sums)
            // (whenever the gradient is due, so is the LM diagonal -- an accepted step, or the start -- and both come from the raw blocks
            //  of the evaluation just queued: one launch; a loop that ends below has written a diagonal nobody reads)
            // (... and with the diagonal the landmarks' damped blocks at the radius the next step will use: k_ba_point_solve's launch)
            k_ba_gradmax<<<std::max(1, std::min(1024, (std::max(std::max(nc, n), 3 * np) + 255) / 256)), 256, 0, st>>>(d, d.scal + 12, reuse_diag ? 0 : 2,
                                                                                                                     opt->min_lm_diagonal, opt->max_lm_diagonal, 1.0 / radius);
            RCN_HIP(hipGetLastError());
            diag_fresh = !reuse_diag;
            points_fresh = !reuse_diag;
            need_gradient = false;
            grad_pending = true;
        }
        if (iter >= opt->max_iterations || radius <= opt->min_trust_region_radius) {
            // no step follows: the gradient test still comes first
            if (grad_pending) {
                RCN_HIP(read_scal(13));
                grad_pending = false;
                if (hs[12] <= opt->gradient_tolerance) { termination = RCN_BA_CONVERGENCE_GRADIENT; break; }
            }
            termination = iter >= opt->max_iterations ? RCN_BA_NO_CONVERGENCE : RCN_BA_CONVERGENCE_RADIUS;
            break;
        }
        ++iter;

        // ---- LM step
        if (!reuse_diag && !diag_fresh) {
            k_ba_diag<<<std::max(dgrid, 1), 256, 0, st>>>(d, 1, opt->min_lm_diagonal, opt->max_lm_diagonal, opt->jacobi_scaling);
            RCN_HIP(hipGetLastError());
        }
        diag_fresh = false;
        const double ir = 1.0 / radius;
        if (phase_times) RCN_HIP(hipEventRecord(ctx->ba_tev[0], st));
        // [0] breakdown / gate flag, [2..7] progress counters of the factorisation's streams.  In the default form (gather Schur build, the
        // right-hand side as row n of the padded system) the flag words, the padded rows and the right-hand-side row are all written by
        // the Schur diagonal launch (its cameras' workgroups and the workgroups behind them), and nothing has to be cleared up front: the Schur kernels write every entry of rhs below n.
        const bool rhs_row = npad > n && !ctx->ba_trsv_fwd;
        const bool fused_finish = gather && rhs_row;
        const bool chain = ctx->trsv_chain && 2 * nblk <= ctx->prop.multiProcessorCount;      // backward substitution as one launch
        if (!fused_finish) {
            RCN_HIP(hipMemsetAsync(d.flag, 0, 8 * sizeof(int), st));
            RCN_HIP(hipMemsetAsync(d.flag + 12, 0, 12 * sizeof(int), st));
            if (!gather) RCN_HIP(hipMemsetAsync(Sb, 0, sizeof(double) * 100 * (size_t)nc * nc, st));
            if (npad > n) RCN_HIP(hipMemsetAsync(d.S + (size_t)n * npad, 0, sizeof(double) * (size_t)(npad - n) * npad, st));
            RCN_HIP(hipMemsetAsync(d.rhs, 0, sizeof(double) * npad, st));
        }
        if (np > 0 && !points_fresh) k_ba_point_solve<<<(np + 127) / 128, 128, 0, st>>>(d, ir);      // (a step behind an accepted one: done in k_ba_gradmax's launch)
        points_fresh = false;
        if (no > 0) k_ba_wy<<<(unsigned)((10 * (size_t)no + 255) / 256), 256, 0, st>>>(d);
        if (gather) {
            if (!pairs_awaited) { RCN_HIP(hipStreamWaitEvent(st, ctx->ba_ev[7], 0)); pairs_awaited = true; }
            if (nc > 1) {
                const int nlow = nc * (nc - 1) / 2;
                if (npairs_lower / (size_t)nlow > 128) k_ba_schur_mfma_wg<<<nlow, 512, 0, st>>>(d, pk_off, pk_list);   // long lists: a workgroup per block
                else k_ba_schur_mfma<4><<<(nlow + 3) / 4, 256, 0, st>>>(d, pk_off, pk_list);
            }
            // (fused_finish: the padded rows, the right-hand-side row and the flag words ride in the same launch)
            const int fin = fused_finish ? (chain ? 1 : 0) : -1;
            const int fin_blocks = fused_finish ? (int)(((size_t)(npad - n) * npad + 1023) / 1024) : 0;
            if (nc < 128) k_ba_schur_diag_mfma<12><<<nc * csplit + fin_blocks, 1024, 0, st>>>(d, pk_off, pk_list, ir, csplit, fin);      // (few cameras: latency-bound, deep gathers)
            else k_ba_schur_diag_mfma<4><<<nc + fin_blocks, 1024, 0, st>>>(d, pk_off, pk_list, ir, 1, fin);
            if (!fused_finish && npad > n) k_ba_S_pad<<<(npad - n + 127) / 128, 128, 0, st>>>(d);
        } else {
            if (np > 0) k_ba_schur<<<np, std::min(256, std::max(64, 64 * ((kmax * kmax + 7) / 8))), 0, st>>>(d, Sb);
            k_ba_S_assemble<<<nc + 1, 256, 0, st>>>(d, Sb, ir);
            k_ba_cam_rhs<<<nc, 64, 0, st>>>(d);
        }
        if (rhs_row && !fused_finish) k_ba_S_rhs_row<<<(n + 1 + 255) / 256, 256, 0, st>>>(d);
        RCN_HIP(hipGetLastError());
        if (phase_times) RCN_HIP(hipEventRecord(ctx->ba_tev[1], st));
        // Dense Cholesky of the padded system: the schedule is DATA (chol_plan.h) -- a list of tile operations in an order that is a
        // correct sequential algorithm, each with its stream (A: the chain of diagonal blocks and critical tiles, B: panels and
        // columns, C: bulk trailing updates) and the device counters it waits for, derived from the tiles it reads and writes.
        // Hand-offs (Gate, above): every stream owns a progress counter; an operation publishes "everything before me on my stream
        // is done" with its FIRST thread (stream order has completed that work and the kernel boundary has released its writes) and
        // waits for other streams' counters itself -- inside the kernel on the chain and for the small kernels, in a ONE-WAVE gate
        // kernel in front of a pipelined launch (a grid of a thousand workgroups that spins while it holds its CU slots could keep
        // the very kernel it waits for from becoming resident).  A bulk update counts the tiles the next steps read first out one
        // by one (two classes, flag[5], flag[6]).  A wait that times out (2 s: a runtime that does not let the three streams
        // progress side by side) raises flag 3; the factorisation is then repeated on ONE stream in list order, and every later
        // one runs that way (ctx->chol_safe).  Same bits either way: no operation's arithmetic depends on where it runs.
        // (one block: the backward substitution rides in the diagonal kernel's launch)
        const bool trsv_in_diag = nblk == 1 && rhs_row && chain && fused_finish && ctx->trsv_chain;
        auto factorise = [&](bool safe) -> hipError_t {
            // (the chain on a high-priority stream of the library's own, forked from and joined to the caller's: RCN_CHOL_CHAIN_STREAM, tools/)
            const bool own_chain = !safe && ctx->chol_chain_stream && ctx->chain_stream;
            hipStream_t str[chol::N_STREAMS] = {own_chain ? ctx->chain_stream : st, safe ? st : ctx->panel_stream, safe ? st : ctx->aux_stream, safe ? st : ctx->panel2_stream, safe ? st : ctx->diag_stream};
#ifdef RCN_DIAG
            const bool swap_ab = !safe && ctx->chol_chain_stream == 4;      // the chain on the panel stream's handle, the panels on the caller's stream
            if (swap_ab) { str[0] = ctx->panel_stream; str[1] = st; }
#else
            const bool swap_ab = false;
#endif
            int *const ctr_base = d.flag + 12;          // the streams' progress counters and the two head-tile counters (the Schur diagonal launch's trailing workgroups clear them)
            int *ctr[chol::N_CTR];
            for (int c = 0; c < chol::N_CTR; ++c) ctr[c] = ctr_base + c;
            const chol::Plan &plan = ctx->chol_plan;
            if (!safe) {
                hipError_t e = hipEventRecord(ctx->ba_ev[0], st);
                if ((own_chain || swap_ab) && e == hipSuccess) e = hipStreamWaitEvent(str[0], ctx->ba_ev[0], 0);
                for (int s2 = 1; s2 < chol::N_STREAMS && e == hipSuccess; ++s2)
                    if (plan.n_ops[s2] && str[s2] != st) e = hipStreamWaitEvent(str[s2], ctx->ba_ev[0], 0);     // the other streams start behind everything queued so far
                if (e != hipSuccess) return e;
            }
            const unsigned *maps = ctx->bulk_map.as<unsigned>();
            const int ldsi = NB * std::max(plan.prm.tl_g, 1);
            const size_t lds_pipe = GST * GSTAGE_BYTES;
#ifdef RCN_DIAG
            const double th0 = now_s();
#endif
            // a bulk update whose launch carries the next super-step's panel product as its tail (chol_plan.h, fuse_with): found per host
            std::vector<int> tail_of(plan.ops.size(), -1);
            if (!safe)
                for (size_t i = 0; i < plan.ops.size(); ++i)
                    if (plan.ops[i].fuse_with >= 0) tail_of[(size_t)plan.ops[i].fuse_with] = (int)i;
            auto gate_of = [&](const chol::Op &o2, bool with_pub) {
                Gate g2 = gate_none(d.flag);
                g2.nw = o2.nw;
                for (int i = 0; i < o2.nw; ++i) { g2.c[i] = ctr[o2.w[i].ctr]; g2.n[i] = o2.w[i].val; }
                if (with_pub) { g2.pub = ctr[o2.stream]; g2.pubval = o2.ticket - 1; }
                return g2;
            };
            // the diagonal blocks: ONE resident workgroup for all of them (k_chol_diag_server), started in front of everything else
            const bool server = !safe && plan.n_ops[chol::ST_E] > 0;
            if (server) {
                const int n_active = rhs_row ? n + 1 : n;
                k_chol_diag_server<<<1, 64 * CDW, NB * DL * 8, str[chol::ST_E]>>>(d.S, npad, d.Linv, d.flag, ctx->diag_items.as<DiagItem>(), plan.n_ops[chol::ST_E], ctr_base, chol::ST_E, n_active);
            }
            for (size_t oi = 0; oi < plan.ops.size(); ++oi) {
                const chol::Op &op = plan.ops[oi];
                if (!safe && op.fuse_with >= 0) continue;      // its tiles went out with the bulk update in front of it
                if (server && op.stream == chol::ST_E) continue;      // the resident workgroup's
                hipStream_t sq = str[op.stream];
                Gate g = gate_none(d.flag);
                if (!safe) {
                    g.nw = op.nw;
                    for (int i = 0; i < op.nw; ++i) { g.c[i] = ctr[op.w[i].ctr]; g.n[i] = op.w[i].val; }
                    g.pub = ctr[op.stream]; g.pubval = op.ticket - 1;
#ifdef RCN_DIAG
                    // RCN_CHOL_BREAK=1: the critical tile of step 1 waits for a count that never comes -- the test of the fallback
                    if (ctx->chol_break && op.kind == chol::TRSM_Q && op.stream == chol::ST_A && op.kb == 1 && g.nw > 0) g.n[0] = 1 << 30;
#endif
                }
                // Where the wait stands.  On the chain (stream A) inside the kernel: its grids are small and nothing is saved by a launch in
                // front.  On every other stream in a ONE-WAVE gate kernel in front of the work, never inside it: a grid of hundreds of
                // workgroups that spins while it holds its CU slots -- and polls one counter from every workgroup -- could keep the very
                // kernel it waits for from becoming resident, and slows the chain's kernels beside it (measured: the critical-tile
                // kernels took 11 us instead of 4 with the panel kernels spinning next to them).
                const bool in_kernel = (op.stream == chol::ST_A && ctx->chol_gate_in_kernel >= 0) || ctx->chol_gate_in_kernel > 0;      // (-1, tools/ only: a gate kernel in front on the chain too)
                const bool pipe_kind = ((op.kind == chol::TRSM_PIPE || op.kind == chol::UPD_PIPE || op.kind == chol::PGEMM) && !op.small) || op.kind == chol::PUBLISH;
                Gate gk = gate_none(d.flag);           // what the kernel itself gets
                if (!safe) {
                    if (pipe_kind) {
                        if (op.nw > 0 || op.awaited) k_ring_gate<<<1, 64, 0, sq>>>(g);
                    } else if (in_kernel) gk = g;
                    else {
                        if (op.nw > 0) k_ring_gate<<<1, 64, 0, sq>>>(g);
                        gk.pub = g.pub; gk.pubval = g.pubval;      // publishing costs one store: no launch for that alone
                    }
                }
                const int gq = 32 * ((4 * op.m + 7) / 8);                          // k_gemm_q: strips of 32 rows on the eight XCD slots
                const int prio = (op.stream == chol::ST_C || (op.stream == chol::ST_D && !ctx->chol_pg_prio)) ? 0 : PIPE_PRIO;
                switch (op.kind) {
                case chol::DIAG: {
                    // (active rows of the block: the system's n rows, and the right-hand-side row behind them when it rides along)
                    const int nact = std::min(NB, (rhs_row ? n + 1 : n) - op.kb * NB);
                    k_chol_diag<<<1, 64 * CDW, NB * DL * 8, sq>>>(d.S, npad, op.kb, d.Linv, d.flag, op.kb == nblk - 1, gk, nact, op.tl,
                                                                   trsv_in_diag ? d.rhs : nullptr, n);
                    break;
                }
                case chol::TRSM_Q:
                    k_gemm_q<0><<<gq, 256, 0, sq>>>(d.S, d.L, npad, op.kb, op.first, op.m, d.Linv, gk, 1, op.tl);
                    break;
                case chol::UPD_Q:
                    k_gemm_q<1><<<gq, 256, 0, sq>>>(d.S, d.L, npad, op.kb, op.first, op.m, d.Linv, gk, op.dj, op.tl);
#ifdef RCN_DIAG
                    if (ctx->chol_break == 2 && op.stream == chol::ST_A && op.kb == 1 && !safe) k_diag_poison<<<1, 1, 0, sq>>>(d.S, npad, op.kb + 1);      // the next diagonal kernel meets a NaN pivot AFTER the timeout
#endif
                    break;
                case chol::TRSM_PIPE:
                    k_gemm_nt_pipe<0, 16, 1><<<op.map_n, 256, lds_pipe, sq>>>(d.L, d.S, npad, op.kb, maps + op.map_off, prio, nullptr, d.Linv, NB, 16, op.tl);
                    break;
                case chol::UPD_PIPE: {
                    if (op.small) { k_gemm_qm<1><<<16 * op.map_n, 256, 0, sq>>>(d.S, d.L, npad, op.kb, maps + op.map_off, op.nst / 16, nullptr, 0, gk, op.tl); break; }
                    int *sg = (op.stream == chol::ST_C && !safe) ? ctr[chol::CTR_SIG1] : nullptr;
                    if (op.nst == 16) k_gemm_nt_pipe<0, 16><<<op.map_n, 256, lds_pipe, sq>>>(d.S, d.L, npad, op.kb, maps + op.map_off, prio, sg, nullptr, 0, 16, op.tl);
                    else if (op.nst == 32) k_gemm_nt_pipe<0, 32><<<op.map_n, 256, lds_pipe, sq>>>(d.S, d.L, npad, op.kb, maps + op.map_off, prio, sg, nullptr, 0, 32, op.tl);
                    else if (tail_of[oi] >= 0) {
                        const chol::Op &tp = plan.ops[(size_t)tail_of[oi]];
                        k_gemm_nt_pipe_tail<<<op.map_n + tp.map_n, 256, lds_pipe, sq>>>(d.S, d.L, npad, op.kb, op.nst, maps + op.map_off, op.map_n, prio, sg, op.tl,
                                                                                          tp.kb, maps + tp.map_off, SI + (size_t)tp.dj * si_elems, ldsi, gate_of(tp, false), tp.tl);
                    }
                    else k_gemm_nt_pipe<0, 0><<<op.map_n, 256, lds_pipe, sq>>>(d.S, d.L, npad, op.kb, maps + op.map_off, prio, sg, nullptr, 0, op.nst, op.tl);
                    break;
                }
                case chol::SINV:
                    k_sinv<<<8 * op.pos + 1, 512, 0, sq>>>(d.L, npad, d.Linv, SI + (size_t)op.dj * si_elems, ldsi, op.kb, op.pos, gk, op.tl);
                    break;
                case chol::PGEMM:
                    if (op.small) { k_gemm_qm<2><<<16 * op.map_n, 256, 0, sq>>>(d.S, d.L, npad, op.kb, maps + op.map_off, 0, SI + (size_t)op.dj * si_elems, ldsi, gk, op.tl); break; }
                    k_gemm_nt_pipe<0, 0, 2><<<op.map_n, 256, lds_pipe, sq>>>(d.L, d.S, npad, op.kb, maps + op.map_off, prio, nullptr, SI + (size_t)op.dj * si_elems, ldsi, 0, op.tl);
                    break;
                case chol::PUBLISH:
                    break;
                }
            }
#ifdef RCN_DIAG
            if (ctx->chol_host_time) fprintf(stderr, "factorise: %zu operations enqueued in %.3f ms of host time\n", plan.ops.size(), 1e3 * (now_s() - th0));
#endif
            hipError_t e = hipGetLastError();
            if (e != hipSuccess || safe) return e;
            // the chain continues (triangular solves) behind the last kernels of the other streams
            for (int s2 = 1; s2 < chol::N_STREAMS && e == hipSuccess; ++s2) {
                if (!plan.n_ops[s2]) continue;
                if (str[s2] == st) continue;
                e = hipEventRecord(ctx->ba_ev[s2], str[s2]);
                if (e == hipSuccess) e = hipStreamWaitEvent(str[0], ctx->ba_ev[s2], 0);
            }
            if ((own_chain || swap_ab) && e == hipSuccess) {
                e = hipEventRecord(ctx->ba_ev[5], str[0]);
                if (e == hipSuccess) e = hipStreamWaitEvent(st, ctx->ba_ev[5], 0);
            }
            return e;
        };
        // (up to two blocks there is no panel rest, no column rest and no bulk update: nothing for the other two streams to do)
        RCN_HIP(factorise(ctx->chol_safe || nblk <= 2));
        if (phase_times) RCN_HIP(hipEventRecord(ctx->ba_tev[2], st));
        // (rhs_row + chain, the default: the sentinel was left by the Schur diagonal launch and the chain kernel reads y out of the factor's last row
        //  itself -- no launch in between; k_ba_y_from_row remains for the per-step kernels and the other ways to build the system)
        const bool y_in_row = chain && fused_finish;
        if (rhs_row && !y_in_row) k_ba_y_from_row<<<(npad + 255) / 256, 256, 0, st>>>(d, chain ? 1 : 0);
        else if (!rhs_row) for (int kb = 0; kb < nblk; ++kb) k_trsv_fwd<<<nblk - kb, 128, 0, st>>>(d.L, npad, kb, d.Linv, d.rhs, d.yc);
        if (chain) {
            if (!rhs_row) RCN_HIP(hipMemsetAsync(d.rhs, 0xFF, sizeof(double) * npad, st));      // the sentinel
            if (trsv_in_diag) {}      // done by k_chol_diag
            else if (y_in_row) k_trsv_bwd_chain<<<nblk, 512, 0, st>>>(d.L, npad, nblk, d.Linv, nullptr, d.rhs, d.flag, d.S, n);
            else k_trsv_bwd_chain<<<nblk, 512, 0, st>>>(d.L, npad, nblk, d.Linv, d.yc, d.rhs, d.flag);
        }
        else for (int kb = nblk - 1; kb >= 0; --kb) k_trsv_bwd<<<kb + 1, 512, 0, st>>>(d.L, npad, kb, d.Linv, d.yc, d.rhs);
        RCN_HIP(hipGetLastError());
        {   // the solution sits in d.rhs: the two kernels that read it take it from there (round 3 copied it to d.yc first)
            BaDev dx = d;
            dx.yc = d.rhs;
            if (no > 0) k_ba_backsub_obs<<<ebj, 128, 0, st>>>(dx);
            k_ba_backsub<<<std::max((std::max(n, np) + 127) / 128, 1), 128, 0, st>>>(dx);
        }
        if (no > 0) k_ba_model<<<ebj, 128, 0, st>>>(d, d.partial, fin(d.scal + 2, -1.0, 1, false, nullptr));
        else k_finish_sum<<<1, 256, 0, st>>>(d.partial, 0, d.scal + 2, -1.0);
        // candidate at alpha = 1 (+ norms, gradient.delta, finite check) and its cost
        // scal: [3] |dx|^2  [4] |x|^2  [5] g.delta, also in [8]  [6] non-finite count, also in [9]
        k_ba_plus<<<pbk, 256, 0, st>>>(d, 1.0, d.partial, pbk, fin(d.scal + 3, 1.0, 2, false, nullptr, d.scal + 8));
        RCN_HIP(hipGetLastError());
        RCN_HIP(eval(false, d.poses2, d.intr2, d.pts2, 1, true));       // + the factorisation's flag word into scal[13]
        if (phase_times) RCN_HIP(hipEventRecord(ctx->ba_tev[3], st));
        RCN_HIP(read_scal(14, true));
        const int hflag = (int)hs[13];
        if (grad_pending) {
            grad_pending = false;
            if (hs[12] <= opt->gradient_tolerance) {      // converged before this step: it does not count and is not looked at
                --iter;
                termination = RCN_BA_CONVERGENCE_GRADIENT;
                break;
            }
        }
        if (phase_times) {
            float ms = 0.f;
            RCN_HIP(hipEventElapsedTime(&ms, ctx->ba_tev[0], ctx->ba_tev[1])); sum->schur_seconds += 1e-3 * ms;
            RCN_HIP(hipEventElapsedTime(&ms, ctx->ba_tev[1], ctx->ba_tev[2])); sum->cholesky_seconds += 1e-3 * ms;
            RCN_HIP(hipEventElapsedTime(&ms, ctx->ba_tev[2], ctx->ba_tev[3])); sum->trisolve_seconds += 1e-3 * ms;
        }
        if (hflag == 4 && ctx->trsv_chain) {
            // the one-launch backward substitution gave up on a flag (its workgroups were not all resident): per-step kernels from now on
            ctx->trsv_chain = false;
            --iter;
            reuse_diag = true;
            continue;
        }
        if (hflag == 3 && !ctx->chol_safe) {
            // a cross-stream wait of the factorisation gave up: this runtime does not run the three streams side by side.
            // Not a numerical failure: switch to the one-stream schedule for good and redo this iteration's linear solve.
            ctx->chol_safe = true;
            --iter;
            reuse_diag = true;       // the LM diagonal of this iteration is already in place
            continue;
        }
        reuse_diag = true;
        const double model_change = hs[2];
        const bool solve_ok = hflag == 0 && hs[9] == 0.0 && std::isfinite(model_change);
        if (!solve_ok || !(model_change > 0.0)) {
            sum->invalid_steps++;
            if (++invalid_run >= opt->max_consecutive_invalid_steps) { termination = RCN_BA_FAILURE; break; }      // Ceres' HandleInvalidStep: pre-increment, `>=` (see the oracle)
            radius /= decrease; decrease *= 2.0; reuse_diag = false;
            if (iter < 160) sum->cost_trace[iter] = cost;
            continue;
        }
        invalid_run = 0;
        double cand_cost = hs[1], dn2 = hs[3], xn2 = hs[4];
        if (opt->intrinsics_mode == 1) {
            // bounds present: ArmijoLineSearch::DoSearch along delta, first trial the full step (already evaluated).
            // A trial that fails the sufficient-decrease test also gets its directional derivative
            // (k_ba_dirgrad against d.dlc / d.dlp = a * delta, hence the division by a).
            const double g0 = hs[8];
            rcn_ls::Sample start, prev, cur;
            start.x = 0.0; start.v = cost; start.g = g0; start.v_ok = start.g_ok = true;
            double a = 1.0, dmax = -1.0;
            int bt = 0;
            bool found = false;
            auto trial = [&](double alpha) -> hipError_t {   // candidate, its cost and the step norms at alpha
                k_ba_plus<<<pbk, 256, 0, st>>>(d, alpha, d.partial, pbk, fin(d.scal + 3, 1.0, 2, false, nullptr, d.scal + 14));      // ([14], [15]: nobody reads them here)
                hipError_t e = eval(false, d.poses2, d.intr2, d.pts2, 1);
                if (e != hipSuccess) return e;
                e = read_scal(5);
                cand_cost = hs[1]; dn2 = hs[3]; xn2 = hs[4];
                return e;
            };
            for (int it = 0;;) {
                cur = rcn_ls::Sample();
                cur.x = a; cur.v = cand_cost; cur.v_ok = std::isfinite(cand_cost);
                if (cur.v_ok && cur.v <= cost + 1e-4 * g0 * a) { found = true; break; }
                if (cur.v_ok) {
                    if (no > 0) k_ba_dirgrad<<<eb, 256, 0, st>>>(d, d.poses2, d.intr2, d.pts2, d.partial);
                    k_finish_sum<<<1, 256, 0, st>>>(d.partial, no > 0 ? eb : 0, d.scal + 10, 1.0);
                }
                if (dmax < 0.0) {                    // |delta|_inf, once per search (the first trial is a = 1)
                    RCN_HIP(hipMemsetAsync(d.scal + 11, 0, sizeof(double), st));
                    k_ba_absmax<<<std::max(1, std::min(1024, (std::max(n, 3 * np) + 255) / 256)), 256, 0, st>>>(
                        d.dlc, (size_t)n, d.dlp, 3 * (size_t)np, d.scal + 11);
                }
                RCN_HIP(hipGetLastError());
                RCN_HIP(read_scal(12));
                if (cur.v_ok) { cur.g = hs[10] / a; cur.g_ok = std::isfinite(cur.g); }
                if (dmax < 0.0) dmax = hs[11] / a;
                if (++it >= 20) break;
                const double an = rcn_ls::next_step(start, prev, cur, 1e-3 * a, 0.6 * a);
                if (an * dmax < 1e-9) break;
                prev = cur;
                a = an;
                sum->line_search_backtracks++;
                ++bt;
                RCN_HIP(trial(a));
            }
            if (!found && bt) RCN_HIP(trial(1.0));   // the search gave up: the full step stands
        }
        if (!std::isfinite(cand_cost)) cand_cost = DBL_MAX;
        if (std::sqrt(dn2) <= opt->parameter_tolerance * (std::sqrt(xn2) + opt->parameter_tolerance)) {
            termination = RCN_BA_CONVERGENCE_PARAMETER;
            if (iter < 160) sum->cost_trace[iter] = cost;
            break;
        }
        const double cost_change = cost - cand_cost;
        if (std::fabs(cost_change) <= opt->function_tolerance * cost) {
            termination = RCN_BA_CONVERGENCE_FUNCTION;
            if (iter < 160) sum->cost_trace[iter] = cost;
            break;
        }
        const double rho = cost_change / model_change;
        if (rho > opt->min_relative_decrease) {
            std::swap(d.poses, d.poses2); std::swap(d.intr, d.intr2); std::swap(d.pts, d.pts2); std::swap(d.crot, d.crot2);
            RCN_HIP(eval(true, d.poses, d.intr, d.pts, 0, false, true));
            cost_pending = true;          // cost = scal[0] at the next read (read_scal, above)
            cost_iter = iter;
            need_gradient = true;
            sum->successful_steps++;
            const double t = 2.0 * rho - 1.0;
            radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
            radius = std::min(opt->max_trust_region_radius, radius);
            decrease = 2.0; reuse_diag = false;
        } else {
            sum->unsuccessful_steps++;
            radius /= decrease; decrease *= 2.0; reuse_diag = true;
        }
        if (iter < 160 && !cost_pending) sum->cost_trace[iter] = cost;
    }
    if (cost_pending) RCN_HIP(read_scal(1));       // (every exit of the loop has read the scalars since the last accepted step: not reached)
    if (!pairs_awaited) RCN_HIP(hipStreamWaitEvent(st, ctx->ba_ev[7], 0));      // (a solve that ended before its first step: the lists it queued are kept for the next one)
    RCN_HIP(hipStreamSynchronize(st));
    sum->solve_seconds = now_s() - t_start;
    if (jac_pending && phase_times) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ba_tev[4], ctx->ba_tev[5]) == hipSuccess) { sum->jacobian_seconds += 1e-3 * ms; sum->jacobian_evals++; }
        else (void)hipGetLastError();
    }
    sum->iterations = iter;
    sum->factor_schedule = ctx->chol_safe ? 1 : 0;
    sum->termination = termination;
    sum->final_cost = cost;
    sum->final_rms_px = std::sqrt(2.0 * cost / std::max(no, 1));
    RCN_HIP(hipMemcpyAsync(pb->poses, d.poses, sizeof(double) * 6 * nc, hipMemcpyDeviceToHost, st));
    RCN_HIP(hipMemcpyAsync(pb->intrinsics, d.intr, sizeof(double) * 6 * nc, hipMemcpyDeviceToHost, st));
    if (np > 0) {
        if (res) RCN_HIP(hipMemcpyAsync(res->pts, d.pts, sizeof(double) * 3 * np, hipMemcpyDeviceToDevice, st));
        else RCN_HIP(hipMemcpyAsync(pb->points, d.pts, sizeof(double) * 3 * np, hipMemcpyDeviceToHost, st));
    }
    RCN_HIP(hipStreamSynchronize(st));
    return termination == RCN_BA_FAILURE ? RCN_ERR_NUMERIC : RCN_OK;
}
