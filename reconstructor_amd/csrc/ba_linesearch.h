// ba_linesearch.h -- the polynomials of Ceres' Armijo line search (internal/ceres/line_search.cc
// InterpolatingPolynomialMinimizingStepSize, internal/ceres/polynomial.cc), restated on the host for the
// bounds-constrained bundle adjustment (TrustRegionMinimizer::DoLineSearch, reached from
// BundleAdjuster.cpp:117-121 when the focal lengths carry an upper bound).  Interpolation type CUBIC (the
// Solver::Options default): a backtrack minimises, on the contraction interval, the polynomial through value AND
// directional derivative at step 0, at the current trial and -- from the second backtrack on -- at the previous one.
#pragma once
#include <cmath>

namespace rcn_ls {

struct Sample { double x = 0.0, v = 0.0, g = 0.0; bool v_ok = false, g_ok = false; };

inline double poly_eval(const double *p, int deg, double x)
{
    double v = 0.0;
    for (int i = 0; i <= deg; ++i) v = v * x + p[i];
    return v;
}

// FindInterpolatingPolynomial: one equation per valid value / gradient; Gaussian elimination with full pivoting
// (Eigen FullPivLU, threshold 0: unknowns behind a zero pivot stay 0).  Highest power first; returns the degree.
inline int fit(const Sample *s, int ns, double *coef)
{
    int nc = 0;
    for (int i = 0; i < ns; ++i) nc += (s[i].v_ok ? 1 : 0) + (s[i].g_ok ? 1 : 0);
    const int deg = nc - 1;
    double A[6][6], b[6], y[6];
    int perm[6], row = 0;
    for (int i = 0; i < ns; ++i) {
        if (s[i].v_ok) {
            for (int j = 0; j <= deg; ++j) A[row][j] = std::pow(s[i].x, deg - j);
            b[row++] = s[i].v;
        }
        if (s[i].g_ok) {
            for (int j = 0; j < deg; ++j) A[row][j] = (deg - j) * std::pow(s[i].x, deg - j - 1);
            A[row][deg] = 0.0;
            b[row++] = s[i].g;
        }
    }
    for (int j = 0; j < nc; ++j) perm[j] = j;
    int rank = 0;
    for (int k = 0; k < nc; ++k) {
        int pi = k, pj = k;
        double best = 0.0;
        for (int i = k; i < nc; ++i)
            for (int j = k; j < nc; ++j)
                if (std::fabs(A[i][j]) > best) { best = std::fabs(A[i][j]); pi = i; pj = j; }
        if (best == 0.0) break;
        for (int j = 0; j < nc; ++j) std::swap(A[k][j], A[pi][j]);
        std::swap(b[k], b[pi]);
        for (int i = 0; i < nc; ++i) std::swap(A[i][k], A[i][pj]);
        std::swap(perm[k], perm[pj]);
        for (int i = k + 1; i < nc; ++i) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j < nc; ++j) A[i][j] -= f * A[k][j];
            b[i] -= f * b[k];
        }
        rank = k + 1;
    }
    for (int k = nc - 1; k >= 0; --k) {
        if (k >= rank) { y[k] = 0.0; continue; }
        double t = b[k];
        for (int j = k + 1; j < rank; ++j) t -= A[k][j] * y[j];
        y[k] = t / A[k][k];
    }
    for (int k = 0; k < nc; ++k) coef[perm[k]] = y[k];
    return deg;
}

// FindPolynomialRoots, real parts only (all MinimizePolynomial reads): leading zeros dropped, closed forms for
// degree 1 and 2 (FindQuadraticPolynomialRoots), a simultaneous (Durand-Kerner) iteration on the monic polynomial
// where Ceres takes the eigenvalues of the companion matrix.  Returns the number of roots.
inline int root_real_parts(const double *p, int deg, double *re)
{
    while (deg > 0 && p[0] == 0.0) { ++p; --deg; }
    if (deg == 0) return 0;
    if (deg == 1) { re[0] = -p[1] / p[0]; return 1; }
    if (deg == 2) {
        const double a = p[0], b = p[1], c = p[2], D = b * b - 4.0 * a * c, sD = std::sqrt(std::fabs(D));
        if (D >= 0.0) {
            if (b >= 0.0) { re[0] = (-b - sD) / (2.0 * a); re[1] = (2.0 * c) / (-b - sD); }
            else { re[0] = (2.0 * c) / (-b + sD); re[1] = (-b + sD) / (2.0 * a); }
        } else re[0] = re[1] = -b / (2.0 * a);
        return 2;
    }
    double m[8], zr[8], zi[8], bound = 0.0;
    for (int i = 0; i <= deg; ++i) m[i] = p[i] / p[0];
    for (int i = 1; i <= deg; ++i) bound = std::fmax(bound, std::fabs(m[i]));
    bound += 1.0;                                   // Cauchy: every root lies within
    {
        double cr = 1.0, ci = 0.0;                   // powers of 0.4 + 0.9 i, scaled to half the bound
        for (int k = 0; k < deg; ++k) {
            zr[k] = 0.5 * bound * cr; zi[k] = 0.5 * bound * ci;
            const double nr = cr * 0.4 - ci * 0.9, ni = cr * 0.9 + ci * 0.4;
            cr = nr; ci = ni;
        }
    }
    for (int it = 0; it < 2000; ++it) {
        double moved = 0.0, size = 0.0;
        for (int k = 0; k < deg; ++k) {
            double pr = 1.0, pim = 0.0;              // monic p(z_k), Horner
            for (int i = 1; i <= deg; ++i) {
                const double tr = pr * zr[k] - pim * zi[k] + m[i], ti = pr * zi[k] + pim * zr[k];
                pr = tr; pim = ti;
            }
            double qr = 1.0, qi = 0.0;               // prod_{j != k} (z_k - z_j)
            for (int j = 0; j < deg; ++j) {
                if (j == k) continue;
                const double dr = zr[k] - zr[j], di = zi[k] - zi[j];
                const double tr = qr * dr - qi * di, ti = qr * di + qi * dr;
                qr = tr; qi = ti;
            }
            const double den = qr * qr + qi * qi;
            if (den == 0.0) continue;
            const double sr = (pr * qr + pim * qi) / den, si = (pim * qr - pr * qi) / den;
            zr[k] -= sr; zi[k] -= si;
            moved = std::fmax(moved, std::fmax(std::fabs(sr), std::fabs(si)));
            size = std::fmax(size, std::fmax(std::fabs(zr[k]), std::fabs(zi[k])));
        }
        if (moved <= 1e-16 * std::fmax(size, 1e-300)) break;
    }
    for (int k = 0; k < deg; ++k) re[k] = zr[k];
    return deg;
}

// MinimizePolynomial over [lo, hi]: the midpoint, the two ends, then every root of the derivative inside
inline double minimize(const double *p, int deg, double lo, double hi)
{
    double best_x = 0.5 * (lo + hi), best = poly_eval(p, deg, best_x);
    const double vlo = poly_eval(p, deg, lo), vhi = poly_eval(p, deg, hi);
    if (vlo < best) { best = vlo; best_x = lo; }
    if (vhi < best) { best = vhi; best_x = hi; }
    if (deg < 2) return best_x;
    double dp[8], re[8];
    for (int i = 0; i < deg; ++i) dp[i] = (deg - i) * p[i];
    const int nr = root_real_parts(dp, deg - 1, re);
    for (int i = 0; i < nr; ++i) {
        if (re[i] < lo || re[i] > hi) continue;
        const double v = poly_eval(p, deg, re[i]);
        if (v < best) { best = v; best_x = re[i]; }
    }
    return best_x;
}

// the next trial step of the Armijo search
inline double next_step(const Sample &start, const Sample &prev, const Sample &cur, double lo, double hi)
{
    if (!cur.v_ok) return std::fmin(std::fmax(cur.x * 0.5, lo), hi);
    Sample s[3];
    int ns = 0;
    s[ns++] = start;
    s[ns++] = cur;
    if (prev.v_ok) s[ns++] = prev;
    double coef[6];
    const int deg = fit(s, ns, coef);
    return minimize(coef, deg, lo, hi);
}

}  // namespace rcn_ls
