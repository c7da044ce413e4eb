// tools/chol_kernels_check.hip -- the factorisation's round-5 kernels one by one against host arithmetic (tests/test_ba_gpu.py runs it):
//   * k_gemm_nt_pipe<0, 0>     the rolled trailing update at 24 .. 128 stages (K = 192 .. 1024), one tile checked in full
//   * k_gemm_nt_pipe<0, 0, 2>  the panel product of the two-level regime, L(i, p + c) = S(i, p .. p + c) W[c][.]', columns 1 .. 7
//   * k_gemm_qm<1>, <2>        the latency form of both for the handful of head tiles
//   * k_sinv                   block rows of a super-block's inverse, g = 4 and g = 8:  W L_JJ = I
// Prints one line per check and "ALL OK" at the end; exit code 1 on the first failure.
#include "../reconstructor_amd/csrc/ba.hip"
#include <cstdio>
#include <vector>

static double rnd(unsigned long long &x)
{
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return (double)(x >> 11) / 9007199254740992.0 - 0.5;
}

int main()
{
    const int nblk = 12, npad = nblk * NB;
    const size_t N = (size_t)npad * npad;
    unsigned long long seed = 88172645463325252ull;
    std::vector<double> hL(N), hS(N);
    for (auto &v : hL) v = rnd(seed);
    for (auto &v : hS) v = rnd(seed);
    double *dS, *dL, *dOut;
    (void)hipMalloc(&dS, N * 8); (void)hipMalloc(&dL, N * 8); (void)hipMalloc(&dOut, N * 8);
    (void)hipMemcpy(dL, hL.data(), N * 8, hipMemcpyHostToDevice);
    unsigned *dmap;
    (void)hipMalloc(&dmap, 64 * sizeof(unsigned));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<0, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES);
    bool ok = true;
    // ---- rolled trailing update: tiles (10, 9), (11, 11) with panels from column block 0
    for (int nst : {24, 28, 32, 48, 64, 100, 128}) {
        unsigned hm[8] = {chol::map_entry(10, 9), chol::map_entry(11, 11, 1), ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
        (void)hipMemcpy(dmap, hm, sizeof(hm), hipMemcpyHostToDevice);
        (void)hipMemcpy(dS, hS.data(), N * 8, hipMemcpyHostToDevice);
        k_gemm_nt_pipe<0, 0><<<8, 256, GST * GSTAGE_BYTES>>>(dS, dL, npad, 0, dmap, 0, nullptr, nullptr, 0, nst, 0);
        std::vector<double> out(N);
        (void)hipMemcpy(out.data(), dS, N * 8, hipMemcpyDeviceToHost);
        double md = 0, mo = 0;
        for (int i = 0; i < npad; ++i)
            for (int j = 0; j < npad; ++j) {
                const int ti = i / NB, tj = j / NB;
                double want = hS[(size_t)i * npad + j];
                if ((ti == 10 && tj == 9) || (ti == 11 && tj == 11))
                    for (int k = 0; k < 8 * nst; ++k) want -= hL[(size_t)i * npad + k] * hL[(size_t)j * npad + k];
                const double dd = fabs(want - out[(size_t)i * npad + j]);
                if ((ti == 10 && tj == 9) || (ti == 11 && tj == 11)) md = fmax(md, dd); else mo = fmax(mo, dd);
            }
        const bool good = md < 1e-11 && mo == 0.0 && hipGetLastError() == hipSuccess;
        printf("rolled update K = %4d: max error %.2e, max change elsewhere %.2e  %s\n", 8 * nst, md, mo, good ? "ok" : "FAILED");
        ok &= good;
    }
    // ---- the panel product with a super-block's inverse (random W, row stride 8 * 128): tiles (9, c), (11, c)
    {
        const int g = 8, ldsi = g * NB, p = 1;
        std::vector<double> hW((size_t)ldsi * ldsi);
        for (auto &v : hW) v = rnd(seed);
        double *dW;
        (void)hipMalloc(&dW, hW.size() * 8);
        (void)hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice);
        std::vector<unsigned> hm;
        for (int c = g - 1; c >= 0; --c) { hm.push_back(chol::map_entry(9, c)); hm.push_back(chol::map_entry(11, c)); }      // (column 0 runs as a 24-stage pass: K = 192)
        while (hm.size() % 8) hm.push_back(~0u);
        (void)hipMemcpy(dmap, hm.data(), hm.size() * sizeof(unsigned), hipMemcpyHostToDevice);
        (void)hipMemcpy(dS, hS.data(), N * 8, hipMemcpyHostToDevice);
        (void)hipMemset(dOut, 0, N * 8);
        k_gemm_nt_pipe<0, 0, 2><<<(unsigned)hm.size(), 256, GST * GSTAGE_BYTES>>>(dOut, dS, npad, p, dmap, PIPE_PRIO, nullptr, dW, ldsi, 0, 0);
        std::vector<double> out(N);
        (void)hipMemcpy(out.data(), dOut, N * 8, hipMemcpyDeviceToHost);
        double md = 0, mo = 0;
        for (int i = 0; i < npad; ++i)
            for (int j = 0; j < npad; ++j) {
                const int ti = i / NB, tj = j / NB, c = tj - p;
                double want = 0.0;
                const bool mine = (ti == 9 || ti == 11) && c >= 0 && c < g;
                if (mine)
                    for (int k = 0; k < (c == 0 ? 192 : NB * (c + 1)); ++k) want += hS[(size_t)i * npad + (size_t)p * NB + k] * hW[((size_t)c * NB + j % NB) * ldsi + k];
                const double dd = fabs(want - out[(size_t)i * npad + j]);
                if (mine) md = fmax(md, dd); else mo = fmax(mo, dd);
            }
        const bool good = md < 1e-11 && mo == 0.0 && hipGetLastError() == hipSuccess;
        printf("panel product with the super-block's inverse, columns 0 .. %d: max error %.2e, elsewhere %.2e  %s\n", g - 1, md, mo, good ? "ok" : "FAILED");
        ok &= good;
        (void)hipFree(dW);
    }
    // ---- the latency form of both (k_gemm_qm): a list of tiles, several panels per tile
    {
        const int g = 4, ldsi = g * NB, p = 2;
        int *dflag;
        (void)hipMalloc(&dflag, 64); (void)hipMemset(dflag, 0, 64);
        // update: tiles (9, 8), (10, 10), four panels from column block p
        unsigned hm[8] = {chol::map_entry(9, 8), chol::map_entry(10, 10), ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
        (void)hipMemcpy(dmap, hm, sizeof(hm), hipMemcpyHostToDevice);
        (void)hipMemcpy(dS, hS.data(), N * 8, hipMemcpyHostToDevice);
        k_gemm_qm<1><<<16 * 8, 256>>>(dS, dL, npad, p, dmap, g, nullptr, 0, gate_none(dflag), 0);
        std::vector<double> out(N);
        (void)hipMemcpy(out.data(), dS, N * 8, hipMemcpyDeviceToHost);
        double md = 0, mo = 0;
        for (int i = 0; i < npad; ++i)
            for (int j = 0; j < npad; ++j) {
                const int ti = i / NB, tj = j / NB;
                const bool mine = (ti == 9 && tj == 8) || (ti == 10 && tj == 10);
                double want = hS[(size_t)i * npad + j];
                if (mine)
                    for (int k = 0; k < NB * g; ++k) want -= hL[(size_t)i * npad + (size_t)p * NB + k] * hL[(size_t)j * npad + (size_t)p * NB + k];
                const double dd = fabs(want - out[(size_t)i * npad + j]);
                if (mine) md = fmax(md, dd); else mo = fmax(mo, dd);
            }
        bool good = md < 1e-11 && mo == 0.0 && hipGetLastError() == hipSuccess;
        printf("latency-form update, K = %d: max error %.2e, elsewhere %.2e  %s\n", NB * g, md, mo, good ? "ok" : "FAILED");
        ok &= good;
        // panel product: tiles (7, c), (11, c), c = 0 .. 3, random W
        std::vector<double> hW((size_t)ldsi * ldsi);
        for (auto &v : hW) v = rnd(seed);
        double *dW;
        (void)hipMalloc(&dW, hW.size() * 8);
        (void)hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice);
        unsigned hm2[8] = {chol::map_entry(7, 3), chol::map_entry(11, 3), chol::map_entry(7, 2), chol::map_entry(11, 2), chol::map_entry(7, 1), chol::map_entry(11, 1), chol::map_entry(7, 0), chol::map_entry(11, 0)};
        (void)hipMemcpy(dmap, hm2, sizeof(hm2), hipMemcpyHostToDevice);
        (void)hipMemcpy(dS, hS.data(), N * 8, hipMemcpyHostToDevice);
        (void)hipMemset(dOut, 0, N * 8);
        k_gemm_qm<2><<<16 * 8, 256>>>(dS, dOut, npad, p, dmap, 0, dW, ldsi, gate_none(dflag), 0);
        (void)hipMemcpy(out.data(), dOut, N * 8, hipMemcpyDeviceToHost);
        md = mo = 0;
        for (int i = 0; i < npad; ++i)
            for (int j = 0; j < npad; ++j) {
                const int ti = i / NB, tj = j / NB, c = tj - p;
                const bool mine = (ti == 7 || ti == 11) && c >= 0 && c < g;
                double want = 0.0;
                if (mine)
                    for (int k = 0; k < NB * (c + 1); ++k) want += hS[(size_t)i * npad + (size_t)p * NB + k] * hW[((size_t)c * NB + j % NB) * ldsi + k];
                const double dd = fabs(want - out[(size_t)i * npad + j]);
                if (mine) md = fmax(md, dd); else mo = fmax(mo, dd);
            }
        good = md < 1e-11 && mo == 0.0 && hipGetLastError() == hipSuccess;
        printf("latency-form panel product, columns 0 .. %d: max error %.2e, elsewhere %.2e  %s\n", g - 1, md, mo, good ? "ok" : "FAILED");
        ok &= good;
        (void)hipFree(dW); (void)hipFree(dflag);
    }
    // ---- the super-block's inverse: a lower-triangular L_JJ (g x g tiles at block p of the big matrix), its tile inverses from the host
    for (int g : {4, 8}) {
        const int p = 2, ldsi = g * NB, n = g * NB;
        std::vector<double> T((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) T[(size_t)i * n + j] = i == j ? 4.0 + rnd(seed) : 0.1 * rnd(seed);
        std::vector<double> hLm(N, 0.0), hLinv((size_t)nblk * NB * NB, 0.0);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (i / NB != j / NB) hLm[((size_t)p * NB + i) * npad + (size_t)p * NB + j] = T[(size_t)i * n + j];
        for (int b = 0; b < g; ++b) {          // inverse of the diagonal tile b by forward substitution
            double *X = hLinv.data() + (size_t)(p + b) * NB * NB;
            for (int c = 0; c < NB; ++c)
                for (int r = c; r < NB; ++r) {
                    double s = r == c ? 1.0 : 0.0;
                    for (int k = c; k < r; ++k) s -= T[(size_t)(b * NB + r) * n + b * NB + k] * X[(size_t)k * NB + c];
                    X[(size_t)r * NB + c] = s / T[(size_t)(b * NB + r) * n + b * NB + r];
                }
        }
        double *dLinv, *dSI;
        int *dflag;
        (void)hipMalloc(&dLinv, hLinv.size() * 8); (void)hipMalloc(&dSI, (size_t)ldsi * ldsi * 8); (void)hipMalloc(&dflag, 64);
        (void)hipMemset(dflag, 0, 64); (void)hipMemset(dSI, 0, (size_t)ldsi * ldsi * 8);
        (void)hipMemcpy(dLinv, hLinv.data(), hLinv.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(dOut, hLm.data(), N * 8, hipMemcpyHostToDevice);
        for (int pos = 0; pos < g; ++pos) k_sinv<<<8 * pos + 1, 512>>>(dOut, npad, dLinv, dSI, ldsi, p, pos, gate_none(dflag), 0);
        std::vector<double> W((size_t)ldsi * ldsi);
        (void)hipMemcpy(W.data(), dSI, W.size() * 8, hipMemcpyDeviceToHost);
        double md = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0.0;
                for (int k = j; k <= i; ++k) s += W[(size_t)i * ldsi + k] * T[(size_t)k * n + j];
                md = fmax(md, fabs(s - (i == j ? 1.0 : 0.0)));
            }
        const bool good = md < 1e-12 && hipGetLastError() == hipSuccess;
        printf("super-block inverse, g = %d: max |W L - I| = %.2e  %s\n", g, md, good ? "ok" : "FAILED");
        ok &= good;
        (void)hipFree(dLinv); (void)hipFree(dSI); (void)hipFree(dflag);
    }
    printf(ok ? "ALL OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
