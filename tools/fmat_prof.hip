// tools/fmat_prof.hip -- diagnostic: cycles per phase of k_fmat_filter on a synthetic grid (RCN_FM_PROF build).
#define RCN_FM_PROF 1
#include "../reconstructor_amd/csrc/fmat.hip"
#include <cstdio>
#include <random>
#include <vector>
int main(int argc, char **argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 2000, n = 512;
    const double frac = argc > 2 ? atof(argv[2]) : 0.3;
    std::mt19937_64 g(5);
    std::uniform_real_distribution<double> U(0, 1);
    std::normal_distribution<double> N(0, 0.5);
    std::vector<int32_t> off(P + 1), a(2 * (size_t)P * n), b(2 * (size_t)P * n);
    for (int p = 0; p <= P; ++p) off[p] = p * n;
    for (int p = 0; p < P; ++p) {
        const double th = 0.4 * U(g) - 0.2, tx = 2 * U(g) - 1, ty = 0.3 * U(g), tz = 0.3 * U(g), f = 614.4;
        for (int i = 0; i < n; ++i) {
            const double X = 4 * U(g) - 2, Y = 4 * U(g) - 2, Z = 4 + 5 * U(g);
            const double X2 = cos(th) * X + sin(th) * Z + tx, Y2 = Y + ty, Z2 = -sin(th) * X + cos(th) * Z + tz;
            size_t o = 2 * ((size_t)p * n + i);
            a[o] = (int)(f * X / Z + 256 + N(g)); a[o + 1] = (int)(f * Y / Z + 168 + N(g));
            if (U(g) < frac) { b[o] = (int)(512 * U(g)); b[o + 1] = (int)(336 * U(g)); }
            else { b[o] = (int)(f * X2 / Z2 + 256 + N(g)); b[o + 1] = (int)(f * Y2 / Z2 + 168 + N(g)); }
        }
    }
    int32_t *d_off, *d_a, *d_b, *d_cnt, *d_it; uint8_t *d_mask;
    (void)hipMalloc(&d_off, off.size() * 4); (void)hipMalloc(&d_a, a.size() * 4); (void)hipMalloc(&d_b, b.size() * 4);
    (void)hipMalloc(&d_cnt, P * 4); (void)hipMalloc(&d_it, P * 4); (void)hipMalloc(&d_mask, (size_t)P * n);
    (void)hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_a, a.data(), a.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    FmatArgs args; args.off = d_off; args.xy1 = d_a; args.xy2 = d_b; args.n_pairs = P; args.mask = d_mask; args.counts = d_cnt; args.iters = d_it; args.F = nullptr;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fm_prof), z, sizeof(z));
        (void)hipEventRecord(e0);
        k_fmat_filter<<<P, 256>>>(args);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_fm_prof), sizeof(z));
        std::vector<int32_t> it(P); (void)hipMemcpy(it.data(), d_it, P * 4, hipMemcpyDeviceToHost);
        double mean = 0; for (int v : it) mean += v; mean /= P;
        double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)z[i];
        printf("%.2f ms, %d pairs, mean iterations %.1f; share of wave time: draw %.1f%% gather+collinear %.1f%% solve %.1f%% score %.1f%% accept %.1f%% mask %.1f%%  (%.0f kcycles per pair)  %s\n", ms, P, mean,
               100 * z[0] / tot, 100 * z[1] / tot, 100 * z[2] / tot, 100 * z[3] / tot, 100 * z[4] / tot, 100 * z[5] / tot, tot / P / 1e3, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
