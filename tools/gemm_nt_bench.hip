// tools/gemm_nt_bench.hip -- diagnostic: the bulk trailing-update kernels on a synthetic 10112^2 system (kb = 0).
#include "../reconstructor_amd/csrc/ba.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int nblk = 79, npad = nblk * NB, mt = nblk - 2;
    const size_t N = (size_t)npad * npad;
    std::vector<double> h(N);
    unsigned long long x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 - 0.5; }
    double *S0, *S1, *L;
    (void)hipMalloc(&S0, N * 8); (void)hipMalloc(&S1, N * 8); (void)hipMalloc(&L, N * 8);
    (void)hipMemcpy(L, h.data(), N * 8, hipMemcpyHostToDevice);
#define SETA(D) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_ring<D>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES)
    SETA(0); SETA(1); SETA(2); SETA(3); SETA(4); SETA(7);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flop = 2.0 * 128 * 128 * 128 * (double)mt * (mt + 1) / 2;
    for (int rep = 0; rep < 4; ++rep) {
        float ms0, ms1;
        (void)hipMemset(S0, 0, N * 8); (void)hipMemset(S1, 0, N * 8);
        ms0 = 0.f;
        (void)hipEventRecord(e0);
        k_gemm_nt_ring<0><<<gemm_nt_grid(mt), 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, mt, 16);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms1, e0, e1);
        if (rep == 3) {
#define RUNV(D) { float m; (void)hipEventRecord(e0); k_gemm_nt_ring<D><<<gemm_nt_grid(mt), 256, GST * GSTAGE_BYTES>>>(S0, L, npad, 0, mt, 16); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&m, e0, e1); printf("  variant %d: %.1f us (%.1f TF)\n", D, m * 1e3, flop / m * 1e-9); }
            RUNV(1) RUNV(2) RUNV(3) RUNV(4) RUNV(7)
        }
        printf("rep %d: ring %.1f us (%.1f TF)   %s\n", rep, ms1 * 1e3, flop / ms1 * 1e-9, hipGetErrorString(hipGetLastError()));
    }
    // check one tile (ti = 5, tj = 3) against the host: S1 = 0 - A B^T
    (void)hipMemset(S1, 0, N * 8);
    k_gemm_nt_ring<0><<<gemm_nt_grid(mt), 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, mt, 16);
    std::vector<double> c((size_t)128 * npad);
    (void)hipMemcpy(c.data(), S1 + (size_t)5 * 128 * npad, c.size() * 8, hipMemcpyDeviceToHost);
    double md = 0;
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
            double sref = 0;
            for (int k = 0; k < 128; ++k) sref -= h[((size_t)5 * 128 + i) * npad + k] * h[((size_t)3 * 128 + j) * npad + k];
            md = fmax(md, fabs(sref - c[(size_t)i * npad + 3 * 128 + j]));
        }
    printf("max |ring - host| on tile (5,3) = %.3e\n", md);
    return 0;
}
