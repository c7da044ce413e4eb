// tools/chol_diag_bench.hip -- diagnostic: phase timing of k_chol_diag on one 128x128 SPD block.
#define RCN_STAMP 1
#include "../reconstructor_amd/csrc/ba.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int n = 128;
    std::vector<double> A(n * n), M(n * n);
    srand(3);
    for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0; for (int k = 0; k < n; ++k) s += M[i * n + k] * M[j * n + k];
            A[i * n + j] = s + (i == j ? n : 0);
        }
    double *dA, *dLinv; int *dflag;
    (void)hipMalloc(&dA, n * n * 8); (void)hipMalloc(&dLinv, n * n * 8); (void)hipMalloc(&dflag, 4);
    (void)hipMemset(dflag, 0, 4); (void)hipMemset(dLinv, 0, n * n * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_diag), hipFuncAttributeMaxDynamicSharedMemorySize, NB * DL * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {      // rep 0 also stores L (checked below); the others time the kernel as the chain runs it
        (void)hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
        (void)hipEventRecord(e0);
        k_chol_diag<<<1, 64 * CDW, NB * DL * 8>>>(dA, n, 0, dLinv, dflag, rep == 0, gate_none(dflag));
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[64];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
        printf("rep %d: %.1f us total; ticks(100MHz?) ", rep, ms * 1e3);
        printf("factor(all leaves+trsm+trail)=%llu | store=", st[13] - st[0]);
        printf("%llu last inverse row=%llu\n", st[14] - st[13], st[16] - st[14]);
        if (rep == 3) {
            printf("  first leaf %llu;", st[1] - st[0]);
            unsigned long long prev = st[1];
            for (int i = 0; i < 7; ++i) { printf(" [solve %llu, update+leaf %llu]", st[32 + 2 * i] - prev, st[33 + 2 * i] - st[32 + 2 * i]); prev = st[33 + 2 * i]; }
            printf("\n");
        }
    }
    // check: L L^T == A and Linv L == I (one more run that stores L)
    (void)hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    k_chol_diag<<<1, 64 * CDW, NB * DL * 8>>>(dA, n, 0, dLinv, dflag, 1, gate_none(dflag));
    std::vector<double> L(n * n), Li(n * n);
    (void)hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(Li.data(), dLinv, n * n * 8, hipMemcpyDeviceToHost);
    double e1m = 0, e2m = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0, s2 = 0;
            for (int k = 0; k <= j; ++k) s += L[i * n + k] * L[j * n + k];
            for (int k = j; k <= i; ++k) s2 += Li[i * n + k] * L[k * n + j];
            e1m = fmax(e1m, fabs(s - A[i * n + j])); e2m = fmax(e2m, fabs(s2 - (i == j)));
        }
    printf("max |LL^T - A| = %.2e, max |Linv L - I| = %.2e\n", e1m, e2m);
    return 0;
}
