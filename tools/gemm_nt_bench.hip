// tools/gemm_nt_bench.hip -- diagnostic: the bulk trailing-update kernel on a synthetic 10112^2 system (kb = 0).
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
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flop = 2.0 * 128 * 128 * 128 * (double)mt * (mt + 1) / 2;
    float ms0 = 0.f; (void)ms0;
    // the pipelined form (k_gemm_nt_pipe) over the balanced tile map of this trailing size
    {
        rcn_ctx *bc = new rcn_ctx();
        if (build_bulk_maps(bc, nblk) != RCN_OK) { printf("map build failed\n"); return 1; }
        const unsigned *map = bc->bulk_map.as<unsigned>() + bc->bulk_map_off[mt];
        const int pgrid = bc->bulk_map_grid[mt];
        printf("  map: %d workgroups for %d tiles (the busiest XCD carries %d)\n", pgrid, mt * (mt + 1) / 2, pgrid / 8);
#define PSET(D, N) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<D, N>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES)
        PSET(0, 16); PSET(4, 16); PSET(7, 16); PSET(0, 32); PSET(4, 32); PSET(7, 32); PSET(1, 16); PSET(2, 16); PSET(3, 16); PSET(1, 32); PSET(2, 32); PSET(3, 32); PSET(0, 48); PSET(0, 64); PSET(7, 64);
#define PRUN(D, N, what)                                                                                                   \
        for (int rep = 0; rep < 5; ++rep) {                                                                                \
            float m;                                                                                                       \
            (void)hipEventRecord(e0);                                                                                      \
            k_gemm_nt_pipe<D, N><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S0, L, npad, 0, map, 2, nullptr);                        \
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&m, e0, e1);                 \
            if (rep == 4) printf("  pipe K = %4d %s: %.1f us (%.1f TF)   %s\n", 8 * N, what, m * 1e3, flop * (N / 16.0) / m * 1e-9, hipGetErrorString(hipGetLastError())); \
        }
        PRUN(0, 16, "full") PRUN(4, 16, "no operand DMA") PRUN(7, 16, "loop only")
        PRUN(0, 32, "full") PRUN(4, 32, "no operand DMA") PRUN(7, 32, "loop only")
        PRUN(1, 16, "no C loads") PRUN(2, 16, "no C stores") PRUN(3, 16, "no C at all")
        PRUN(1, 32, "no C loads") PRUN(2, 32, "no C stores") PRUN(3, 32, "no C at all") PRUN(0, 32, "full again")
        PRUN(0, 48, "full") PRUN(0, 64, "full") PRUN(7, 64, "loop only") PRUN(0, 32, "full once more")
        // correctness of the pipelined form on one tile, against a non-zero C, for both pass lengths
        for (int two = 0; two < 4; ++two) {
            std::vector<double> c0((size_t)128 * npad);
            for (size_t i = 0; i < c0.size(); ++i) c0[i] = (double)((i * 2654435761u) % 1000) * 1e-3;
            (void)hipMemset(S1, 0, N * 8);
            (void)hipMemcpy(S1 + (size_t)5 * 128 * npad, c0.data(), c0.size() * 8, hipMemcpyHostToDevice);
            if (two == 3) k_gemm_nt_pipe<0, 64><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, map, 2, nullptr);
            else if (two == 2) k_gemm_nt_pipe<0, 48><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, map, 2, nullptr);
            else if (two) k_gemm_nt_pipe<0, 32><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, map, 2, nullptr);
            else k_gemm_nt_pipe<0, 16><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, map, 2, nullptr);
            std::vector<double> c2((size_t)128 * npad);
            (void)hipMemcpy(c2.data(), S1 + (size_t)5 * 128 * npad, c2.size() * 8, hipMemcpyDeviceToHost);
            double md2 = 0;
            for (int i = 0; i < 128; ++i)
                for (int j = 0; j < 128; ++j) {
                    double sref = c0[(size_t)i * npad + 3 * 128 + j];
                    for (int k = 0; k < 128 * (two + 1); ++k) sref -= h[((size_t)5 * 128 + i) * npad + k] * h[((size_t)3 * 128 + j) * npad + k];
                    md2 = fmax(md2, fabs(sref - c2[(size_t)i * npad + 3 * 128 + j]));
                }
            printf("max |pipe(K = %d) - host| on tile (5,3) = %.3e\n", 128 * (two + 1), md2);
        }
    }
    return 0;
}
