// tools/gemm_nt_bench.hip -- diagnostic: the bulk trailing-update kernel on a synthetic 10112^2 system (panels from column block 0).
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
    // the balanced tile map of the first bulk update of a right-looking factorisation: every tile (i, j), 2 <= j <= i < nblk
    chol::Params prm;
    prm.nblk = nblk; prm.tl_g = 0; prm.pair = 0;
    const chol::Plan pl = chol::make_plan(prm);
    const chol::Op *bulk = nullptr;
    for (const chol::Op &o : pl.ops)
        if (o.kind == chol::UPD_PIPE && o.stream == chol::ST_C) { bulk = &o; break; }
    if (!bulk) { printf("no bulk update in the plan\n"); return 1; }
    unsigned *map;
    (void)hipMalloc(&map, bulk->map_n * sizeof(unsigned));
    (void)hipMemcpy(map, pl.maps.data() + bulk->map_off, bulk->map_n * sizeof(unsigned), hipMemcpyHostToDevice);
    const int pgrid = bulk->map_n;
    printf("  map: %d workgroups for %d tiles (the busiest XCD carries %d)\n", pgrid, mt * (mt + 1) / 2, pgrid / 8);
#define PSET(D, N) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt_pipe<D, N>), hipFuncAttributeMaxDynamicSharedMemorySize, GST * GSTAGE_BYTES)
    PSET(0, 16); PSET(4, 16); PSET(7, 16); PSET(0, 32); PSET(4, 32); PSET(7, 32); PSET(1, 16); PSET(2, 16); PSET(3, 16); PSET(1, 32); PSET(2, 32); PSET(3, 32); PSET(0, 0); PSET(7, 0); PSET(3, 0);
#define PRUN(D, N, nst, what)                                                                                              \
    for (int rep = 0; rep < 5; ++rep) {                                                                                    \
        float m;                                                                                                           \
        (void)hipEventRecord(e0);                                                                                          \
        k_gemm_nt_pipe<D, N><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S0, L, npad, 0, map, 0, nullptr, nullptr, 0, nst, 0);       \
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&m, e0, e1);                     \
        if (rep == 4) printf("  pipe K = %4d %s: %.1f us (%.1f TF)   %s\n", 8 * nst, what, m * 1e3, flop * (nst / 16.0) / m * 1e-9, hipGetErrorString(hipGetLastError())); \
    }
    PRUN(0, 16, 16, "full") PRUN(4, 16, 16, "no operand DMA") PRUN(7, 16, 16, "loop only")
    PRUN(0, 32, 32, "full") PRUN(4, 32, 32, "no operand DMA") PRUN(7, 32, 32, "loop only")
    PRUN(1, 16, 16, "no C loads") PRUN(2, 16, 16, "no C stores") PRUN(3, 16, 16, "no C at all")
    PRUN(1, 32, 32, "no C loads") PRUN(2, 32, 32, "no C stores") PRUN(3, 32, 32, "no C at all") PRUN(0, 32, 32, "full again")
    PRUN(0, 0, 32, "rolled, full") PRUN(0, 0, 48, "rolled, full") PRUN(0, 0, 64, "rolled, full") PRUN(7, 0, 64, "rolled, loop only") PRUN(3, 0, 64, "rolled, no C at all")
    PRUN(0, 0, 96, "rolled, full") PRUN(0, 0, 128, "rolled, full") PRUN(7, 0, 128, "rolled, loop only") PRUN(0, 32, 32, "full once more")
    // correctness of the pipelined form on one tile, against a non-zero C
    for (int nst : {16, 32, 48, 64, 128}) {
        std::vector<double> c0((size_t)128 * npad);
        for (size_t i = 0; i < c0.size(); ++i) c0[i] = (double)((i * 2654435761u) % 1000) * 1e-3;
        (void)hipMemset(S1, 0, N * 8);
        (void)hipMemcpy(S1 + (size_t)5 * 128 * npad, c0.data(), c0.size() * 8, hipMemcpyHostToDevice);
        if (nst == 16) k_gemm_nt_pipe<0, 16><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, map, 0, nullptr, nullptr, 0, 16, 0);
        else if (nst == 32) k_gemm_nt_pipe<0, 32><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, map, 0, nullptr, nullptr, 0, 32, 0);
        else k_gemm_nt_pipe<0, 0><<<pgrid, 256, GST * GSTAGE_BYTES>>>(S1, L, npad, 0, map, 0, nullptr, nullptr, 0, nst, 0);
        std::vector<double> c2((size_t)128 * npad);
        (void)hipMemcpy(c2.data(), S1 + (size_t)5 * 128 * npad, c2.size() * 8, hipMemcpyDeviceToHost);
        double md2 = 0;
        for (int i = 0; i < 128; ++i)
            for (int j = 0; j < 128; ++j) {
                double sref = c0[(size_t)i * npad + 3 * 128 + j];
                for (int k = 0; k < 8 * nst; ++k) sref -= h[((size_t)5 * 128 + i) * npad + k] * h[((size_t)3 * 128 + j) * npad + k];
                md2 = fmax(md2, fabs(sref - c2[(size_t)i * npad + 3 * 128 + j]));
            }
        printf("max |pipe(K = %d) - host| on tile (5,3) = %.3e\n", 8 * nst, md2);
    }
    return 0;
}
