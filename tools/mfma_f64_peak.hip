// tools/mfma_f64_peak.hip -- diagnostic: the sustained rate of v_mfma_f64_16x16x4_f64 on random operands held in
// registers (no memory traffic), at 1 and 2 waves per SIMD: the clock-limited ceiling the Cholesky's bulk update and
// its roofline fraction should be read against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_f64(const double *__restrict__ in, double *__restrict__ out, int iters)
{
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[64 * i + (threadIdx.x & 63)]; b[i] = in[256 + 64 * i + (threadIdx.x & 63)]; }
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)        // inline asm: hipcc parks loop-carried f64 accumulators in AGPRs and copies all 128 registers every trip
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i >> 2]));
    }
    double s = 0.0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    std::vector<double> h(512);
    srand(1);
    for (auto &v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
    double *din, *dout;
    (void)hipMalloc(&din, h.size() * 8); (void)hipMalloc(&dout, 4096 * 256 * 8);
    (void)hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int blocks = 256 * bpc, iters = 20000 / bpc;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            k_f64<<<blocks, 256>>>(din, dout, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 16.0 * 2 * 16 * 16 * 4;
            printf("f64 16x16x4  waves/SIMD=%d rep=%d  %7.2f ms  %5.1f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", bpc, rep, ms, flops / ms / 1e9,
                   ms * 1e-3 * 2.4e9 / ((double)iters * 16 * bpc));
        }
    }
    return 0;
}
