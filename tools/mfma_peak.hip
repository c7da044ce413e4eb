// tools/mfma_peak.hip -- diagnostic: sustained v_mfma_f32_32x32x16_f16 rate on random operands
// (the DVFS-limited ceiling that the coarse kernel's roofline fraction should be read against).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(const half8 *__restrict__ in, float *__restrict__ out, int iters)
{
    half8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    std::vector<_Float16> h(512 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
    half8 *din; float *dout;
    hipMalloc(&din, h.size() * 2);
    hipMalloc(&dout, 4096 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int blocks = 256 * blocks_per_cu, iters = 40000 / blocks_per_cu;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k_mfma<4><<<blocks, 256>>>(din, dout, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * 2 * 32 * 32 * 16;
            printf("waves/SIMD=%d rep=%d  %.2f ms  %.0f TFLOP/s\n", blocks_per_cu, rep, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
