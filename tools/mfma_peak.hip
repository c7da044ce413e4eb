// tools/mfma_peak.hip -- diagnostic: sustained MFMA rates on RANDOM operands (operands in registers, no memory
// traffic), i.e. the clock-limited ceilings the coarse kernel's roofline fraction should be read against:
//   v_mfma_f32_32x32x16_f16, v_mfma_f32_16x16x32_f16 (same output tile per wave: 4 vs 16 accumulators of 16 / 4 registers)
//   and the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (unit scales).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_f16_32(const half8 *__restrict__ in, float *__restrict__ out, int iters)
{
    half8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_f16_16(const half8 *__restrict__ in, float *__restrict__ out, int iters)
{
    half8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fp8_scaled(const i32x8 *__restrict__ in, float *__restrict__ out, int iters)
{
    i32x8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int one = 0x7f7f7f7f;        // E8M0 scale 2^0 in every byte
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], 0 /*A: fp8 e4m3*/, 0 /*B: fp8 e4m3*/, 0, one, 0, one);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    std::vector<_Float16> h(512 * 8);
    std::vector<unsigned char> h8(512 * 32);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
    for (auto &v : h8) {                       // random e4m3 bit patterns of small magnitude (exponent field <= 7), both signs
        const unsigned r = (unsigned)rand();
        v = (unsigned char)(((r & 1) << 7) | (((r >> 1) % 8) << 3) | ((r >> 8) & 7));
    }
    half8 *din; i32x8 *din8; float *dout;
    hipMalloc(&din, h.size() * 2);
    hipMalloc(&din8, h8.size());
    hipMalloc(&dout, 4096 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(din8, h8.data(), h8.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; ++which)
        for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
            const int blocks = 256 * blocks_per_cu, iters = 40000 / blocks_per_cu;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) k_f16_32<<<blocks, 256>>>(din, dout, iters);
                else if (which == 1) k_f16_16<<<blocks, 256>>>(din, dout, iters / 1);
                else k_fp8_scaled<<<blocks, 256>>>(din8, dout, iters / 2);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // flop per wave-iteration: 4 x 32x32x16 | 16 x 16x16x32 | 4 x 32x32x64
                const double per = which == 0 ? 4.0 * 2 * 32 * 32 * 16 : which == 1 ? 16.0 * 2 * 16 * 16 * 32 : 4.0 * 2 * 32 * 32 * 64;
                const double flops = (double)blocks * 4 * (which == 2 ? iters / 2 : iters) * per;
                printf("%-28s waves/SIMD=%d rep=%d  %7.2f ms  %5.0f TFLOP/s\n",
                       which == 0 ? "f16 32x32x16" : which == 1 ? "f16 16x16x32" : "fp8 e4m3 scaled 32x32x64", blocks_per_cu, rep, ms, flops / ms / 1e9);
            }
        }
    return 0;
}
