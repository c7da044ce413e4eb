// desc.hip -- the producer side of the descriptor contract: rows written straight into the device layout
// the matcher ingests ([n][K][D] fp32, rcn_desc_upload_batch_device / the slot of rcn_shard_reserve).
//
// rcn_desc_sample_device restates FeatureSuperPoint's processDescriptors (FeatureSuperPoint.cpp:183-211) --
// the step that turns the network's dense descriptor map into one 256-d unit-norm row per keypoint -- on
// the GPU, so that the rows never visit the host (the reference builds a std::vector<float> per keypoint and
// featDescToCV re-gathers them for every pair):
//     cell   = (keypoint.x / 8, keypoint.y / 8)                  integer division           :191-192
//     v[c]   = map[0][cell.y][cell.x][c],  c < 256                                          :195
//     norm   = sqrt(sum_c (double)(v[c] * v[c]))   product in fp32, sum in fp64, ascending c   datatypes.h:59-67
//     row[c] = (float)((double)v[c] / norm)                                                 :201-205
// Bit-exact against oracle/desc_oracle.c: the products are rounded to fp32 before they are widened, and one
// lane adds them in ascending order.
#include "rcn_internal.h"

namespace {

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_desc_sample(const float *__restrict__ map, long long sc, long long sy, long long sx,
                                                     int Hc, int Wc, const int32_t *__restrict__ kp, int K, int D,
                                                     float *__restrict__ out)
{
    __shared__ float sq[4][256 + 8];
    __shared__ double nrm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + w;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float *base = nullptr;
    if (k < K) {
        const int xc = kp[2 * k] / 8, yc = kp[2 * k + 1] / 8;
        base = map + (long long)yc * sy + (long long)xc * sx;
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < D) v[i] = base[(long long)c * sc];
            sq[w][c] = c < D ? __fmul_rn(v[i], v[i]) : 0.f;
        }
    }
    __syncthreads();
    if (k < K && lane == 0) {
        double sum = 0.0;
        for (int c = 0; c < D; ++c) sum = __dadd_rn(sum, (double)sq[w][c]);
        nrm[w] = sqrt(sum);
    }
    __syncthreads();
    if (k < K) {
        const double n = nrm[w];
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < D) out[(size_t)k * D + c] = (float)((double)v[i] / n);
        }
    }
}

}  // namespace

extern "C" int rcn_desc_sample_device(rcn_ctx *ctx, const float *desc_map_dev, int64_t stride_c, int64_t stride_y, int64_t stride_x,
                                      int32_t Hc, int32_t Wc, const int32_t *kp_xy_dev, int32_t K, int32_t D, float *out_rows_dev)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (K < 0 || D < 1 || D > 256 || Hc < 1 || Wc < 1 || (K > 0 && (!desc_map_dev || !kp_xy_dev || !out_rows_dev))) {
        ctx->set_error("rcn_desc_sample_device: bad argument (1 <= D <= 256)");
        return RCN_ERR_ARG;
    }
    if (K == 0) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    k_desc_sample<<<(K + 3) / 4, 256, 0, ctx->stream>>>(desc_map_dev, stride_c, stride_y, stride_x, Hc, Wc, kp_xy_dev, K, D, out_rows_dev);
    RCN_HIP(hipGetLastError());
    return RCN_OK;
}
