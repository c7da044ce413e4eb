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
                                                     float *__restrict__ out, unsigned *__restrict__ n_bad)
{
    __shared__ float sq[4][256 + 8];
    __shared__ double nrm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + w;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float *base = nullptr;
    bool inside = false;
    if (k < K) {
        // a keypoint outside the map (the reference's tensor indexing would throw there): nothing is read, the row
        // comes out as zeros and the call is counted in *n_bad (rcn_desc_sample_errors)
        const int x = kp[2 * k], y = kp[2 * k + 1];
        const int xc = x / 8, yc = y / 8;
        inside = x >= 0 && y >= 0 && xc < Wc && yc < Hc;
        if (!inside && lane == 0) atomicAdd(n_bad, 1u);
    }
    if (inside) {
        const int xc = kp[2 * k] / 8, yc = kp[2 * k + 1] / 8;
        base = map + (long long)yc * sy + (long long)xc * sx;
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < D) v[i] = base[(long long)c * sc];
            sq[w][c] = c < D ? __fmul_rn(v[i], v[i]) : 0.f;
        }
    }
    __syncthreads();
    if (inside && lane == 0) {
        double sum = 0.0;
        for (int c = 0; c < D; ++c) sum = __dadd_rn(sum, (double)sq[w][c]);
        nrm[w] = sqrt(sum);
    }
    __syncthreads();
    if (k < K) {
        const double n = inside ? nrm[w] : 1.0;
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < D) out[(size_t)k * D + c] = inside ? (float)((double)v[i] / n) : 0.f;
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
    if (!ctx->desc_bad.p) {
        RCN_HIP(ctx->desc_bad.reserve(sizeof(unsigned)));
        RCN_HIP(hipMemsetAsync(ctx->desc_bad.p, 0, sizeof(unsigned), ctx->stream));
    }
    k_desc_sample<<<(K + 3) / 4, 256, 0, ctx->stream>>>(desc_map_dev, stride_c, stride_y, stride_x, Hc, Wc, kp_xy_dev, K, D, out_rows_dev,
                                                        ctx->desc_bad.as<unsigned>());
    RCN_HIP(hipGetLastError());
    return RCN_OK;
}

// Keypoints that fell outside their descriptor map in the rcn_desc_sample_device calls since the last read (their rows
// are zeros).  Waits for the ctx stream; RCN_ERR_ARG when there were any.
extern "C" int rcn_desc_sample_errors(rcn_ctx *ctx, int32_t *n_out_of_range)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    unsigned n = 0;
    if (ctx->desc_bad.p) {
        RCN_HIP(hipSetDevice(ctx->device));
        RCN_HIP(hipMemcpyAsync(&n, ctx->desc_bad.p, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
        RCN_HIP(hipMemsetAsync(ctx->desc_bad.p, 0, sizeof(unsigned), ctx->stream));
        RCN_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (n_out_of_range) *n_out_of_range = (int32_t)n;
    if (n) { ctx->set_error("rcn_desc_sample_device: " + std::to_string(n) + " keypoint(s) outside the descriptor map (rows zeroed)"); return RCN_ERR_ARG; }
    return RCN_OK;
}
