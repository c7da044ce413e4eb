/*
 * desc_oracle.c -- CPU restatement of the reference's descriptor producer step.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as the other oracle files).  PARITY UNPINNED: the reference holds no
 * fixture for this step and its network (TorchScript SuperPoint) is absent; the arithmetic restated here is
 * the reference's own C++ (no third-party code involved):
 *
 *   FeatureSuperPoint.cpp:183-211  processDescriptors: cell = (keypoint.x / 8, keypoint.y / 8) in integer
 *                                  arithmetic, the first 256 channels of descriptors[0][cell.y][cell.x],
 *                                  every element divided by FeatDesc::norm()
 *   datatypes.h:59-67              FeatDesc::norm: `sum += descElem * descElem` with float elements and a double
 *                                  sum -- the product is an fp32 product, widened afterwards -- then sqrt
 *   :204                           `desc[idx] /= descNorm`: float / double, evaluated in double, rounded to float
 *
 * map is addressed by element strides (channel, row, column): the network hands over [1][256][H/8][W/8] and the
 * reference reads it through a permuted view (:253), i.e. stride_c = Hc*Wc, stride_y = Wc, stride_x = 1.
 */
#include <math.h>
#include <stdint.h>

void orc_desc_sample(const float *map, int64_t sc, int64_t sy, int64_t sx, const int32_t *kp_xy, int K, int D, float *out)
{
    for (int k = 0; k < K; ++k) {
        const int xc = kp_xy[2 * k] / 8, yc = kp_xy[2 * k + 1] / 8;
        const float *base = map + (int64_t)yc * sy + (int64_t)xc * sx;
        double sum = 0.0;
        for (int c = 0; c < D; ++c) {
            const float v = base[(int64_t)c * sc];
            const float p = v * v;              /* -ffp-contract=off: an fp32 product, as `descElem * descElem` is */
            sum += (double)p;
        }
        const double norm = sqrt(sum);
        for (int c = 0; c < D; ++c) out[(int64_t)k * D + c] = (float)((double)base[(int64_t)c * sc] / norm);
    }
}
