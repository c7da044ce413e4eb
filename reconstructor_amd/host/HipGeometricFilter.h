// HipGeometricFilter.h -- GeometricFilter::estimateFundamental with the reference's signature
// (GeometricFilter.h:33-35, GeometricFilter.cpp:39-61), the RANSAC / LMedS search handed to
// rcn_fmat_filter (include/rcn.h) instead of cv::findFundamentalMat.
// Return value: Eigen::Matrix3d in the reference; Mat3d here (M(r,c)).  Zero when no model was found
// (GeometricFilter.cpp:50-53), else the winning 7-point hypothesis -- OpenCV returns an 8-point refit
// on the inliers, which no caller of the reference reads (SequentialReconstructor.cpp:250).
#pragma once
#include <stdexcept>
#include <string>

#include "../../include/rcn.h"
#include "rcn_types.h"

namespace reconstructor::Core {

struct Mat3d {
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double &operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
};

class GeometricFilter {
public:
    explicit GeometricFilter(rcn_ctx *ctx = nullptr) : ctx_(ctx), owned_(false)
    {
        if (!ctx_) {
            if (rcn_create(0, &ctx_) != RCN_OK) throw std::runtime_error("GeometricFilter: no usable gfx950 device");
            owned_ = true;
        }
    }
    ~GeometricFilter() { if (owned_) rcn_destroy(ctx_); }
    GeometricFilter(const GeometricFilter &) = delete;
    GeometricFilter &operator=(const GeometricFilter &) = delete;

    // inlierMatchIds is appended to, one flag per match, exactly when a model was found
    // (writeInliersToVector, utils.cpp:328-343); it stays empty otherwise and the caller drops the pair.
    Mat3d estimateFundamental(const std::vector<FeaturePtr<>> &features1, const std::vector<FeaturePtr<>> &features2,
                              std::vector<bool> &inlierMatchIds)
    {
        const size_t n = features1.size();
        if (features2.size() != n) throw std::invalid_argument("estimateFundamental: one feature of image 2 per feature of image 1");
        std::vector<int32_t> xy1(2 * n + 2), xy2(2 * n + 2);
        for (size_t i = 0; i < n; ++i) {      // featuresToCvPoints, utils.cpp:165-177
            xy1[2 * i] = features1[i]->featCoord.x; xy1[2 * i + 1] = features1[i]->featCoord.y;
            xy2[2 * i] = features2[i]->featCoord.x; xy2[2 * i + 1] = features2[i]->featCoord.y;
        }
        std::vector<uint8_t> mask(n + 1);
        int32_t count = 0;
        Mat3d F;
        if (rcn_fmat_filter(ctx_, xy1.data(), xy2.data(), (int32_t)n, mask.data(), &count, F.m) != RCN_OK)
            throw std::runtime_error(std::string("estimateFundamental: ") + rcn_last_error(ctx_));
        if (count == -1 || count == -2) return Mat3d();       // cv::findFundamentalMat returns an empty matrix below 7 points too
        for (size_t i = 0; i < n; ++i) inlierMatchIds.push_back(mask[i] != 0);
        return F;
    }

private:
    rcn_ctx *ctx_;
    bool owned_;
};

}  // namespace reconstructor::Core
