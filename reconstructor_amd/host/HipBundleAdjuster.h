// HipBundleAdjuster.h -- BundleAdjuster::adjust with the reference's signature
// (BundleAdjuster.h:80-84), packing / unpacking as BundleAdjuster.cpp:17-70 and :149-187 do and
// handing the solve to rcn_ba_solve (include/rcn.h) instead of ceres::Solve (:145-146).
// Pose matrices: anything indexable as M(r,c) (Eigen::Matrix4d in the reference, Mat4d here).
#pragma once
#include <stdexcept>
#include <string>

#include "../../include/rcn.h"
#include "rcn_types.h"

namespace reconstructor::Core {

namespace detail {
// Eigen::AngleAxisd::fromRotationMatrix goes through a quaternion; same route here.
template <class M> inline void rotationToAngleAxis(const M &T, double w[3])
{
    const double r00 = T(0, 0), r11 = T(1, 1), r22 = T(2, 2), tr = r00 + r11 + r22;
    double q[4];   // w, x, y, z
    if (tr > 0) {
        double s = std::sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (T(2, 1) - T(1, 2)) / s; q[2] = (T(0, 2) - T(2, 0)) / s; q[3] = (T(1, 0) - T(0, 1)) / s;
    } else if (r00 > r11 && r00 > r22) {
        double s = std::sqrt(1.0 + r00 - r11 - r22) * 2;
        q[0] = (T(2, 1) - T(1, 2)) / s; q[1] = 0.25 * s; q[2] = (T(0, 1) + T(1, 0)) / s; q[3] = (T(0, 2) + T(2, 0)) / s;
    } else if (r11 > r22) {
        double s = std::sqrt(1.0 + r11 - r00 - r22) * 2;
        q[0] = (T(0, 2) - T(2, 0)) / s; q[1] = (T(0, 1) + T(1, 0)) / s; q[2] = 0.25 * s; q[3] = (T(1, 2) + T(2, 1)) / s;
    } else {
        double s = std::sqrt(1.0 + r22 - r00 - r11) * 2;
        q[0] = (T(1, 0) - T(0, 1)) / s; q[1] = (T(0, 2) + T(2, 0)) / s; q[2] = (T(1, 2) + T(2, 1)) / s; q[3] = 0.25 * s;
    }
    if (q[0] < 0) for (double &v : q) v = -v;
    const double n = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < 1e-300) { w[0] = w[1] = w[2] = 0; return; }
    const double angle = 2.0 * std::atan2(n, q[0]);
    for (int i = 0; i < 3; ++i) w[i] = q[1 + i] / n * angle;
}
// BundleAdjuster.cpp:157-185: angle-axis + translation back into the 4x4 pose; the axis is divided by (angle + 1e-6),
// i.e. it is not exactly unit length -- restated as the reference has it.
template <class M> inline void angleAxisToPose(const double *w6, M &T)
{
    const double *w = w6;
    const double angle = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double x = w[0] / (angle + 1e-6), y = w[1] / (angle + 1e-6), z = w[2] / (angle + 1e-6);
    const double c = std::cos(angle), s = std::sin(angle), t = 1 - c;
    T(0, 0) = t * x * x + c;     T(0, 1) = t * x * y - s * z; T(0, 2) = t * x * z + s * y;
    T(1, 0) = t * x * y + s * z; T(1, 1) = t * y * y + c;     T(1, 2) = t * y * z - s * x;
    T(2, 0) = t * x * z - s * y; T(2, 1) = t * y * z + s * x; T(2, 2) = t * z * z + c;
    for (int i = 0; i < 3; ++i) { T(i, 3) = w6[3 + i]; T(3, i) = 0; }
    T(3, 3) = 1;
}
}  // namespace detail

class BundleAdjuster {
public:
    explicit BundleAdjuster(rcn_ctx *ctx = nullptr) : ctx_(ctx), owned_(false)
    {
        if (!ctx_) {
            if (rcn_create(0, &ctx_) != RCN_OK) throw std::runtime_error("BundleAdjuster: no usable gfx950 device");
            owned_ = true;
        }
    }
    ~BundleAdjuster() { if (owned_) rcn_destroy(ctx_); }

    template <class Pose4>
    std::unordered_map<int, int> adjust(std::unordered_map<int, std::vector<FeaturePtr<>>> &features,
                                        std::vector<Landmark> &landmarks,
                                        std::unordered_map<int, Pose4> &imgIdx2camPose,
                                        std::unordered_map<int, PinholeCamera> &imgIdx2camIntrinsics,
                                        std::vector<int> imgIdxOrder)
    {
        const int nCams = (int)imgIdxOrder.size();
        std::vector<double> extr(6 * (size_t)nCams), intr(6 * (size_t)nCams), pts(3 * landmarks.size());
        std::unordered_map<int, int> global2local;
        for (int l = 0; l < nCams; ++l) {                                   // BundleAdjuster.cpp:34-63
            const int g = imgIdxOrder[l];
            const PinholeCamera &K = imgIdx2camIntrinsics[g];
            const double k6[6] = {K.fX, K.fY, K.cX, K.cY, K.k1, K.k2};
            std::copy(k6, k6 + 6, intr.begin() + 6 * l);
            const Pose4 &T = imgIdx2camPose[g];
            detail::rotationToAngleAxis(T, &extr[6 * l]);
            for (int i = 0; i < 3; ++i) extr[6 * l + 3 + i] = T(i, 3);
            global2local[g] = l;
        }
        std::vector<double> uv;
        std::vector<int32_t> cam, pt;
        for (size_t j = 0; j < landmarks.size(); ++j) {                     // :65-97, landmark-major
            pts[3 * j] = landmarks[j].x; pts[3 * j + 1] = landmarks[j].y; pts[3 * j + 2] = landmarks[j].z;
            for (const TriangulatedFeature &tf : landmarks[j].triangulatedFeatures) {
                const FeaturePtr<> &f = features[tf.imgIdx][tf.featIdx];
                uv.push_back((double)f->featCoord.x);
                uv.push_back((double)f->featCoord.y);
                cam.push_back(global2local[tf.imgIdx]);
                pt.push_back((int32_t)j);
            }
        }
        rcn_ba_problem pb{nCams, (int32_t)landmarks.size(), (int32_t)cam.size(), 0, extr.data(), intr.data(),
                          pts.data(), uv.data(), cam.data(), pt.data()};
        rcn_ba_options opt;
        rcn_ba_default_options(nCams, &opt);                                // :99-142
        const int rc = rcn_ba_solve(ctx_, &pb, &opt, &summary);
        if (rc != RCN_OK && rc != RCN_ERR_NUMERIC)                          // the reference ignores Ceres' summary (:145-147)
            throw std::runtime_error(std::string("rcn_ba_solve: ") + rcn_last_error(ctx_));
        for (size_t j = 0; j < landmarks.size(); ++j) {                     // :150-155
            landmarks[j].x = pts[3 * j]; landmarks[j].y = pts[3 * j + 1]; landmarks[j].z = pts[3 * j + 2];
        }
        for (int l = 0; l < nCams; ++l) {                                   // :157-185
            const int g = imgIdxOrder[l];
            Pose4 T = imgIdx2camPose[g];
            detail::angleAxisToPose(&extr[6 * l], T);
            imgIdx2camPose[g] = T;
            PinholeCamera &K = imgIdx2camIntrinsics[g];
            K.fX = intr[6 * l]; K.fY = intr[6 * l + 1]; K.cX = intr[6 * l + 2]; K.cY = intr[6 * l + 3];
            K.k1 = intr[6 * l + 4]; K.k2 = intr[6 * l + 5];
        }
        return global2local;                                                // :187
    }
    rcn_ba_summary summary{};

private:
    rcn_ctx *ctx_;
    bool owned_;
};

}  // namespace reconstructor::Core
