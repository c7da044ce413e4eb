// HipLandmarkValidity.h -- SequentialReconstructor::checkLandmarkValidity and ::removeOutlierLandmarks
// (SequentialReconstructor.cpp:869-954, :956-976) over the reference's own containers, with the
// sweep itself handed to rcn_landmark_validity (include/rcn.h).  In the reference these are private
// members working on `features`, `landmarks`, `imgIdx2camPose`, `imgIdx2camIntrinsics`; here the
// same containers are passed in.  Pose matrices: anything indexable as M(r,c).
#pragma once
#include <stdexcept>
#include <string>

#include "../../include/rcn.h"
#include "rcn_types.h"

namespace reconstructor::Core {

class LandmarkValidator {
public:
    explicit LandmarkValidator(rcn_ctx *ctx = nullptr) : ctx_(ctx), owned_(false)
    {
        if (!ctx_) {
            if (rcn_create(0, &ctx_) != RCN_OK) throw std::runtime_error("LandmarkValidator: no usable gfx950 device");
            owned_ = true;
        }
    }
    ~LandmarkValidator() { if (owned_) rcn_destroy(ctx_); }
    LandmarkValidator(const LandmarkValidator &) = delete;
    LandmarkValidator &operator=(const LandmarkValidator &) = delete;

    double maxProjectionError = 4.0;      // SequentialReconstructor.h:256
    double minTriangulationAngle = 1.0;   // SequentialReconstructor.h:257

    // Returns the inlier flags and erases the rejected observations from
    // landmark.triangulatedFeatures, exactly what the reference's member function does.
    template <class Pose4>
    std::vector<bool> checkLandmarkValidity(std::unordered_map<int, std::vector<FeaturePtr<>>> &features,
                                            std::vector<Landmark> &landmarks,
                                            std::unordered_map<int, Pose4> &imgIdx2camPose,
                                            std::unordered_map<int, PinholeCamera> &imgIdx2camIntrinsics)
    {
        std::unordered_map<int, int> local;            // image index -> row in the flat camera arrays
        std::vector<double> poses, intr, pts(3 * landmarks.size());
        std::vector<int32_t> ptOff(landmarks.size() + 1, 0), obsCam, obsXY;
        for (size_t j = 0; j < landmarks.size(); ++j) {
            const Landmark &lm = landmarks[j];
            pts[3 * j] = lm.x; pts[3 * j + 1] = lm.y; pts[3 * j + 2] = lm.z;
            for (const auto &tf : lm.triangulatedFeatures) {
                auto it = local.find(tf.imgIdx);
                if (it == local.end()) {
                    it = local.emplace(tf.imgIdx, (int)local.size()).first;
                    const Pose4 &T = imgIdx2camPose.at(tf.imgIdx);
                    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) poses.push_back(T(r, c));
                    const PinholeCamera &cam = imgIdx2camIntrinsics.at(tf.imgIdx);
                    const double k[6] = {cam.fX, cam.fY, cam.cX, cam.cY, cam.k1, cam.k2};
                    intr.insert(intr.end(), k, k + 6);
                }
                const auto &feat = features.at(tf.imgIdx).at(tf.featIdx);
                obsCam.push_back(it->second);
                obsXY.push_back(feat->featCoord.x); obsXY.push_back(feat->featCoord.y);
            }
            ptOff[j + 1] = (int32_t)obsCam.size();
        }
        std::vector<uint8_t> inl(landmarks.size() + 1), keep(obsCam.size() + 1);
        rcn_landmark_problem pb = {(int32_t)local.size(), (int32_t)landmarks.size(), (int32_t)obsCam.size(), 0,
                                   poses.data(), intr.data(), pts.data(), ptOff.data(), obsCam.data(), obsXY.data()};
        int32_t nIn = 0;
        if (rcn_landmark_validity(ctx_, &pb, maxProjectionError, minTriangulationAngle, inl.data(), keep.data(), &nIn) != RCN_OK)
            throw std::runtime_error(std::string("checkLandmarkValidity: ") + rcn_last_error(ctx_));
        std::vector<bool> inlierLandmarks(landmarks.size());
        for (size_t j = 0; j < landmarks.size(); ++j) {
            auto &tfs = landmarks[j].triangulatedFeatures;
            std::vector<TriangulatedFeature> left;
            for (size_t i = 0; i < tfs.size(); ++i)
                if (keep[ptOff[j] + i]) left.push_back(tfs[i]);
            tfs.swap(left);
            inlierLandmarks[j] = inl[j] != 0;
        }
        return inlierLandmarks;
    }

    // SequentialReconstructor.cpp:956-976: keep the inlier landmarks in order; the observations still
    // attached to a dropped landmark become free again (landmarkId = -1).
    static void removeOutlierLandmarks(std::unordered_map<int, std::vector<FeaturePtr<>>> &features,
                                       std::vector<Landmark> &landmarks, const std::vector<bool> &inlierIds)
    {
        size_t w = 0;
        for (size_t j = 0; j < landmarks.size() && j < inlierIds.size(); ++j) {
            if (!inlierIds[j]) {
                for (const TriangulatedFeature &tf : landmarks[j].triangulatedFeatures)
                    features.at(tf.imgIdx).at(tf.featIdx)->landmarkId = -1;
                continue;
            }
            if (w != j) landmarks[w] = std::move(landmarks[j]);
            ++w;
        }
        landmarks.resize(w);
    }

private:
    rcn_ctx *ctx_;
    bool owned_;
};

}  // namespace reconstructor::Core
