// HipBundleSession.h -- BundleAdjuster::adjust (BundleAdjuster.h:80-84) over a device-resident session.
//
// The reference calls adjust once per registered view (SequentialReconstructor.cpp:1040-1094) and rebuilds the whole
// ceres::Problem from its containers every time; HipBundleAdjuster.h does the same with rcn_ba_solve.  This adapter
// keeps the problem in HBM between the calls (rcn_ba_session_*, include/rcn.h): a call sends only what the pipeline
// added since the previous one -- the new view, the new landmarks, the observations push_back'ed onto existing
// tracks -- and re-sends every camera's 12 numbers (the pipeline owns the 4x4 pose matrices: they are re-packed each
// call exactly as BundleAdjuster.cpp:46-59 does, so the unpack quirk of :157-185 round-trips the same way).
// Same argument list, same in-place update, same return value, and the same numbers bit for bit as
// HipBundleAdjuster.h on the same containers (tests/cpp/session_adapter_test.cpp).
// If the containers changed in any other way (a landmark or an observation erased by removeOutlierLandmarks, a
// landmark moved, the view order permuted) the session is rebuilt from the containers; rebuilds() counts that.
#pragma once
#include <stdexcept>
#include <string>

#include "HipBundleAdjuster.h"

namespace reconstructor::Core {

class IncrementalBundleAdjuster {
public:
    explicit IncrementalBundleAdjuster(rcn_ctx *ctx = nullptr) : ctx_(ctx), owned_(false)
    {
        if (!ctx_) {
            if (rcn_create(0, &ctx_) != RCN_OK) throw std::runtime_error("IncrementalBundleAdjuster: no usable gfx950 device");
            owned_ = true;
        }
    }
    ~IncrementalBundleAdjuster()
    {
        if (session_) rcn_ba_session_destroy(session_);
        if (owned_) rcn_destroy(ctx_);
    }
    IncrementalBundleAdjuster(const IncrementalBundleAdjuster &) = delete;
    IncrementalBundleAdjuster &operator=(const IncrementalBundleAdjuster &) = delete;

    template <class Pose4>
    std::unordered_map<int, int> adjust(std::unordered_map<int, std::vector<FeaturePtr<>>> &features,
                                        std::vector<Landmark> &landmarks,
                                        std::unordered_map<int, Pose4> &imgIdx2camPose,
                                        std::unordered_map<int, PinholeCamera> &imgIdx2camIntrinsics,
                                        std::vector<int> imgIdxOrder)
    {
        const int nCams = (int)imgIdxOrder.size();
        if (!session_ || !extends(landmarks, imgIdxOrder)) reset();
        std::unordered_map<int, int> global2local;
        for (int l = 0; l < nCams; ++l) global2local[imgIdxOrder[l]] = l;

        // cameras: every call re-packs all of them from the pipeline's matrices (BundleAdjuster.cpp:34-63)
        std::vector<double> extr(6 * (size_t)nCams), intr(6 * (size_t)nCams);
        for (int l = 0; l < nCams; ++l) {
            const int g = imgIdxOrder[l];
            const PinholeCamera &K = imgIdx2camIntrinsics[g];
            const double k6[6] = {K.fX, K.fY, K.cX, K.cY, K.k1, K.k2};
            std::copy(k6, k6 + 6, intr.begin() + 6 * l);
            const Pose4 &T = imgIdx2camPose[g];
            detail::rotationToAngleAxis(T, &extr[6 * l]);
            for (int i = 0; i < 3; ++i) extr[6 * l + 3 + i] = T(i, 3);
        }
        for (int l = (int)order_.size(); l < nCams; ++l) {
            int32_t idx = -1;
            check(rcn_ba_session_add_camera(session_, &extr[6 * l], &intr[6 * l], &idx), "rcn_ba_session_add_camera");
            order_.push_back(imgIdxOrder[l]);
        }
        check(rcn_ba_session_cameras(session_, nullptr, nullptr, extr.data(), intr.data()), "rcn_ba_session_cameras");

        // new landmarks, then the new tail of every track (push_back order)
        const size_t known = tracks_.size();
        if (landmarks.size() > known) {
            std::vector<double> xyz(3 * (landmarks.size() - known));
            for (size_t j = known; j < landmarks.size(); ++j) {
                xyz[3 * (j - known)] = landmarks[j].x; xyz[3 * (j - known) + 1] = landmarks[j].y; xyz[3 * (j - known) + 2] = landmarks[j].z;
            }
            int32_t first = -1;
            check(rcn_ba_session_add_points(session_, (int32_t)(landmarks.size() - known), xyz.data(), &first), "rcn_ba_session_add_points");
            tracks_.resize(landmarks.size());
        }
        std::vector<int32_t> pt, cam, xy;
        for (size_t j = 0; j < landmarks.size(); ++j) {
            const auto &tf = landmarks[j].triangulatedFeatures;
            for (size_t k = tracks_[j].size(); k < tf.size(); ++k) {
                const FeaturePtr<> &f = features[tf[k].imgIdx][tf[k].featIdx];
                pt.push_back((int32_t)j);
                cam.push_back(global2local[tf[k].imgIdx]);
                xy.push_back((int32_t)f->featCoord.x);
                xy.push_back((int32_t)f->featCoord.y);
                tracks_[j].emplace_back(tf[k].imgIdx, tf[k].featIdx);
            }
        }
        if (!pt.empty())
            check(rcn_ba_session_add_observations(session_, (int32_t)pt.size(), pt.data(), cam.data(), xy.data()), "rcn_ba_session_add_observations");

        const int rc = rcn_ba_session_solve(session_, nullptr, &summary);       // default options: BundleAdjuster.cpp:99-142
        if (rc != RCN_OK && rc != RCN_ERR_NUMERIC)                              // the reference ignores Ceres' summary (:145-147)
            throw std::runtime_error(std::string("rcn_ba_session_solve: ") + rcn_last_error(ctx_));

        // write back (BundleAdjuster.cpp:149-187)
        xyz_.resize(3 * landmarks.size());
        check(rcn_ba_session_read_points(session_, xyz_.data()), "rcn_ba_session_read_points");
        for (size_t j = 0; j < landmarks.size(); ++j) { landmarks[j].x = xyz_[3 * j]; landmarks[j].y = xyz_[3 * j + 1]; landmarks[j].z = xyz_[3 * j + 2]; }
        check(rcn_ba_session_cameras(session_, extr.data(), intr.data(), nullptr, nullptr), "rcn_ba_session_cameras");
        for (int l = 0; l < nCams; ++l) {
            const int g = imgIdxOrder[l];
            Pose4 T = imgIdx2camPose[g];
            detail::angleAxisToPose(&extr[6 * l], T);
            imgIdx2camPose[g] = T;
            PinholeCamera &K = imgIdx2camIntrinsics[g];
            K.fX = intr[6 * l]; K.fY = intr[6 * l + 1]; K.cX = intr[6 * l + 2]; K.cY = intr[6 * l + 3];
            K.k1 = intr[6 * l + 4]; K.k2 = intr[6 * l + 5];
        }
        return global2local;
    }

    rcn_ba_summary summary{};
    int rebuilds() const { return rebuilds_; }

private:
    // the containers extend what the session holds: same view order so far, same landmarks at the same coordinates,
    // every known track a prefix of the landmark's triangulatedFeatures
    bool extends(const std::vector<Landmark> &landmarks, const std::vector<int> &imgIdxOrder) const
    {
        if (imgIdxOrder.size() < order_.size() || landmarks.size() < tracks_.size()) return false;
        for (size_t l = 0; l < order_.size(); ++l)
            if (imgIdxOrder[l] != order_[l]) return false;
        for (size_t j = 0; j < tracks_.size(); ++j) {
            const auto &tf = landmarks[j].triangulatedFeatures;
            if (tf.size() < tracks_[j].size()) return false;
            if (landmarks[j].x != xyz_[3 * j] || landmarks[j].y != xyz_[3 * j + 1] || landmarks[j].z != xyz_[3 * j + 2]) return false;
            for (size_t k = 0; k < tracks_[j].size(); ++k)
                if (tf[k].imgIdx != tracks_[j][k].first || tf[k].featIdx != tracks_[j][k].second) return false;
        }
        return true;
    }
    void reset()
    {
        if (session_) { rcn_ba_session_destroy(session_); session_ = nullptr; ++rebuilds_; }
        if (rcn_ba_session_create(ctx_, &session_) != RCN_OK)
            throw std::runtime_error(std::string("rcn_ba_session_create: ") + rcn_last_error(ctx_));
        order_.clear(); tracks_.clear(); xyz_.clear();
    }
    void check(int rc, const char *what) const
    {
        if (rc != RCN_OK) throw std::runtime_error(std::string(what) + ": " + rcn_last_error(ctx_));
    }

    rcn_ctx *ctx_;
    bool owned_;
    rcn_ba_session *session_ = nullptr;
    std::vector<int> order_;                                    // imgIdxOrder as the session knows it
    std::vector<std::vector<std::pair<int, int>>> tracks_;      // (imgIdx, featIdx) per landmark, in track order
    std::vector<double> xyz_;                                   // landmark coordinates as written back by the last adjust
    int rebuilds_ = 0;
};

}  // namespace reconstructor::Core
