// HipFeatureMatcher.h -- drop-in FeatureMatcher plugin backed by librcn.so (include/rcn.h).
//
// `FeatureMatcher` is the reference's abstract base verbatim in shape (FeatureMatcher.h:11-27);
// `HipL2Matcher` takes the place of `FlannMatcher` (FeatureMatcher.h:32-46): a maintainer adds
// one enum value + one `case` in SequentialReconstructor's constructor switch
// (SequentialReconstructor.cpp:31-41) -- see INTEGRATION.md.
// Re-entrant: the reference shares one matcher between 4 OpenMP threads
// (SequentialReconstructor.cpp:202,232); the ctx serialises calls internally.
#pragma once
#include <stdexcept>
#include <string>

#include "../../include/rcn.h"
#include "rcn_types.h"

namespace reconstructor::Core {

class FeatureMatcher {
public:
    FeatureMatcher(const bool /*featNormalization*/ = false) {}
    virtual void matchFeatures(const std::vector<FeaturePtr<>> &features1,
                               const std::vector<FeaturePtr<>> &features2,
                               std::map<int, int> &matches,
                               const std::pair<int, int> imgShape1,
                               const std::pair<int, int> imgShape2) = 0;
    virtual ~FeatureMatcher() {}
};

// dense row-major K x D floats: what featDescToCV builds (FeatureMatcher.cpp:11-25)
inline std::vector<float> featDescToDense(const std::vector<FeaturePtr<>> &features, int &D)
{
    D = features.empty() ? 0 : (int)features[0]->featDesc.desc.size();
    std::vector<float> dense((size_t)features.size() * D);
    for (size_t i = 0; i < features.size(); ++i) {
        const std::vector<float> &d = features[i]->featDesc.desc;
        if ((int)d.size() != D) throw std::runtime_error("descriptor lengths differ");
        std::copy(d.begin(), d.end(), dense.begin() + i * D);
    }
    return dense;
}

class HipL2Matcher : public FeatureMatcher {
public:
    explicit HipL2Matcher(int device = 0) : owned_(true)
    {
        if (rcn_create(device, &ctx_) != RCN_OK) throw std::runtime_error("HipL2Matcher: no usable gfx950 device");
    }
    // The documented binding: ONE rcn_ctx per GPU shared by every plugin of the pipeline (matcher, geometric filter,
    // bundle adjuster, validity sweep) -- workspaces, streams and resident descriptors are per ctx, so nothing is
    // allocated or torn down per plugin call.  The ctx outlives the plugin.
    explicit HipL2Matcher(rcn_ctx *shared) : ctx_(shared), owned_(false)
    {
        if (!ctx_) throw std::runtime_error("HipL2Matcher: null ctx");
    }
    ~HipL2Matcher() override { if (owned_) rcn_destroy(ctx_); }
    HipL2Matcher(const HipL2Matcher &) = delete;
    HipL2Matcher &operator=(const HipL2Matcher &) = delete;

    void matchFeatures(const std::vector<FeaturePtr<>> &features1,
                       const std::vector<FeaturePtr<>> &features2, std::map<int, int> &matches,
                       const std::pair<int, int> /*imgShape1*/, const std::pair<int, int> /*imgShape2*/) override
    {
        if (features1.empty() || features2.empty()) return;   // the reference asserts here (:39)
        int D1 = 0, D2 = 0;
        const std::vector<float> q = featDescToDense(features1, D1), t = featDescToDense(features2, D2);
        if (D1 != D2) throw std::runtime_error("descriptor lengths differ");
        std::vector<int32_t> out(features1.size(), -1);
        int32_t count = 0;
        const int rc = rcn_match_pair(ctx_, q.data(), (int32_t)features1.size(), t.data(), (int32_t)features2.size(),
                                      D1, ratioThresh, out.data(), &count);
        if (rc != RCN_OK) throw std::runtime_error(std::string("rcn_match_pair: ") + rcn_last_error(ctx_));
        for (size_t i = 0; i < out.size(); ++i)
            if (out[i] >= 0) matches[(int)i] = out[i];
    }
    rcn_ctx *context() { return ctx_; }

private:
    rcn_ctx *ctx_ = nullptr;
    bool owned_ = true;
    const float ratioThresh = 0.7;   // FeatureMatcher.h:45
};

}  // namespace reconstructor::Core
