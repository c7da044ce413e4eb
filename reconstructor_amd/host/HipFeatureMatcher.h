// HipFeatureMatcher.h -- drop-in FeatureMatcher plugin backed by librcn.so (include/rcn.h).
//
// `FeatureMatcher` is the reference's abstract base verbatim in shape (FeatureMatcher.h:11-27);
// `HipL2Matcher` takes the place of `FlannMatcher` (FeatureMatcher.h:32-46): a maintainer adds
// one enum value + one `case` in SequentialReconstructor's constructor switch
// (SequentialReconstructor.cpp:31-41) -- see INTEGRATION.md.
// Re-entrant: the reference shares one matcher between 4 OpenMP threads
// (SequentialReconstructor.cpp:202,232); the plugin serialises its callers.
//
// Per-image device cache.  The reference's loop hands the SAME images to the plugin again and again --
// `auto features1 = features[imgId1]` copies the vector of shared pointers for every pair
// (SequentialReconstructor.cpp:213-214), so image i crosses this boundary N - 1 times -- and FlannMatcher
// re-packs both images on every call (featDescToCV, FeatureMatcher.cpp:11-25).  Here an image is recognised by
// the Feature objects its shared pointers name (first, last, count, descriptor length) plus a signature over the
// address of EVERY row's descriptor storage and the contents of up to 64 evenly spaced rows -- a feature vector that
// was freed and re-detected into recycled Feature objects still moves its descriptor allocations or their sampled
// contents -- packed and uploaded ONCE, and kept resident in the ctx (fp32 rows, fp16 copy, norms) under an id of
// the plugin's own range; a pair call then is one rcn_match_grid over two resident ids.  Least recently used images
// are replaced beyond `capacity` images.  Descriptors are written once by the detector and never edited afterwards
// in the reference; a caller that edits rows IN PLACE (same storage, rows outside the sample) MUST call invalidate().
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>
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
        // one descriptor length per call (featDescToCV would build two matrices knnMatch then refuses)
        if (features1[0]->featDesc.desc.size() != features2[0]->featDesc.desc.size()) throw std::runtime_error("descriptor lengths differ");
        std::lock_guard<std::mutex> lk(mu_);                  // the cache and the pair of ids it hands out belong to one caller at a time
        const int32_t pr[2] = {resident(features1), resident(features2)};
        std::vector<int32_t> out(features1.size(), -1);
        int32_t count = 0;
        int rc = pr[0] == pr[1] ? RCN_ERR_ARG : rcn_match_grid(ctx_, pr, 1, ratioThresh, out.data(), (int64_t)features1.size(), &count);
        if (pr[0] == pr[1]) {
            // an image against itself (the loop never asks for it): not a grid pair of two residents -- the uncached call
            int D1 = 0, D2 = 0;
            const std::vector<float> q = featDescToDense(features1, D1), t = featDescToDense(features2, D2);
            if (D1 != D2) throw std::runtime_error("descriptor lengths differ");
            rc = rcn_match_pair(ctx_, q.data(), (int32_t)features1.size(), t.data(), (int32_t)features2.size(), D1, ratioThresh, out.data(), &count);
        }
        if (rc != RCN_OK) throw std::runtime_error(std::string("HipL2Matcher::matchFeatures: ") + rcn_last_error(ctx_));
        for (size_t i = 0; i < out.size(); ++i)
            if (out[i] >= 0) matches[(int)i] = out[i];
    }
    // The same call without the cache -- pack both images, upload both, match, on every call (what round 2 shipped; kept
    // for the latency comparison of tests/cpp/plugin_loop_test.cpp and for callers whose descriptors change between calls)
    void matchFeaturesUncached(const std::vector<FeaturePtr<>> &features1, const std::vector<FeaturePtr<>> &features2, std::map<int, int> &matches)
    {
        if (features1.empty() || features2.empty()) return;
        int D1 = 0, D2 = 0;
        const std::vector<float> q = featDescToDense(features1, D1), t = featDescToDense(features2, D2);
        if (D1 != D2) throw std::runtime_error("descriptor lengths differ");
        std::vector<int32_t> out(features1.size(), -1);
        int32_t count = 0;
        std::lock_guard<std::mutex> lk(mu_);
        dropOwn();                                                             // rcn_match_pair needs the ctx's D to itself (only the plugin's own ids go)
        const int rc = rcn_match_pair(ctx_, q.data(), (int32_t)features1.size(), t.data(), (int32_t)features2.size(), D1, ratioThresh, out.data(), &count);
        if (rc != RCN_OK) throw std::runtime_error(std::string("rcn_match_pair: ") + rcn_last_error(ctx_));
        for (size_t i = 0; i < out.size(); ++i)
            if (out[i] >= 0) matches[(int)i] = out[i];
    }
    // forget every cached image (descriptors were edited in place, or the ctx's descriptors were cleared behind the plugin's back)
    void invalidate()
    {
        std::lock_guard<std::mutex> lk(mu_);
        cache_.clear();
    }
    void setCapacity(size_t images) { std::lock_guard<std::mutex> lk(mu_); capacity_ = images < 2 ? 2 : images; }
    size_t uploads() const { return uploads_.load(); }        // images packed + uploaded so far (a 25-image loop: 25, not 600)
    rcn_ctx *context() { return ctx_; }

private:
    struct Entry {
        const void *first, *last;
        size_t K;
        int D;
        uint64_t sig;
        int32_t id;
        uint64_t stamp;
    };
    static uint64_t signature(const std::vector<FeaturePtr<>> &f)
    {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](uint64_t v) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; };
        for (const FeaturePtr<> &p : f) {                       // where every row lives, and how long it is
            mix(reinterpret_cast<uintptr_t>(p->featDesc.desc.data()));
            mix(p->featDesc.desc.size());
        }
        const size_t n = f.size(), step = n > 64 ? n / 64 : 1;  // what up to 64 evenly spaced rows (first and last among them) hold
        auto row = [&](size_t r) { for (float v : f[r]->featDesc.desc) { uint32_t u; std::memcpy(&u, &v, 4); mix(u); } };
        for (size_t r = 0; r < n; r += step) row(r);
        row(n - 1);
        return h;
    }
    // the plugin's own images out of the ctx (a shared ctx keeps everybody else's descriptors)
    void dropOwn()
    {
        for (const Entry &e : cache_)
            if (rcn_desc_remove(ctx_, e.id) != RCN_OK) throw std::runtime_error(std::string("rcn_desc_remove: ") + rcn_last_error(ctx_));
        cache_.clear();
    }
    // id under which the image is resident in the ctx, uploading it first when it is not (mu_ held)
    int32_t resident(const std::vector<FeaturePtr<>> &f)
    {
        const int D = (int)f[0]->featDesc.desc.size();
        const uint64_t sig = signature(f);
        if (!cache_.empty() && cache_[0].D != D) dropOwn();   // another descriptor kind: the ctx holds one D at a time.  Only the plugin's
                                                              // own ids go; if other users of a shared ctx still hold the old length, the upload
                                                              // below fails with "all resident images must share D" instead of wiping them
        for (Entry &e : cache_)
            if (e.first == f.front().get() && e.last == f.back().get() && e.K == f.size() && e.D == D && e.sig == sig) {
                e.stamp = ++clock_;
                return e.id;
            }
        Entry e{f.front().get(), f.back().get(), f.size(), D, sig, 0, ++clock_};
        if (cache_.size() < capacity_) {
            e.id = kIdBase + (int32_t)cache_.size();
            cache_.push_back(e);
        } else {
            size_t lru = 0;
            for (size_t i = 1; i < cache_.size(); ++i)
                if (cache_[i].stamp < cache_[lru].stamp) lru = i;
            e.id = cache_[lru].id;                            // re-uploading an id replaces the image it named
            cache_[lru] = e;
        }
        int Dd = 0;
        const std::vector<float> dense = featDescToDense(f, Dd);
        if (rcn_desc_upload(ctx_, e.id, dense.data(), (int32_t)f.size(), D) != RCN_OK) {
            for (size_t i = 0; i < cache_.size(); ++i)
                if (cache_[i].id == e.id) { cache_.erase(cache_.begin() + i); break; }
            throw std::runtime_error(std::string("rcn_desc_upload: ") + rcn_last_error(ctx_));
        }
        ++uploads_;
        return e.id;
    }
    static constexpr int32_t kIdBase = 0x40000000;     // the plugin's own id range inside a shared ctx
    rcn_ctx *ctx_ = nullptr;
    bool owned_ = true;
    std::mutex mu_;
    std::vector<Entry> cache_;
    size_t capacity_ = 64;
    std::atomic<size_t> uploads_{0};
    uint64_t clock_ = 0;
    const float ratioThresh = 0.7;   // FeatureMatcher.h:45
};

}  // namespace reconstructor::Core
