// HipPairGridDriver.h -- the reference's pair loop on every GPU of the node, from C++.
//
// Takes the place of the body of SequentialReconstructor::matchFeatures
// (SequentialReconstructor.cpp:199-279) for the FakeImgMatcher grid (every i != j, ImageMatcher.cpp:6-23):
// one host thread per GPU, each owning one rcn_ctx + one rcn_shard (RCCL communicators over xGMI);
// nothing here touches HIP or RCCL directly -- only include/rcn.h.
//
//   images   0 .. n-1 (the loop indexes `features` by position, :204-214), rank r packs and uploads the
//            block it owns (featDescToCV's gather, FeatureMatcher.cpp:11-25, once per image, ragged K)
//   exchange rcn_shard_exchange: RCCL all-reduce of the scale statistics, all-gather of the fp16 payload
//   match    rcn_shard_match: pair number p of the canonical i < j list runs on rank p % world
//   results  featureMatches[(i,j)][query feature] = train feature and the inverted map under (j,i), exactly
//            what the loop stores when it meets (j,i) after (i,j) (:219-227).  A pair whose forward
//            matching stored NOTHING has no entry to invert: the loop then matches (j,i) in its own right
//            (query = j), and so does this driver, in a second pass over those pairs.
//   filter   matchFeatures(features, featureMatches, true) also runs the geometric filter of :237-269 on every pair
//            with at least 7 matches (rcn_shard_filter: the match table never leaves HBM); the recommended binding
//            for the whole loop.
// Failure: a rank whose local step fails still enters rcn_shard_exchange, whose status vote ends the collective
// phase on every rank together (include/rcn.h, rcn_shard_fail); the driver then throws -- it cannot hang.  The library's
// own waits are bounded as well (rcn_shard_set_timeout): a rank that died costs its peers a timeout, not a hang.
#pragma once
#include <algorithm>
#include <atomic>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <unordered_map>
#include <memory>

#include "../../include/rcn.h"
#include "rcn_types.h"

namespace reconstructor::Core {

// key hash of featureMatches (the reference declares its own pair_hash, SequentialReconstructor.h:51-63;
// any hash of the pair serves -- the container's contents do not depend on it)
struct pair_hash {
    std::size_t operator()(const std::pair<int, int> &p) const
    {
        return std::hash<long long>()(((long long)p.first << 32) ^ (unsigned)p.second);
    }
};
using FeatureMatches = std::unordered_map<std::pair<int, int>, std::unordered_map<int, int>, pair_hash>;

class HipPairGridDriver {
public:
    // nDevices = 0: every visible GPU
    explicit HipPairGridDriver(int nDevices = 0)
    {
        const int have = rcn_device_count();
        world_ = nDevices > 0 ? nDevices : have;
        if (world_ < 1 || world_ > have) throw std::runtime_error("HipPairGridDriver: no usable gfx950 device");
        uint8_t id[RCN_SHARD_ID_BYTES];
        if (rcn_shard_unique_id(id) != RCN_OK) throw std::runtime_error("HipPairGridDriver: rcn_shard_unique_id failed");
        ctx_.assign(world_, nullptr);
        shard_.assign(world_, nullptr);
        // Two phases, joined in between: every ctx first (local: a device that cannot be had is known here, and then NO rank
        // goes on), only then rcn_shard_create, which is collective -- a rank that skipped it would leave the others inside
        // ncclCommInitRank for ever.
        std::vector<std::string> err(world_);
        run([&](int r) { if (rcn_create(r, &ctx_[r]) != RCN_OK) err[r] = "rcn_create failed on device " + std::to_string(r); });
        for (const auto &e : err)
            if (!e.empty()) { release(); throw std::runtime_error("HipPairGridDriver: " + e); }
        run([&](int r) {
            if (rcn_shard_create(ctx_[r], r, world_, id, &shard_[r]) != RCN_OK) err[r] = std::string("rcn_shard_create: ") + rcn_last_error(ctx_[r]);
        });
        for (const auto &e : err)
            if (!e.empty()) { release(); throw std::runtime_error("HipPairGridDriver: " + e); }
    }
    ~HipPairGridDriver() { release(); }
    HipPairGridDriver(const HipPairGridDriver &) = delete;
    HipPairGridDriver &operator=(const HipPairGridDriver &) = delete;

    int world() const { return world_; }
    rcn_ctx *context(int rank = 0) { return ctx_[rank]; }

    // The features / matches cache the reference lists as a TODO (README.md:39): everything matchFeatures consumed and
    // produced, in one checksummed file (rcn_store_save: written to <path>.tmp, then renamed).  Pure host code.
    static void saveStore(const std::string &path, std::unordered_map<int, std::vector<FeaturePtr<>>> &features, const FeatureMatches &featureMatches)
    {
        const int n = (int)features.size();
        std::vector<int32_t> ids(n), Ks(n);
        std::vector<std::vector<float>> desc(n);
        std::vector<std::vector<int32_t>> xy(n);
        std::vector<const float *> dp(n);
        std::vector<const int32_t *> cp(n);
        int D = 0;
        for (int i = 0; i < n; ++i) {
            const auto &f = features.at(i);
            ids[i] = i; Ks[i] = (int32_t)f.size();
            if (!f.empty()) D = (int)f[0]->featDesc.desc.size();
            desc[i].reserve(f.size() * (size_t)D);
            xy[i].reserve(2 * f.size());
            for (const auto &ft : f) {
                desc[i].insert(desc[i].end(), ft->featDesc.desc.begin(), ft->featDesc.desc.end());
                xy[i].push_back(ft->featCoord.x); xy[i].push_back(ft->featCoord.y);
            }
            dp[i] = desc[i].data(); cp[i] = xy[i].data();
        }
        std::vector<std::pair<int, int>> keys;
        for (const auto &kv : featureMatches) keys.push_back(kv.first);
        std::sort(keys.begin(), keys.end());
        std::vector<int32_t> pairs, qt;
        std::vector<int64_t> offs(1, 0);
        for (const auto &k : keys) {
            const auto &m = featureMatches.at(k);
            std::vector<std::pair<int, int>> e(m.begin(), m.end());
            std::sort(e.begin(), e.end());
            pairs.push_back(k.first); pairs.push_back(k.second);
            for (const auto &p : e) { qt.push_back(p.first); qt.push_back(p.second); }
            offs.push_back((int64_t)qt.size() / 2);
        }
        rcn_store_contents c{};
        c.n_images = n; c.D = D; c.has_coords = 1; c.n_pairs = (int32_t)keys.size();
        c.img_ids = ids.data(); c.img_K = Ks.data(); c.desc = dp.data(); c.coords = cp.data();
        c.pairs = pairs.data(); c.offsets = offs.data(); c.qt = qt.data();
        if (rcn_store_save(path.c_str(), &c) != RCN_OK) throw std::runtime_error("HipPairGridDriver::saveStore: rcn_store_save failed for " + path);
    }
    // ... and back: the containers the rest of the reference's pipeline reads, as matchFeatures left them
    static void loadStore(const std::string &path, std::unordered_map<int, std::vector<FeaturePtr<>>> &features, FeatureMatches &featureMatches)
    {
        rcn_store *st = nullptr;
        if (rcn_store_open(path.c_str(), &st) != RCN_OK) throw std::runtime_error("HipPairGridDriver::loadStore: cannot open or verify " + path);
        rcn_store_contents c{};
        if (rcn_store_contents_of(st, &c) != RCN_OK) { rcn_store_close(st); throw std::runtime_error("HipPairGridDriver::loadStore: no contents"); }
        features.clear(); featureMatches.clear();
        for (int i = 0; i < c.n_images; ++i) {
            auto &f = features[c.img_ids[i]];
            for (int k = 0; k < c.img_K[i]; ++k) {
                const float *row = c.desc[i] + (size_t)k * c.D;
                const FeatCoord<> xy = c.has_coords ? FeatCoord<>(c.coords[i][2 * k], c.coords[i][2 * k + 1]) : FeatCoord<>();
                f.push_back(std::make_shared<Feature<>>(xy, FeatDesc(row, row + c.D)));
            }
        }
        for (int p = 0; p < c.n_pairs; ++p) {
            auto &m = featureMatches[{c.pairs[2 * p], c.pairs[2 * p + 1]}];
            for (int64_t e = c.offsets[p]; e < c.offsets[p + 1]; ++e) m[c.qt[2 * e]] = c.qt[2 * e + 1];
        }
        rcn_store_close(st);
    }

    // features[imgId] for imgId = 0 .. n-1 (SequentialReconstructor.h:205); fills featureMatches (:226).
    // filter = the argument of SequentialReconstructor::matchFeatures(bool filter): pairs with at least 7 matches keep
    // only the inliers of the fundamental-matrix search (:237-269, rcn_shard_filter on the table in HBM); a pair for
    // which no model is found stores nothing and is therefore matched again the other way round, like a pair without
    // matches.
    void matchFeatures(std::unordered_map<int, std::vector<FeaturePtr<>>> &features, FeatureMatches &featureMatches, bool filter = false)
    {
        const int n = (int)features.size();
        if (n < 2) return;
        // everything that can be checked without a GPU is checked before a thread is started: no rank may walk away
        // from a collective because of something every rank could have known
        int Kmax = 0, D = 0;
        for (int i = 0; i < n; ++i) {
            auto it = features.find(i);
            if (it == features.end()) throw std::runtime_error("HipPairGridDriver: image ids must be 0 .. n-1");
            Kmax = std::max(Kmax, (int)it->second.size());
            for (const auto &f : it->second) {
                const int d = (int)f->featDesc.desc.size();
                if (D && d != D) throw std::runtime_error("HipPairGridDriver: descriptor lengths differ");
                D = d;
            }
        }
        if (Kmax == 0 || D == 0) return;
        struct RankOut {
            std::vector<int32_t> pairs, qt, rev_pairs, rev_out, rev_cnt;
            std::vector<int64_t> offs;
            std::string err;
        };
        std::vector<RankOut> out(world_);
        run([&](int r) {
            RankOut &o = out[r];
            rcn_shard *sh = shard_[r];
            auto fail = [&](const char *what) { o.err = std::string(what) + ": " + rcn_last_error(ctx_[r]); };
            // Local phase.  A failure here is this rank's alone: it is recorded (the library remembers its own, the
            // driver reports the rest through rcn_shard_fail) and the rank STILL enters rcn_shard_exchange, whose
            // status vote makes every rank leave the collective phase together with an error.
            bool local_ok = rcn_shard_reserve(sh, n, Kmax, D, nullptr) == RCN_OK;
            if (!local_ok) fail("rcn_shard_reserve");
            int32_t lo = 0, cnt = 0;
            rcn_shard_owned_images(n, world_, r, &lo, &cnt);
            std::vector<float> dense;
            for (int img = lo; local_ok && img < lo + cnt; ++img) {
                const auto &f = features[img];
                dense.resize(f.size() * (size_t)D);
                for (size_t k = 0; k < f.size(); ++k) {
                    const std::vector<float> &d = f[k]->featDesc.desc;
                    std::copy(d.begin(), d.end(), dense.begin() + k * D);
                }
                if (rcn_shard_put_image(sh, img, dense.data(), (int32_t)f.size()) != RCN_OK) { fail("rcn_shard_put_image"); local_ok = false; }
            }
            if (local_ok && filter) {
                // integer pixel coordinates of every image of the grid (featuresToCvPoints, utils.cpp:165-177): 8 bytes
                // per keypoint, to every rank -- a pair's train image may be anybody's
                std::vector<int32_t> xy;
                for (int img = 0; local_ok && img < n; ++img) {
                    const auto &f = features[img];
                    xy.resize(2 * f.size() + 2);
                    for (size_t k = 0; k < f.size(); ++k) { xy[2 * k] = f[k]->featCoord.x; xy[2 * k + 1] = f[k]->featCoord.y; }
                    if (rcn_coords_upload(ctx_[r], img, xy.data(), (int32_t)f.size()) != RCN_OK) { fail("rcn_coords_upload"); local_ok = false; }
                }
            }
            if (!local_ok) rcn_shard_fail(sh, RCN_ERR_ARG);
            // Collective phase: the one call every rank makes whatever happened above.
            if (rcn_shard_exchange(sh, nullptr, nullptr) != RCN_OK) { if (o.err.empty()) fail("rcn_shard_exchange"); return; }
            // From here on nothing is collective: a failing rank simply reports after the join.
            if (rcn_shard_match(sh, ratioThresh, nullptr, 0, nullptr) != RCN_OK) { fail("rcn_shard_match"); return; }
            if (filter && rcn_shard_filter(sh, nullptr) != RCN_OK) { fail("rcn_shard_filter"); return; }
            const int64_t P = rcn_shard_pair_count(n, world_, r);
            o.pairs.resize(2 * (size_t)P);
            rcn_shard_pairs(n, world_, r, o.pairs.data());
            o.offs.assign((size_t)P + 1, 0);
            int64_t total = 0;
            int rc = rcn_shard_lists(sh, o.offs.data(), nullptr, 0, &total);
            if (rc != RCN_OK && total == 0) { fail("rcn_shard_lists"); return; }
            o.qt.resize(2 * (size_t)total);
            if (total > 0 && rcn_shard_lists(sh, o.offs.data(), o.qt.data(), total, &total) != RCN_OK) { fail("rcn_shard_lists"); return; }
            // second pass: pairs that stored nothing are matched again the other way round (:219-232)
            for (int64_t p = 0; p < P; ++p)
                if (o.offs[p + 1] == o.offs[p]) { o.rev_pairs.push_back(o.pairs[2 * p + 1]); o.rev_pairs.push_back(o.pairs[2 * p]); }
            const int32_t R = (int32_t)(o.rev_pairs.size() / 2);
            if (R > 0) {
                o.rev_out.assign((size_t)R * Kmax, -1);
                o.rev_cnt.assign(R, 0);
                if (rcn_match_grid_filtered(ctx_[r], o.rev_pairs.data(), R, ratioThresh, filter ? 1 : 0, o.rev_out.data(), Kmax, o.rev_cnt.data(), nullptr) != RCN_OK) {
                    fail("rcn_match_grid_filtered");
                    return;
                }
            }
        });
        for (const auto &o : out)
            if (!o.err.empty()) throw std::runtime_error("HipPairGridDriver: " + o.err);
        for (const RankOut &o : out) {
            const size_t P = o.pairs.size() / 2;
            for (size_t p = 0; p < P; ++p) {
                if (o.offs[p + 1] == o.offs[p]) continue;
                const std::pair<int, int> cur(o.pairs[2 * p], o.pairs[2 * p + 1]), inv(cur.second, cur.first);
                auto &fwd = featureMatches[cur];
                auto &bwd = featureMatches[inv];
                for (int64_t e = o.offs[p]; e < o.offs[p + 1]; ++e) {
                    fwd[o.qt[2 * e]] = o.qt[2 * e + 1];
                    bwd[o.qt[2 * e + 1]] = o.qt[2 * e];
                }
            }
            const size_t R = o.rev_pairs.size() / 2;
            for (size_t p = 0; p < R; ++p) {
                if (o.rev_cnt[p] == 0) continue;
                auto &m = featureMatches[{o.rev_pairs[2 * p], o.rev_pairs[2 * p + 1]}];
                const int32_t *row = o.rev_out.data() + p * (size_t)Kmax;
                for (int q = 0; q < Kmax; ++q)
                    if (row[q] >= 0) m[q] = row[q];
            }
        }
    }

    // Test hook: the next matchFeatures reports a failure of its own on this rank in front of the exchange.
    void injectLocalFailure(int rank) { if (rank >= 0 && rank < world_) rcn_shard_fail(shard_[rank], RCN_ERR_ARG); }

private:
    template <class F> void run(F &&body)
    {
        if (world_ == 1) { body(0); return; }
        std::vector<std::thread> th;
        for (int r = 0; r < world_; ++r) th.emplace_back([&, r] { body(r); });
        for (auto &t : th) t.join();
    }
    void release()
    {
        // communicator teardown is collective as well
        run([&](int r) {
            if (shard_[r]) rcn_shard_destroy(shard_[r]);
            if (ctx_[r]) rcn_destroy(ctx_[r]);
            shard_[r] = nullptr; ctx_[r] = nullptr;
        });
    }
    int world_ = 0;
    std::vector<rcn_ctx *> ctx_;
    std::vector<rcn_shard *> shard_;
    const float ratioThresh = 0.7;   // FeatureMatcher.h:45
};

}  // namespace reconstructor::Core
