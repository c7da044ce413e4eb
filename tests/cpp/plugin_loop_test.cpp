// plugin_loop_test.cpp -- the reference's pair loop driven through the UNMODIFIED plugin call
// (featMatcher->matchFeatures(features1, features2, curMatches, shape1, shape2), SequentialReconstructor.cpp:202-232):
// every ordered pair (i, j), i != j, handed over as fresh copies of the two feature vectors, from 4 host threads as the
// reference's OpenMP loop does.  Measures the per-call time with and without the plugin's per-image device cache and
// writes every pair's map for comparison with the oracle.
// Input: i32 n, i32 D, then per image: i32 K, K*D floats.  Output: i32 n_pairs, then per ordered pair (i, j):
//   i32 i, i32 j, i32 count, count x (i32 query, i32 train); then f64 ms_per_call_cached, f64 ms_per_call_uncached, i64 uploads.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../reconstructor_amd/host/HipFeatureMatcher.h"

using namespace reconstructor::Core;

static void rd(FILE *f, void *p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    FILE *o = fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int32_t n, D;
    rd(f, &n, 4); rd(f, &D, 4);
    std::unordered_map<int, std::vector<FeaturePtr<>>> features;
    for (int i = 0; i < n; ++i) {
        int32_t K;
        rd(f, &K, 4);
        std::vector<float> rows((size_t)K * D);
        rd(f, rows.data(), 4 * rows.size());
        features[i] = {};
        for (int k = 0; k < K; ++k)
            features[i].push_back(std::make_shared<Feature<>>(FeatCoord<>(k, k), FeatDesc(rows.begin() + (size_t)k * D, rows.begin() + (size_t)(k + 1) * D)));
    }
    try {
        HipL2Matcher matcher;
        FeatureMatcher *featMatcher = &matcher;
        std::vector<std::pair<int, int>> order;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (i != j && !features[i].empty() && !features[j].empty()) order.push_back({i, j});
        std::vector<std::map<int, int>> res(order.size());
        auto run = [&](bool cached) {
            std::atomic<size_t> next{0};
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < 4; ++t)                       // MAX_NUM_THREADS (SequentialReconstructor.h:17)
                th.emplace_back([&] {
                    for (size_t p; (p = next++) < order.size();) {
                        auto features1 = features[order[p].first];          // by-value copies, as :213-214
                        auto features2 = features[order[p].second];
                        std::map<int, int> curMatches;
                        if (cached) featMatcher->matchFeatures(features1, features2, curMatches, {336, 512}, {336, 512});
                        else matcher.matchFeaturesUncached(features1, features2, curMatches);
                        res[p] = std::move(curMatches);
                    }
                });
            for (auto &t : th) t.join();
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (double)order.size();
        };
        const double ms_cached = run(true);
        const int64_t uploads = (int64_t)matcher.uploads();
        int32_t np = (int32_t)order.size();
        fwrite(&np, 4, 1, o);
        for (size_t p = 0; p < order.size(); ++p) {
            int32_t rec[3] = {order[p].first, order[p].second, (int32_t)res[p].size()};
            fwrite(rec, 4, 3, o);
            for (const auto &qt : res[p]) { int32_t e[2] = {qt.first, qt.second}; fwrite(e, 4, 2, o); }
        }
        const double ms_cached2 = run(true);                 // every image resident by now
        const double ms_uncached = run(false);
        fwrite(&ms_cached2, 8, 1, o); fwrite(&ms_uncached, 8, 1, o); fwrite(&uploads, 8, 1, o);
        printf("plugin_loop_test ok: %d images, %zu calls; %.3f ms per call with the cache (first sweep %.3f), %.3f without; %lld uploads\n",
               n, order.size(), ms_cached2, ms_cached, ms_uncached, (long long)uploads);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    fclose(f); fclose(o);
    return 0;
}
