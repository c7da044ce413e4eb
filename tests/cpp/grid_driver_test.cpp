// grid_driver_test.cpp -- drives HipPairGridDriver the way SequentialReconstructor::matchFeatures is driven
// (SequentialReconstructor.cpp:199-279): `features` in, `featureMatches` out, on every visible GPU
// (RCCL communicators even at world size 1).  Input written by tests/test_cpp_grid_driver.py:
//   i32 n, i32 D, then per image: i32 K, K*D floats, K*2 i32 pixel coordinates.
// Output: i32 world, i32 n_entries, then per map entry
//   (ascending (i, j)): i32 i, i32 j, i32 count, count x (i32 query, i32 train) in ascending query order.
// argv: in out [devices (0 = all)] [filter 0/1] [rank whose local step is made to fail, -1 = none]
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "../../reconstructor_amd/host/HipPairGridDriver.h"

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
        features[i] = {};
        std::vector<float> rows((size_t)K * D);
        std::vector<int32_t> xy(2 * (size_t)K);
        rd(f, rows.data(), 4 * rows.size());
        rd(f, xy.data(), 4 * xy.size());
        for (int k = 0; k < K; ++k)
            features[i].push_back(std::make_shared<Feature<>>(FeatCoord<>(xy[2 * k], xy[2 * k + 1]),
                                                              FeatDesc(rows.begin() + (size_t)k * D, rows.begin() + (size_t)(k + 1) * D)));
    }
    const bool filter = argc > 4 && atoi(argv[4]) != 0;
    const int inject = argc > 5 ? atoi(argv[5]) : -1;
    FeatureMatches featureMatches;
    int world = 0;
    try {
        HipPairGridDriver driver(argc > 3 ? atoi(argv[3]) : 0);
        world = driver.world();
        if (inject >= 0) {
            // fault injection: the rank reports a failure of its own; every entry point must come back with an error
            // (the driver throws) instead of leaving the peers inside a collective -- and the driver must work again after
            driver.injectLocalFailure(inject);
            bool threw = false;
            try { driver.matchFeatures(features, featureMatches, filter); } catch (const std::exception &e) { threw = true; fprintf(stderr, "injected: %s\n", e.what()); }
            if (!threw || !featureMatches.empty()) { fprintf(stderr, "the injected failure went unnoticed\n"); return 3; }
        }
        driver.matchFeatures(features, featureMatches, filter);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    if (argc > 6) {
        // shard + filter + store in one C++ run: everything the loop consumed and produced goes to the store file and comes
        // back equal -- descriptors and coordinates bit for bit, every map entry
        HipPairGridDriver::saveStore(argv[6], features, featureMatches);
        std::unordered_map<int, std::vector<FeaturePtr<>>> f2;
        FeatureMatches m2;
        HipPairGridDriver::loadStore(argv[6], f2, m2);
        bool same = f2.size() == features.size() && m2.size() == featureMatches.size();
        for (int i = 0; same && i < n; ++i) {
            same = f2[i].size() == features[i].size();
            for (size_t k = 0; same && k < features[i].size(); ++k)
                same = f2[i][k]->featDesc.desc == features[i][k]->featDesc.desc && f2[i][k]->featCoord.x == features[i][k]->featCoord.x &&
                       f2[i][k]->featCoord.y == features[i][k]->featCoord.y;
        }
        for (const auto &kv : featureMatches) {
            if (!same) break;
            auto it = m2.find(kv.first);
            same = it != m2.end() && it->second == kv.second;
        }
        if (!same) { fprintf(stderr, "the store file does not give back what was saved\n"); return 4; }
        printf("store round trip ok: %s\n", argv[6]);
    }
    std::vector<std::pair<int, int>> keys;
    for (const auto &kv : featureMatches) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    int32_t hdr[2] = {world, (int32_t)keys.size()};
    fwrite(hdr, 4, 2, o);
    size_t total = 0;
    for (const auto &k : keys) {
        const auto &m = featureMatches[k];
        std::vector<std::pair<int, int>> e(m.begin(), m.end());
        std::sort(e.begin(), e.end());
        int32_t rec[3] = {k.first, k.second, (int32_t)e.size()};
        fwrite(rec, 4, 3, o);
        for (const auto &qt : e) { int32_t p[2] = {qt.first, qt.second}; fwrite(p, 4, 2, o); }
        total += e.size();
    }
    fclose(f); fclose(o);
    printf("grid_driver_test ok: world %d, %d images, %zu map entries, %zu matches\n", world, n, keys.size(), total);
    return 0;
}
