// grid_driver_test.cpp -- drives HipPairGridDriver the way SequentialReconstructor::matchFeatures is driven
// (SequentialReconstructor.cpp:199-279, filter off): `features` in, `featureMatches` out, on every visible GPU
// (RCCL communicators even at world size 1).  Input written by tests/test_cpp_grid_driver.py:
//   i32 n, i32 D, then per image: i32 K, K*D floats.      Output: i32 world, i32 n_entries, then per map entry
//   (ascending (i, j)): i32 i, i32 j, i32 count, count x (i32 query, i32 train) in ascending query order.
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
    std::vector<float> row(D);
    for (int i = 0; i < n; ++i) {
        int32_t K;
        rd(f, &K, 4);
        features[i] = {};
        for (int k = 0; k < K; ++k) {
            rd(f, row.data(), 4 * (size_t)D);
            features[i].push_back(std::make_shared<Feature<>>(FeatCoord<>(k, k), FeatDesc(row.begin(), row.end())));
        }
    }
    FeatureMatches featureMatches;
    int world = 0;
    try {
        HipPairGridDriver driver(argc > 3 ? atoi(argv[3]) : 0);
        world = driver.world();
        driver.matchFeatures(features, featureMatches);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
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
