// adapter_test.cpp -- exercises the C++ adapters exactly as the reference's pipeline would:
//   featMatcher->matchFeatures(features1, features2, curMatches, shape1, shape2)   (SequentialReconstructor.cpp:232)
//   featFilter->estimateFundamental(featuresMatched1, featuresMatched2, inlierMatchIds)           (:250)
//   BundleAdjuster().adjust(features, landmarks, imgIdx2camPose, imgIdx2camIntrinsics, imgIdxOrder)   (:1064-1069)
//   checkLandmarkValidity() / removeOutlierLandmarks()                               (:1071-1074)
// Reads a small binary problem written by tests/test_cpp_adapter.py, writes the results back.
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "../../reconstructor_amd/host/HipBundleAdjuster.h"
#include "../../reconstructor_amd/host/HipFeatureMatcher.h"
#include "../../reconstructor_amd/host/HipGeometricFilter.h"
#include "../../reconstructor_amd/host/HipLandmarkValidity.h"

using namespace reconstructor::Core;

static void rd(FILE *f, void *p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    FILE *o = fopen(argv[2], "wb");
    if (!f || !o) return 2;
    // ---- matcher: K1, K2, D, then K1*D and K2*D floats
    int32_t K1, K2, D;
    rd(f, &K1, 4); rd(f, &K2, 4); rd(f, &D, 4);
    auto load = [&](int K) {
        std::vector<FeaturePtr<>> v;
        std::vector<float> row(D);
        for (int i = 0; i < K; ++i) {
            rd(f, row.data(), 4 * (size_t)D);
            v.push_back(std::make_shared<Feature<>>(FeatCoord<>(i, i), FeatDesc(row.begin(), row.end())));
        }
        return v;
    };
    auto f1 = load(K1), f2 = load(K2);
    std::unique_ptr<FeatureMatcher> featMatcher = std::make_unique<HipL2Matcher>();
    std::map<int, int> curMatches;
    featMatcher->matchFeatures(f1, f2, curMatches, {336, 512}, {336, 512});
    int32_t n = (int32_t)curMatches.size();
    fwrite(&n, 4, 1, o);
    for (auto &[q, t] : curMatches) { int32_t p[2] = {q, t}; fwrite(p, 4, 2, o); }

    // ---- epipolar filter on those matches, as SequentialReconstructor.cpp:237-269 does: feature
    //      coordinates follow the descriptors in the input (K1 + K2 integer pairs)
    {
        std::vector<int32_t> c1(2 * (size_t)K1), c2(2 * (size_t)K2);
        rd(f, c1.data(), 8 * (size_t)K1); rd(f, c2.data(), 8 * (size_t)K2);
        for (int i = 0; i < K1; ++i) f1[i]->featCoord = FeatCoord<>(c1[2 * i], c1[2 * i + 1]);
        for (int i = 0; i < K2; ++i) f2[i]->featCoord = FeatCoord<>(c2[2 * i], c2[2 * i + 1]);
        std::map<int, int> filtered;
        HipL2Matcher *hm0 = static_cast<HipL2Matcher *>(featMatcher.get());
        auto featFilter = std::make_unique<GeometricFilter>(hm0->context());
        if (curMatches.size() >= 7) {
            std::vector<FeaturePtr<>> featuresMatched1, featuresMatched2;
            for (const auto &[featIdx1, featIdx2] : curMatches) { featuresMatched1.push_back(f1[featIdx1]); featuresMatched2.push_back(f2[featIdx2]); }
            std::vector<bool> inlierMatchIds;
            featFilter->estimateFundamental(featuresMatched1, featuresMatched2, inlierMatchIds);
            if (inlierMatchIds.size() != 0) {
                int curMatchId = 0;
                for (const auto &[featIdx1, featIdx2] : curMatches) { if (inlierMatchIds[curMatchId]) filtered[featIdx1] = featIdx2; ++curMatchId; }
            }
        } else filtered = curMatches;
        int32_t nf = (int32_t)filtered.size();
        fwrite(&nf, 4, 1, o);
        for (auto &[q, t] : filtered) { int32_t p[2] = {q, t}; fwrite(p, 4, 2, o); }
    }

    // ---- BA: nc, np, order[nc], poses 4x4 (row-major) per cam, intr 6 per cam, points, then per
    //      landmark: count + (local cam, x, y)
    int32_t nc, np;
    rd(f, &nc, 4); rd(f, &np, 4);
    std::vector<int> order(nc);
    rd(f, order.data(), 4 * (size_t)nc);
    std::unordered_map<int, Mat4d> poses;
    std::unordered_map<int, PinholeCamera> intr;
    std::unordered_map<int, std::vector<FeaturePtr<>>> feats;
    for (int l = 0; l < nc; ++l) { Mat4d T; rd(f, T.m, 128); poses[order[l]] = T; }
    for (int l = 0; l < nc; ++l) {
        double k[6]; rd(f, k, 48);
        PinholeCamera c; c.fX = k[0]; c.fY = k[1]; c.cX = k[2]; c.cY = k[3]; c.k1 = k[4]; c.k2 = k[5];
        intr[order[l]] = c; feats[order[l]] = {};
    }
    std::vector<Landmark> landmarks;
    for (int j = 0; j < np; ++j) {
        double X[3]; int32_t cnt;
        rd(f, X, 24); rd(f, &cnt, 4);
        Landmark lm(X[0], X[1], X[2]);
        for (int k = 0; k < cnt; ++k) {
            int32_t rec[3]; rd(f, rec, 12);
            const int g = order[rec[0]];
            feats[g].push_back(std::make_shared<Feature<>>(FeatCoord<>(rec[1], rec[2]), FeatDesc()));
            feats[g].back()->landmarkId = j;
            lm.triangulatedFeatures.emplace_back(g, (int)feats[g].size() - 1);
        }
        landmarks.push_back(lm);
    }
    HipL2Matcher *hm = static_cast<HipL2Matcher *>(featMatcher.get());
    BundleAdjuster bundleAdjuster(hm->context());
    auto g2l = bundleAdjuster.adjust(feats, landmarks, poses, intr, order);
    fwrite(&bundleAdjuster.summary.final_rms_px, 8, 1, o);
    fwrite(&bundleAdjuster.summary.iterations, 4, 1, o);
    for (auto &lm : landmarks) { double X[3] = {lm.x, lm.y, lm.z}; fwrite(X, 8, 3, o); }
    for (int l = 0; l < nc; ++l) fwrite(poses[order[l]].m, 8, 16, o);
    for (int l = 0; l < nc; ++l) { int32_t v = g2l[order[l]]; fwrite(&v, 4, 1, o); }
    // ---- validity sweep on the adjusted scene, with the thresholds read from the input
    double thr[2];
    rd(f, thr, 16);
    LandmarkValidator validator(hm->context());
    validator.maxProjectionError = thr[0]; validator.minTriangulationAngle = thr[1];
    auto inlierIds = validator.checkLandmarkValidity(feats, landmarks, poses, intr);
    for (size_t j = 0; j < landmarks.size(); ++j) {
        int32_t rec[2] = {inlierIds[j] ? 1 : 0, (int32_t)landmarks[j].triangulatedFeatures.size()};
        fwrite(rec, 4, 2, o);
    }
    LandmarkValidator::removeOutlierLandmarks(feats, landmarks, inlierIds);
    int32_t left = (int32_t)landmarks.size(), unassigned = 0;
    for (auto &kv : feats) for (auto &ft : kv.second) unassigned += ft->landmarkId == -1;
    fwrite(&left, 4, 1, o); fwrite(&unassigned, 4, 1, o);
    fclose(f); fclose(o);
    printf("adapter_test ok: %d matches, BA %d iterations, rms %.6f, %d landmarks valid\n", n, bundleAdjuster.summary.iterations, bundleAdjuster.summary.final_rms_px, left);
    return 0;
}
