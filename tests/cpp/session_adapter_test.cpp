// session_adapter_test -- the reference's incremental loop (SequentialReconstructor.cpp:1040-1094: one global
// adjust per registered view) driven through BOTH bundle-adjuster adapters on equal containers:
//   BundleAdjuster::adjust              (HipBundleAdjuster.h: re-packs and re-sends the whole problem every call)
//   IncrementalBundleAdjuster::adjust   (HipBundleSession.h: device-resident session, sends what was added)
// After every view the landmarks, poses and intrinsics of the two must be equal bit for bit.  One round erases a
// landmark and an observation the way removeOutlierLandmarks / checkLandmarkValidity do: the session must notice
// and rebuild itself.
// Input (tests/test_cpp_adapter.py): nc, np, order[nc], 4x4 poses, intrinsics, then per scene point its xyz, an
// observation count and (local camera, x, y) records in ascending camera order.
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../reconstructor_amd/host/HipBundleSession.h"

using namespace reconstructor::Core;

static void rd(FILE *f, void *p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

struct Obs { int cam, x, y; };
struct Containers {
    std::unordered_map<int, std::vector<FeaturePtr<>>> feats;
    std::vector<Landmark> landmarks;
    std::unordered_map<int, Mat4d> poses;
    std::unordered_map<int, PinholeCamera> intr;
};

static bool same(const Containers &a, const Containers &b, const std::vector<int> &order)
{
    if (a.landmarks.size() != b.landmarks.size()) return false;
    for (size_t j = 0; j < a.landmarks.size(); ++j)
        if (memcmp(&a.landmarks[j].x, &b.landmarks[j].x, 8) || memcmp(&a.landmarks[j].y, &b.landmarks[j].y, 8) || memcmp(&a.landmarks[j].z, &b.landmarks[j].z, 8)) return false;
    for (int g : order) {
        if (memcmp(a.poses.at(g).m, b.poses.at(g).m, 128)) return false;
        const PinholeCamera &x = a.intr.at(g), &y = b.intr.at(g);
        const double u[6] = {x.fX, x.fY, x.cX, x.cY, x.k1, x.k2}, v[6] = {y.fX, y.fY, y.cX, y.cY, y.k1, y.k2};
        if (memcmp(u, v, 48)) return false;
    }
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: session_adapter_test <scene.bin>\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int32_t nc, np;
    rd(f, &nc, 4); rd(f, &np, 4);
    std::vector<int> order(nc);
    rd(f, order.data(), 4 * (size_t)nc);
    std::vector<Mat4d> T0(nc);
    std::vector<PinholeCamera> K0(nc);
    for (int l = 0; l < nc; ++l) rd(f, T0[l].m, 128);
    for (int l = 0; l < nc; ++l) { double k[6]; rd(f, k, 48); K0[l].fX = k[0]; K0[l].fY = k[1]; K0[l].cX = k[2]; K0[l].cY = k[3]; K0[l].k1 = k[4]; K0[l].k2 = k[5]; }
    std::vector<std::array<double, 3>> X0(np);
    std::vector<std::vector<Obs>> obs(np);
    for (int j = 0; j < np; ++j) {
        int32_t cnt;
        rd(f, X0[j].data(), 24); rd(f, &cnt, 4);
        for (int k = 0; k < cnt; ++k) { int32_t r[3]; rd(f, r, 12); obs[j].push_back({r[0], r[1], r[2]}); }
    }
    fclose(f);

    rcn_ctx *ctx = nullptr;
    if (rcn_create(0, &ctx) != RCN_OK) { fprintf(stderr, "no device\n"); return 3; }
    BundleAdjuster whole(ctx);
    IncrementalBundleAdjuster incremental(ctx);
    Containers A, B;                        // A: re-packed every call, B: through the session
    std::vector<int> lm_of(np, -1);         // scene point -> landmark index
    std::vector<size_t> used(np, 0);        // observations of the scene point already in its track
    std::vector<int> registered;
    int rounds = 0, expected_rebuilds = 0;
    for (int r = 0; r < nc; ++r) {
        // register view r: its initial pose estimate (what PnP would hand over), its features
        const int g = order[r];
        registered.push_back(g);
        for (Containers *c : {&A, &B}) { c->poses[g] = T0[r]; c->intr[g] = K0[r]; c->feats[g] = {}; }
        if (r < 2) continue;                // the reference starts adjusting with the third view
        for (int j = 0; j < np; ++j) {
            size_t visible = 0;
            while (visible < obs[j].size() && obs[j][visible].cam <= r) ++visible;
            if (visible < 2 || lm_of[j] == -2) continue;      // -2: erased by the sweep below, never triangulated again
            if (lm_of[j] < 0) {             // triangulated now: a new landmark at the end of the vector
                lm_of[j] = (int)A.landmarks.size();
                for (Containers *c : {&A, &B}) c->landmarks.emplace_back(X0[j][0], X0[j][1], X0[j][2]);
            }
            for (size_t k = used[j]; k < visible; ++k) {      // push_back onto the track
                const int gi = order[obs[j][k].cam];
                for (Containers *c : {&A, &B}) {
                    c->feats[gi].push_back(std::make_shared<Feature<>>(FeatCoord<>(obs[j][k].x, obs[j][k].y), FeatDesc()));
                    c->feats[gi].back()->landmarkId = lm_of[j];
                    c->landmarks[lm_of[j]].triangulatedFeatures.emplace_back(gi, (int)c->feats[gi].size() - 1);
                }
            }
            used[j] = visible;
        }
        if (r == nc - 2 && A.landmarks.size() > 8) {
            // what the validity sweep does between two adjusts: one observation erased from a long track, one
            // landmark erased from the middle of the vector (later landmarks slide down)
            for (Containers *c : {&A, &B}) {
                for (auto &lm : c->landmarks)
                    if (lm.triangulatedFeatures.size() > 3) { lm.triangulatedFeatures.erase(lm.triangulatedFeatures.begin() + 1); break; }
                c->landmarks.erase(c->landmarks.begin() + 5);
            }
            for (int j = 0; j < np; ++j) { if (lm_of[j] == 5) { lm_of[j] = -2; } else if (lm_of[j] > 5) --lm_of[j]; }
            ++expected_rebuilds;
        }
        auto ga = whole.adjust(A.feats, A.landmarks, A.poses, A.intr, registered);
        auto gb = incremental.adjust(B.feats, B.landmarks, B.poses, B.intr, registered);
        ++rounds;
        if (ga != gb || !same(A, B, registered) || whole.summary.iterations != incremental.summary.iterations ||
            memcmp(&whole.summary.final_cost, &incremental.summary.final_cost, 8)) {
            fprintf(stderr, "view %d: the two adapters disagree (iterations %d / %d, rms %.17g / %.17g)\n", r, whole.summary.iterations,
                    incremental.summary.iterations, whole.summary.final_rms_px, incremental.summary.final_rms_px);
            return 1;
        }
    }
    if (incremental.rebuilds() != expected_rebuilds) { fprintf(stderr, "rebuilds %d, expected %d\n", incremental.rebuilds(), expected_rebuilds); return 1; }
    printf("session_adapter_test ok: %d adjusts, %zu landmarks, %d rebuild(s), last rms %.6f\n", rounds, B.landmarks.size(), incremental.rebuilds(),
           incremental.summary.final_rms_px);
    rcn_destroy(ctx);
    return 0;
}
