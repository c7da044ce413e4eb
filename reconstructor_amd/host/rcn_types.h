// rcn_types.h -- boundary types of the hot path, re-declared without OpenCV / Eigen / PCL.
//
// Same names and member names as the reference (namespace reconstructor::Core) so that the
// adapters below read like the reference's own call sites:
//   FeatCoord / FeatDesc / Feature / FeaturePtr   datatypes.h:10-107
//   TriangulatedFeature / Landmark                datatypes.h:125-183
//   PinholeCamera                                 Camera.h:12-120 (fX,fY,cX,cY,k1,k2 + project)
// Only the members the matcher / bundle-adjuster boundary touches are declared.
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <unordered_map>
#include <utility>
#include <vector>

namespace reconstructor::Core {

template <typename coordType = int> struct FeatCoord {
    FeatCoord() = default;
    FeatCoord(coordType x_, coordType y_) : x(x_), y(y_) {}
    coordType x{}, y{};
};

struct FeatDesc {
    FeatDesc() = default;
    template <typename It> FeatDesc(It first, It last) : desc(first, last) {}
    std::vector<float> desc;   // 128 (SIFT) / 32 (ORB as float) / 256 (SuperPoint) floats
};

template <typename coordType = int> struct Feature {
    Feature() = default;
    Feature(FeatCoord<coordType> c, FeatDesc d) : featCoord(c), featDesc(std::move(d)) {}
    FeatCoord<coordType> featCoord;
    FeatDesc featDesc;
    int landmarkId = -1;
};
template <typename coordType = int> using FeaturePtr = std::shared_ptr<Feature<coordType>>;

struct TriangulatedFeature {
    TriangulatedFeature() = default;
    TriangulatedFeature(int img, int feat) : imgIdx(img), featIdx(feat) {}
    int imgIdx = 0, featIdx = 0;
};

struct Landmark {
    Landmark() = default;
    Landmark(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    std::vector<TriangulatedFeature> triangulatedFeatures;
    double x = 0, y = 0, z = 0;
};

// 4x4 row-major double matrix standing in for Eigen::Matrix4d at the boundary: M(r,c).
struct Mat4d {
    double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double &operator()(int r, int c) { return m[4 * r + c]; }
    double operator()(int r, int c) const { return m[4 * r + c]; }
};

}  // namespace reconstructor::Core

// global namespace in the reference too (Camera.h:12)
class PinholeCamera {
public:
    PinholeCamera() = default;
    PinholeCamera(int height, int width, double fx, double fy) : fX(fx), fY(fy), cX(width / 2), cY(height / 2) {}
    // u = fX (x + d) + cX, v = fY (y + d) + cY with d = k1 r + k2 r^2, r = x^2 + y^2: the
    // reference's additive model (Camera.h:59-76), identical to the BA residual.
    std::pair<double, double> project(double X, double Y, double Z) const
    {
        const double x = X / Z, y = Y / Z, r = x * x + y * y, d = k1 * r + k2 * r * r;
        return {fX * (x + d) + cX, fY * (y + d) + cY};
    }
    double fX = 0, fY = 0, cX = 0, cY = 0, k1 = 0, k2 = 0;
};
