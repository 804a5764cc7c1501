// Minimal stand-ins so that /root/reference/include/utility.h PARSES without PCL/Eigen.
// Test infrastructure only (oracle/_ref build of the reference's own km.cpp, which uses
// nothing from these headers).  Written from scratch; not PCL code.
#pragma once
#include <cstddef>
#include <iostream>
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZ { float x, y, z; };
struct PointXY { float x, y; };
struct PointXYZI { float x, y, z, intensity; };
struct PointXYZRGB { float x, y, z; unsigned rgba; };
struct PointXYZRGBA { float x, y, z; unsigned rgba; };
struct Normal { float normal_x, normal_y, normal_z, curvature; };
struct PointXYZINormal { float x, y, z, intensity, normal_x, normal_y, normal_z, curvature; };
struct FPFHSignature33 { float histogram[33]; };
template <typename T> struct PointCloud {
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  std::vector<T> points;
  std::size_t size() const { return points.size(); }
  const T& operator[](std::size_t i) const { return points[i]; }
  T& operator[](std::size_t i) { return points[i]; }
  void push_back(const T& p) { points.push_back(p); }
};
}  // namespace pcl
namespace Eigen {
struct StubMat {
  void resize(long, long) {}
  float& operator()(long, long) { static float f; return f; }
  StubMat operator*(const StubMat&) const { return StubMat(); }
};
typedef StubMat Matrix4f;
typedef StubMat Matrix4Xf;
}  // namespace Eigen
