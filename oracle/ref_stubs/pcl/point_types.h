// TEST INFRASTRUCTURE.  Minimal stand-ins so that the reference's headers PARSE without PCL (oracle/_ref builds of the reference's
// own km.cpp, stereo_binary_feature.cpp and the plain-C++ members of ghicp_reg.cpp; tests/cpp compile tests of the drop-in headers).
// Written from scratch against PCL's public interface; not PCL code.
#pragma once
#include <cstddef>
#include <iostream>
#include <memory>
#include <vector>
#include <Eigen/Core>
namespace pcl {
struct PointXYZ { float x, y, z; };
struct PointXY { float x, y; };
struct PointXYZI { float x, y, z, intensity; };
struct PointXYZRGB { float x, y, z; unsigned rgba; };
struct PointXYZRGBA { float x, y, z; unsigned rgba; };
struct Normal { float normal_x, normal_y, normal_z, curvature; };
struct PointNormal { float x, y, z, normal_x, normal_y, normal_z, curvature; };
struct PointXYZINormal { float x, y, z, intensity, normal_x, normal_y, normal_z, curvature; };
struct FPFHSignature33 { float histogram[33]; };
template <typename T> struct PointCloud {
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  typedef std::shared_ptr<const PointCloud<T>> ConstPtr;
  std::vector<T> points;
  unsigned width = 0, height = 0;
  std::size_t size() const { return points.size(); }
  const T& operator[](std::size_t i) const { return points[i]; }
  T& operator[](std::size_t i) { return points[i]; }
  void push_back(const T& p) { points.push_back(p); }
};
struct PointIndices { std::vector<int> indices; };
typedef std::shared_ptr<PointIndices> PointIndicesPtr;
template <typename T> class KdTreeFLANN {};
namespace search { template <typename T> class KdTree { public: typedef std::shared_ptr<KdTree<T>> Ptr; }; }
template <typename A, typename B> class NormalEstimation {};
template <typename A, typename B, typename C> class FPFHEstimationOMP {};
template <typename A, typename B, typename C> class SampleConsensusInitialAlignment {};
}  // namespace pcl
