// Minimal stand-ins for the few PCL / Eigen types that appear in the reference's public signatures, used ONLY
// when the real libraries are absent (this image).  Define GHICP_WITH_PCL to build against real PCL/Eigen: the
// class and method names in ghicp_reg.h / km.h / keypoint_detect.hpp / binary_feature_extraction.hpp are the
// reference's, so caller code compiles unchanged either way.
#ifndef GHICP_SHIM_TYPES_H_
#define GHICP_SHIM_TYPES_H_
#ifdef GHICP_WITH_PCL
#include <pcl/PointIndices.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <Eigen/Core>
#else
#include <cstddef>
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZ { float x, y, z, pad_; };
struct alignas(16) PointXYZI { float x, y, z, pad_, intensity, pad2_[3]; };  // 32 bytes like pcl::PointXYZI
template <typename T> struct PointCloud {
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  std::vector<T> points;
  unsigned width = 0, height = 1;
  std::size_t size() const { return points.size(); }
  T& operator[](std::size_t i) { return points[i]; }
  const T& operator[](std::size_t i) const { return points[i]; }
  void push_back(const T& p) { points.push_back(p); width = (unsigned)points.size(); }
};
struct PointIndices { std::vector<int> indices; };
typedef std::shared_ptr<PointIndices> PointIndicesPtr;
}  // namespace pcl
namespace Eigen {
struct Matrix4d {
  double m[16];
  double& operator()(int r, int c) { return m[r * 4 + c]; }
  double operator()(int r, int c) const { return m[r * 4 + c]; }
  static Matrix4d Identity() { Matrix4d M; for (int i = 0; i < 16; i++) M.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return M; }
};
struct Matrix4f {
  float m[16];  // row-major
  float& operator()(int r, int c) { return m[r * 4 + c]; }
  float operator()(int r, int c) const { return m[r * 4 + c]; }
  static Matrix4f Identity() { Matrix4f M; for (int i = 0; i < 16; i++) M.m[i] = (i % 5 == 0) ? 1.0f : 0.0f; return M; }
};
struct MatrixX3d {
  std::vector<double> d;  // row-major rows() x 3
  long n = 0;
  void resize(long r, long) { n = r; d.assign((std::size_t)r * 3, 0.0); }
  long rows() const { return n; }
  double& operator()(long r, long c) { return d[(std::size_t)r * 3 + c]; }
  double operator()(long r, long c) const { return d[(std::size_t)r * 3 + c]; }
};
}  // namespace Eigen
#endif
#endif
