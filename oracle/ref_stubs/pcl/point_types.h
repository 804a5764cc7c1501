// TEST INFRASTRUCTURE.  Minimal stand-ins so that the reference's headers PARSE without PCL (oracle/_ref builds of the reference's
// own km.cpp, stereo_binary_feature.cpp and the plain-C++ members of ghicp_reg.cpp; tests/cpp compile tests of the drop-in headers).
// Written from scratch against PCL's public interface; not PCL code.
#pragma once
#include <cstddef>
#include <iostream>
#include <memory>
#include <algorithm>
#include <utility>
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
  typedef typename std::vector<T>::iterator iterator;
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  Ptr makeShared() const { return Ptr(new PointCloud<T>(*this)); }
};
// pcl::getMinMax3D (common.h): per-coordinate minimum / maximum of the finite points, 4th component 0
template <typename T> inline void getMinMax3D(const PointCloud<T>& c, Eigen::Vector4f& mn, Eigen::Vector4f& mx) {
  mn = Eigen::Vector4f(); mx = Eigen::Vector4f();
  for (int d = 0; d < 3; d++) { mn(d) = 3.402823466e+38f; mx(d) = -3.402823466e+38f; }
  for (const T& p : c.points) {
    const float v[3] = {p.x, p.y, p.z};
    for (int d = 0; d < 3; d++) { if (v[d] < mn(d)) mn(d) = v[d]; if (v[d] > mx(d)) mx(d) = v[d]; }
  }
}
struct PointIndices { std::vector<int> indices; };
typedef std::shared_ptr<PointIndices> PointIndicesPtr;
template <typename T> inline float kd_d2_(const T& p, const T& q) { const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z; return dx * dx + dy * dy + dz * dz; }
inline float kd_d2_(const PointXY& p, const PointXY& q) { const float dx = p.x - q.x, dy = p.y - q.y; return dx * dx + dy * dy; }
// pcl::KdTreeFLANN: exact search (FLANN with eps = 0).  radiusSearch: squared L2 distance in float, strict d^2 < r^2 (FLANN's
// RadiusResultSet), the query itself included when it is a point of the cloud, results sorted by distance (ties: lower index).
template <typename T> class KdTreeFLANN {
 public:
  void setInputCloud(const typename PointCloud<T>::ConstPtr& c) { cloud_ = c; }
  int radiusSearch(const T& q, double radius, std::vector<int>& idx, std::vector<float>& d2, unsigned max_nn = 0) const {
    idx.clear(); d2.clear();
    const float r2 = (float)(radius * radius);
    std::vector<std::pair<float, int>> hits;
    for (std::size_t i = 0; i < cloud_->points.size(); i++) {
      const T& p = cloud_->points[i];
      const float d = kd_d2_(p, q);
      if (d < r2) hits.push_back(std::make_pair(d, (int)i));
    }
    std::sort(hits.begin(), hits.end());
    for (std::size_t k = 0; k < hits.size() && (max_nn == 0 || k < max_nn); k++) { idx.push_back(hits[k].second); d2.push_back(hits[k].first); }
    return (int)idx.size();
  }
 private:
  typename PointCloud<T>::ConstPtr cloud_;
};
namespace search { template <typename T> class KdTree { public: typedef std::shared_ptr<KdTree<T>> Ptr; }; }
template <typename A, typename B> class NormalEstimation {};
template <typename A, typename B, typename C> class FPFHEstimationOMP {};
template <typename A, typename B, typename C> class SampleConsensusInitialAlignment {};
}  // namespace pcl
