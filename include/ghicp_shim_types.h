// Stand-ins for the few PCL / Eigen types that appear in the reference's public signatures, used ONLY when the real libraries are
// absent (this image has neither).  They expose the REAL interface and nothing else -- Eigen: column-major storage, rows(), cols(),
// operator()(i, j), data(), resize(), Identity(), x()/y()/z(); PCL: points / width / height, 16-byte-padded point structs -- so
// the drop-in headers (ghicp_reg.h, km.h, common_reg.h, keypoint_detect.hpp, binary_feature_extraction.hpp) and caller code are
// written against that interface alone.  Define GHICP_WITH_PCL to build against real PCL / Eigen instead
// (tests/test_dropin_cpu.py compiles every header that way against interface-only fakes of both libraries).
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
struct alignas(16) PointXYZ { float x, y, z, pad_; };
struct alignas(16) PointXYZI { float x, y, z, pad_, intensity, pad2_[3]; };  // 32 bytes like pcl::PointXYZI
struct FPFHSignature33 { float histogram[33]; };
template <typename T> struct PointCloud {
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  typedef std::shared_ptr<const PointCloud<T>> ConstPtr;
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
const int Dynamic = -1;
template <typename S, int R, int C>
class Matrix {  // column-major, like Eigen's default
 public:
  typedef S Scalar;
  Matrix() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C), d_((std::size_t)r_ * c_, S()) {}
  Matrix(long r, long c) : r_(R == Dynamic ? r : R), c_(C == Dynamic ? c : C), d_((std::size_t)r_ * c_, S()) {}
  long rows() const { return r_; }
  long cols() const { return c_; }
  void resize(long r, long c) { r_ = R == Dynamic ? r : R; c_ = C == Dynamic ? c : C; d_.assign((std::size_t)r_ * c_, S()); }
  S& operator()(long i, long j) { return d_[(std::size_t)(i + j * r_)]; }
  const S& operator()(long i, long j) const { return d_[(std::size_t)(i + j * r_)]; }
  S& operator()(long i) { return d_[(std::size_t)i]; }
  const S& operator()(long i) const { return d_[(std::size_t)i]; }
  S& x() { return d_[0]; }
  S& y() { return d_[1]; }
  S& z() { return d_[2]; }
  const S& x() const { return d_[0]; }
  const S& y() const { return d_[1]; }
  const S& z() const { return d_[2]; }
  S* data() { return d_.data(); }
  const S* data() const { return d_.data(); }
  static Matrix Identity() { Matrix m; for (long i = 0; i < m.r_ && i < m.c_; i++) m(i, i) = S(1); return m; }

 private:
  long r_, c_;
  std::vector<S> d_;
};
typedef Matrix<double, Dynamic, 3> MatrixX3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<float, 3, 1> Vector3f;
}  // namespace Eigen
#endif
#endif
