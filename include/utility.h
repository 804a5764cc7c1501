// Drop-in for the part of the reference's include/utility.h that is API surface of the hot path:
// enums (utility.h:51-64), Bounds (utility.h:78-90) and the cloud typedefs (utility.h:24-46).
#ifndef GHICP_DROPIN_UTILITY_H_
#define GHICP_DROPIN_UTILITY_H_
#include "ghicp_c.h"
#include "ghicp_shim_types.h"

#include <stdexcept>
#include <string>

typedef pcl::PointCloud<pcl::PointXYZI>::Ptr pcXYZIPtr;
typedef pcl::PointCloud<pcl::PointXYZI> pcXYZI;
typedef pcl::PointCloud<pcl::PointXYZ>::Ptr pcXYZPtr;
typedef pcl::PointCloud<pcl::PointXYZ> pcXYZ;
typedef pcl::PointCloud<pcl::FPFHSignature33>::Ptr fpfhFeaturePtr;  // utility.h:45-46
typedef pcl::PointCloud<pcl::FPFHSignature33> fpfhFeature;

namespace ghicp {
enum FeatureType { BSC, RoPS, FPFH, None };      // utility.h:51-57 (== GHICP_FEATURE_*)
enum CorrespondenceType { NN, NNR, KM };         // utility.h:59-64 (== GHICP_CORR_*)
struct CenterPoint {                             // utility.h:66-76 (the reference's constructor assigns its PARAMETERS; the members start at 0 here)
  double x, y, z;
  CenterPoint(double x_ = 0, double y_ = 0, double z_ = 0) : x(x_), y(y_), z(z_) {}
};
struct Bounds {                                  // utility.h:78-90
  double min_x, min_y, min_z, max_x, max_y, max_z;
  Bounds() { min_x = min_y = min_z = max_x = max_y = max_z = 0.0; }
};

namespace detail {
// One lazily created context in host-pointer mode shared by the drop-in classes (the reference is a
// single-threaded host program; for concurrent use create ghicp_ctx objects directly).
inline ghicp_ctx* ctx() {
  static ghicp_ctx* c = nullptr;
  if (!c) {
    if (ghicp_ctx_create(0, &c) != GHICP_OK) throw std::runtime_error("ghicp: no MI355X device / HIP library (there is no CPU fallback)");
    ghicp_ctx_set_host_pointers(c, 1);
  }
  return c;
}
inline void check(int rc) {
  if (rc != GHICP_OK) throw std::runtime_error(std::string("ghicp: ") + ghicp_last_error(ctx()));
}
template <typename PointT> inline const float* xyz(const pcl::PointCloud<PointT>& c) { return c.points.empty() ? nullptr : &c.points[0].x; }
template <typename PointT> inline int stride() { return (int)(sizeof(PointT) / sizeof(float)); }
}  // namespace detail
}  // namespace ghicp
#endif
