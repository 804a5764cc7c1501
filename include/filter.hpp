// Drop-in for include/filter.hpp: CFilter<PointT>::voxelfilter (filter.hpp:28-88) on the GPU, and CloudUtility<PointT>::getCloudBound
// (utility.h:153-183), which CFilter inherits and test/ghicp_main.cpp:87-93 calls on the down-sampled source for bbx_magnitude.
// SORFilter / DisFilter / ActiveObjectFilter (filter.hpp:91-183) are outside the hot path and are not declared.
#ifndef GHICP_DROPIN_FILTER_HPP_
#define GHICP_DROPIN_FILTER_HPP_
#include <iostream>
#include <vector>

#include "utility.h"

namespace ghicp {
template <typename PointT> class CloudUtility {  // utility.h:132-200 (the members the hot path's callers use)
 public:
  // utility.h:136-150: centroid as a running sum of coordinate / size in double (host arithmetic: one pass over the caller's vector)
  void getCloudCenterPoint(const typename pcl::PointCloud<PointT>& cloud, CenterPoint& centerPoint) {
    double cx = 0, cy = 0, cz = 0;
    const size_t n = cloud.points.size();
    for (size_t i = 0; i < n; i++) {
      cx += cloud.points[i].x / n;
      cy += cloud.points[i].y / n;
      cz += cloud.points[i].z / n;
    }
    centerPoint.x = cx; centerPoint.y = cy; centerPoint.z = cz;
  }
  void getBoundAndCenter(const typename pcl::PointCloud<PointT>& cloud, Bounds& bound, CenterPoint& centerPoint) {  // utility.h:186-190
    getCloudCenterPoint(cloud, centerPoint);
    getCloudBound(cloud, bound);
  }

  // utility.h:153-183: the six extremes of the cloud (doubles holding float values).  An empty cloud is undefined behaviour in the
  // reference (it reads cloud[0]); here the bounds stay as they were.
  void getCloudBound(const typename pcl::PointCloud<PointT>& cloud, Bounds& bound) {
    double b[6];
    const int64_t n = (int64_t)cloud.points.size();
    if (n <= 0) return;
    detail::check(ghicp_cloud_bounds(detail::ctx(), detail::xyz(cloud), n, detail::stride<PointT>(), b));
    bound.min_x = b[0]; bound.min_y = b[1]; bound.min_z = b[2];
    bound.max_x = b[3]; bound.max_y = b[4]; bound.max_z = b[5];
  }
};

template <typename PointT> class CFilter : public CloudUtility<PointT> {
 public:
  // filter.hpp:28-88: one point per occupied voxel, in ascending voxel order, appended to cloud_out; the reference's phantom entries
  // (its id_pairs vector is sized AND pushed to, quirk Q1) put a copy of input point 0 first.  Whole point structs are copied, so
  // intensity and padding travel like in the reference's push_back.
  bool voxelfilter(const typename pcl::PointCloud<PointT>::Ptr& cloud_in, typename pcl::PointCloud<PointT>::Ptr& cloud_out, float voxel_size) {
    const int64_t n = (int64_t)cloud_in->points.size();
    std::vector<int32_t> keep((size_t)n + 1);
    int64_t m = 0;
    detail::check(ghicp_voxel_filter(detail::ctx(), detail::xyz(*cloud_in), n, detail::stride<PointT>(), voxel_size, keep.data(), &m));
    for (int64_t i = 0; i < m; i++) cloud_out->push_back(cloud_in->points[(size_t)keep[(size_t)i]]);
    std::cout << "Downsample done (" << cloud_out->points.size() << " points)" << std::endl;
    return 1;
  }
};
}  // namespace ghicp
#endif
