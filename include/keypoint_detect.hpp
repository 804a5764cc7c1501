// Drop-in for include/keypoint_detect.hpp: CKeypointDetect<PointT>::keypointDetectionBasedOnCurvature
// (PCA curvature -> prune -> greedy NMS) on the GPU.
#ifndef GHICP_DROPIN_KEYPOINT_DETECT_HPP_
#define GHICP_DROPIN_KEYPOINT_DETECT_HPP_
#include <iostream>

#include "utility.h"

namespace ghicp {
template <typename PointT> class CKeypointDetect {
 public:
  CKeypointDetect(float neighborhood_radius, float ratio_unstable_thre, int min_point_num_neighborhood, float curvature_non_max_radius)
      : _neighborhood_radius(neighborhood_radius), _ratio_unstable_thre(ratio_unstable_thre),
        _min_point_num_neighborhood(min_point_num_neighborhood), _curvature_non_max_radius(curvature_non_max_radius) {}

  bool keypointDetectionBasedOnCurvature(const typename pcl::PointCloud<PointT>::Ptr& inputPointCloud, pcl::PointIndicesPtr& keypointIndices) {
    keypointIndices = pcl::PointIndicesPtr(new pcl::PointIndices());
    const int64_t m = (int64_t)inputPointCloud->points.size();
    keypointIndices->indices.assign((size_t)(m > 0 ? m : 1), 0);
    int64_t k = 0;
    detail::check(ghicp_keypoints(detail::ctx(), detail::xyz(*inputPointCloud), m, detail::stride<PointT>(), _neighborhood_radius, _ratio_unstable_thre,
                                  _min_point_num_neighborhood, _curvature_non_max_radius, keypointIndices->indices.data(), &k));
    keypointIndices->indices.resize((size_t)k);
    std::cout << "Keypoint detection done (" << k << " keypoints)" << std::endl;
    return true;
  }

  // keypoint_detect.hpp:53-111: the ratio threshold is lowered in steps of 0.05 while more than 50000 keypoints come out
  bool keypointDetectionBasedOnCurvature_adaptive(const typename pcl::PointCloud<PointT>::Ptr& inputPointCloud, pcl::PointIndicesPtr& keypointIndices) {
    keypointIndices = pcl::PointIndicesPtr(new pcl::PointIndices());
    const int64_t m = (int64_t)inputPointCloud->points.size();
    keypointIndices->indices.assign((size_t)(m > 0 ? m : 1), 0);
    int64_t k = 0;
    detail::check(ghicp_keypoints_adaptive(detail::ctx(), detail::xyz(*inputPointCloud), m, detail::stride<PointT>(), _neighborhood_radius,
                                           _ratio_unstable_thre, _min_point_num_neighborhood, _curvature_non_max_radius, 50000, 5000,
                                           keypointIndices->indices.data(), &k, nullptr, nullptr));
    keypointIndices->indices.resize((size_t)k);
    return true;
  }

 private:
  float _neighborhood_radius, _ratio_unstable_thre;
  int _min_point_num_neighborhood;
  float _curvature_non_max_radius;
};
}  // namespace ghicp
#endif
