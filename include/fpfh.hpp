// Drop-in for include/fpfh.hpp: FPFHfeature<PointT> as test/ghicp_main.cpp:118-127 uses it -- compute_fpfh_feature (fpfh.hpp:36-58:
// pcl::NormalEstimation + pcl::FPFHEstimationOMP, both k = 20, over the whole down-sampled cloud) runs on the GPU through ghicp_fpfh;
// keyfpfh (fpfh.hpp:93-115) is the reference's own row gather over host vectors (device-resident callers use ghicp_fpfh_keypoints);
// compute_fpfh_distance (fpfh.hpp:135-165) is the reference's scalar formula for ONE pair of histograms -- GHRegistration evaluates the
// whole K_S x K_T matrix with ghicp_fd_fpfh (include/ghicp_reg.h).  fpfhalign (SAC-IA, fpfh.hpp:117-133) and compute_fpfh_keypoint
// (fpfh.hpp:60-91, unused by main and writing through an empty cloud) are outside the hot path.
#ifndef GHICP_DROPIN_FPFH_HPP_
#define GHICP_DROPIN_FPFH_HPP_
#include <cstring>
#include <iostream>
#include <vector>

#include "utility.h"

namespace ghicp {
template <typename PointT> class FPFHfeature {
 public:
  FPFHfeature(double neighbor_radius) : radius(neighbor_radius) {}

  bool compute_fpfh_feature(const typename pcl::PointCloud<PointT>::Ptr& input_cloud, fpfhFeaturePtr& fpfh) {
    const int64_t m = (int64_t)input_cloud->points.size();
    std::vector<float> hist((size_t)m * 33 + 1);
    detail::check(ghicp_fpfh(detail::ctx(), detail::xyz(*input_cloud), m, detail::stride<PointT>(), 20, 20, nullptr, hist.data()));
    fpfh->points.resize((size_t)m);
    fpfh->width = (unsigned)m;
    fpfh->height = 1;
    for (int64_t i = 0; i < m; i++) std::memcpy(fpfh->points[(size_t)i].histogram, &hist[(size_t)i * 33], 33 * sizeof(float));
    std::cout << "Extract FPFH feature done." << std::endl;
    return 1;
  }

  bool keyfpfh(const fpfhFeaturePtr& source_fpfh, const fpfhFeaturePtr& target_fpfh, const pcl::PointIndicesPtr& sindices,
               const pcl::PointIndicesPtr& tindices, fpfhFeaturePtr& source_kfpfh, fpfhFeaturePtr& target_kfpfh) {
    source_kfpfh->width = (unsigned)sindices->indices.size();
    source_kfpfh->height = 1;
    target_kfpfh->width = (unsigned)tindices->indices.size();
    target_kfpfh->height = 1;
    source_kfpfh->points.resize(sindices->indices.size());
    target_kfpfh->points.resize(tindices->indices.size());
    for (size_t i = 0; i < sindices->indices.size(); i++) source_kfpfh->points[i] = source_fpfh->points[(size_t)sindices->indices[i]];
    for (size_t i = 0; i < tindices->indices.size(); i++) target_kfpfh->points[i] = target_fpfh->points[(size_t)tindices->indices[i]];
    return 1;
  }

  // |correlation| of two 33-bin histograms, three sequential f32 sums (fpfh.hpp:135-165; a constant histogram gives NaN as there):
  // the 1 x 1 case of the matrix kernel GHRegistration uses, so there is one implementation of the formula
  float compute_fpfh_distance(float his1[33], float his2[33]) {
    float d = 0.f;
    detail::check(ghicp_fd_fpfh(detail::ctx(), his1, 1, his2, 1, &d));
    return d;
  }

 private:
  double radius;
};
}  // namespace ghicp
#endif
