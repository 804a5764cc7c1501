// Drop-in for include/common_reg.h (+ src/common_reg.cpp): ghicp::CRegistration<PointT>, the fine-registration
// wrappers the reference builds on PCL's ICP family.  Same class / method names and argument meaning; the work runs on
// the GPU through the C ABI (ghicp_icp, ghicp_cal_overlap, ghicp_transform_cloud_f32, ghicp_inv_transform).
//   icp_reg        common_reg.cpp:45-107    point-to-point, closed-form SVD per iteration
//   ptplicp_reg    common_reg.cpp:122-199   point-to-plane LLS; normals by k-NN PCA (covariance_K <= 20)
//   calOverlap     common_reg.cpp:294-317
//   transformcloud common_reg.cpp:325-349
//   invTransform   common_reg.cpp:357-370   (R^T with the negated translation -- "Not Mathimatically" an inverse)
// Differences a caller can observe: ptplicp_reg returns true (the reference falls off the end without a return);
// gicp_reg / Coarsereg_FPFHSAC / the control-point solvers are not part of the hot path and are not provided.
#ifndef GHICP_DROPIN_COMMON_REG_H_
#define GHICP_DROPIN_COMMON_REG_H_
#include <ctime>
#include <iostream>

#include "utility.h"

namespace ghicp {
template <typename PointT> class CRegistration {
 public:
  bool icp_reg(const typename pcl::PointCloud<PointT>::Ptr& SourceCloud, const typename pcl::PointCloud<PointT>::Ptr& TargetCloud,
               typename pcl::PointCloud<PointT>::Ptr& TransformedSource, Eigen::Matrix4f& transformationS2T, int max_iter,
               bool use_reciprocal_correspondence, bool use_trimmed_rejector, float thre_dis, float min_overlap_for_reg) {
    return run(GHICP_ICP_POINT_TO_POINT, "Point-to-Point", SourceCloud, TargetCloud, TransformedSource, transformationS2T, max_iter,
               use_reciprocal_correspondence, use_trimmed_rejector, thre_dis, 0, min_overlap_for_reg);
  }

  bool ptplicp_reg(const typename pcl::PointCloud<PointT>::Ptr& SourceCloud, const typename pcl::PointCloud<PointT>::Ptr& TargetCloud,
                   typename pcl::PointCloud<PointT>::Ptr& TransformedSource, Eigen::Matrix4f& transformationS2T, int max_iter,
                   bool use_reciprocal_correspondence, bool use_trimmed_rejector, float thre_dis, int covariance_K, float min_overlap_for_reg) {
    return run(GHICP_ICP_POINT_TO_PLANE, "Point-to-Plane", SourceCloud, TargetCloud, TransformedSource, transformationS2T, max_iter,
               use_reciprocal_correspondence, use_trimmed_rejector, thre_dis, covariance_K, min_overlap_for_reg);
  }

  float calOverlap(const typename pcl::PointCloud<PointT>::Ptr& Cloud1, const typename pcl::PointCloud<PointT>::Ptr& Cloud2, float thre_dis) {
    float ratio = 0.f;
    detail::check(ghicp_cal_overlap(detail::ctx(), detail::xyz(*Cloud1), (int64_t)Cloud1->points.size(), detail::stride<PointT>(), detail::xyz(*Cloud2),
                                    (int64_t)Cloud2->points.size(), detail::stride<PointT>(), thre_dis, &ratio));
    return ratio;
  }

  void transformcloud(typename pcl::PointCloud<PointT>::Ptr& Cloud, typename pcl::PointCloud<PointT>::Ptr& TransformedCloud, Eigen::Matrix4f& transformation) {
    float T[16];
    to_rows(transformation, T);
    const size_t n = Cloud->points.size();
    std::vector<float> out(n * 3 + 3);
    detail::check(ghicp_transform_cloud_f32(detail::ctx(), detail::xyz(*Cloud), (int64_t)n, detail::stride<PointT>(), T, out.data()));
    append_xyz(out, n, *TransformedCloud);  // the reference push_backs onto whatever the output already holds (:341-347)
    std::cout << "Transform done ..." << std::endl;
  }

  void invTransform(const Eigen::Matrix4f& transformation, Eigen::Matrix4f& invtransformation) {
    float T[16], I[16];
    to_rows(transformation, T);
    ghicp_inv_transform(T, I);
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) invtransformation(r, c) = I[r * 4 + c];
  }

  // statistics of the last icp_reg / ptplicp_reg call (the reference only logs them)
  ghicp_icp_stats last_stats = {};

 private:
  static void to_rows(const Eigen::Matrix4f& M, float* T) {
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) T[r * 4 + c] = M(r, c);
  }
  static void append_xyz(const std::vector<float>& xyz, size_t n, pcl::PointCloud<PointT>& cloud) {
    for (size_t i = 0; i < n; i++) {
      PointT pt = PointT();
      pt.x = xyz[i * 3]; pt.y = xyz[i * 3 + 1]; pt.z = xyz[i * 3 + 2];
      cloud.points.push_back(pt);
    }
    cloud.width = (unsigned)cloud.points.size();
    cloud.height = 1;
  }
  bool run(int metric, const char* name, const typename pcl::PointCloud<PointT>::Ptr& S, const typename pcl::PointCloud<PointT>::Ptr& T,
           typename pcl::PointCloud<PointT>::Ptr& out, Eigen::Matrix4f& S2T, int max_iter, bool reciprocal, bool trimmed, float thre_dis, int cov_k,
           float min_overlap) {
    const clock_t t0 = clock();
    ghicp_icp_params p;
    ghicp_icp_params_default(&p);
    p.max_iter = max_iter;
    p.use_reciprocal = reciprocal ? 1 : 0;
    p.use_trimmed = trimmed ? 1 : 0;
    p.metric = metric;
    p.thre_dis = thre_dis;
    p.min_overlap = min_overlap;
    if (metric == GHICP_ICP_POINT_TO_PLANE) p.covariance_k = cov_k;
    const size_t n = S->points.size();
    std::vector<float> xyz(n * 3 + 3);
    float T16[16];
    detail::check(ghicp_icp(detail::ctx(), detail::xyz(*S), (int64_t)n, detail::stride<PointT>(), detail::xyz(*T), (int64_t)T->points.size(),
                            detail::stride<PointT>(), &p, T16, xyz.data(), &last_stats));
    if (!last_stats.done) {
      std::cout << "The overlap ratio is too small. This registration would not be done." << std::endl;  // common_reg.cpp:68-69
      return false;
    }
    out->points.clear();  // icp.align(*TransformedSource) overwrites the output cloud
    append_xyz(xyz, n, *out);
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) S2T(r, c) = T16[r * 4 + c];
    std::cout << name << " ICP done in " << float(clock() - t0) / CLOCKS_PER_SEC << " s" << std::endl;
    for (int r = 0; r < 4; r++) std::cout << T16[r * 4] << " " << T16[r * 4 + 1] << " " << T16[r * 4 + 2] << " " << T16[r * 4 + 3] << std::endl;
    std::cout << "The fitness score of this registration is " << last_stats.fitness << std::endl;
    std::cout << "-----------------------------------------------------------------------------" << std::endl;
    return true;
  }
};
}  // namespace ghicp
#endif
