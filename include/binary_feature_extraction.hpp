// Drop-in for include/binary_feature_extraction.hpp: BSCEncoder<PointT>::extractBinaryFeatures on the GPU.
#ifndef GHICP_DROPIN_BFE_HPP_
#define GHICP_DROPIN_BFE_HPP_
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <utility>
#include <vector>

#include "stereo_binary_feature.h"
#include "utility.h"

namespace ghicp {
template <typename PointT> class BSCEncoder : public StereoBinaryFeature {
 public:
  float extract_radius_;
  unsigned int voxel_side_num_;
  std::vector<std::pair<int, int>> grid_index_pairs_2d_;

  // bfe:63-117: the sample pattern is generated from rand() (and written) or read from ./sample_pattern.txt;
  // a missing file leaves every pair (0,0) exactly like the reference.
  BSCEncoder(float extract_radius, unsigned int voxel_side_num, bool build_sample_pattern = false)
      : extract_radius_(extract_radius), voxel_side_num_(voxel_side_num) {
    if (voxel_side_num_ != 7) throw std::runtime_error("BSCEncoder: the GPU encoder implements the 7x7 grid used by the reference's main()");
    const int cells = 49;
    if (build_sample_pattern) {
      for (int i = 0; i < cells; i++) {
        int a, b;
        do { a = rand() % cells; b = rand() % cells; } while (a == b || contains(a, b));
        grid_index_pairs_2d_.push_back(std::pair<int, int>(a, b));
      }
      std::ofstream fout("sample_pattern.txt");
      for (auto& p : grid_index_pairs_2d_) fout << p.first << " " << p.second << std::endl;
    } else {
      std::ifstream fin("sample_pattern.txt");
      grid_index_pairs_2d_.assign(cells, std::pair<int, int>(0, 0));
      for (auto& p : grid_index_pairs_2d_) fin >> p.first >> p.second;
    }
  }

  void extractBinaryFeatures(const typename pcl::PointCloud<PointT>::Ptr& input_cloud, const pcl::PointIndicesPtr& indices, int dof_type,
                             doubleVectorSBF& bscFeatures) {
    const int64_t K = (int64_t)indices->indices.size();
    if (K <= 0) { std::cout << "The input indice is NaN\n"; return; }
    std::vector<int32_t> pat(98);
    for (int i = 0; i < 49; i++) { pat[2 * i] = grid_index_pairs_2d_[i].first; pat[2 * i + 1] = grid_index_pairs_2d_[i].second; }
    std::vector<uint8_t> feat((size_t)4 * K * 56);
    std::vector<float> lcs((size_t)K * 12);
    detail::check(ghicp_bsc_encode(detail::ctx(), detail::xyz(*input_cloud), (int64_t)input_cloud->points.size(), detail::stride<PointT>(),
                                   indices->indices.data(), K, extract_radius_, dof_type, pat.data(), feat.data(), lcs.data()));
    const int nvar = dof_type > 4 ? 4 : (dof_type > 0 ? 2 : 1);
    for (int v = 0; v < 4; v++) {
      vectorSBF col((size_t)K);  // unused variants keep size-0 features (bfe:619-620)
      if (v < nvar)
        for (int64_t k = 0; k < K; k++) {
          StereoBinaryFeature f(441);
          std::memcpy(f.feature_, &feat[((size_t)v * K + k) * 56], 56);
          f.keypointIndex_ = (size_t)k;
          const float sx = (v == 1 || v == 3) ? -1.f : 1.f, sy = (v == 1 || v == 2) ? -1.f : 1.f, sz = (v >= 2) ? -1.f : 1.f;  // bfe:786-824
          for (int d = 0; d < 3; d++) {
            f.localSystem_.xAxis(d) = sx * lcs[(size_t)k * 12 + d]; f.localSystem_.yAxis(d) = sy * lcs[(size_t)k * 12 + 3 + d];
            f.localSystem_.zAxis(d) = sz * lcs[(size_t)k * 12 + 6 + d]; f.localSystem_.origin(d) = lcs[(size_t)k * 12 + 9 + d];
          }
          col[(size_t)k] = f;
        }
      bscFeatures.push_back(col);
    }
    std::cout << "Extract BSC feature done." << std::endl;
  }

 private:
  bool contains(int a, int b) const {
    for (auto& p : grid_index_pairs_2d_) if ((p.first == a && p.second == b) || (p.first == b && p.second == a)) return true;
    return false;
  }
};
}  // namespace ghicp
#endif
