// TEST INFRASTRUCTURE: more of the reference's OWN code behind C entry points (see ghicp_ref_shim.cpp).  The member functions below
// use nothing of PCL / Eigen beyond plain containers, but live in class templates whose other members do; the Makefile (target `ref`)
// extracts their line ranges from the headers where they lie into the git-ignored oracle/_ref/ at build time, and this file gives them
// a class to live in that declares the same data members (themselves extracted: binary_feature_extraction.hpp:37-60):
//   include/binary_feature_extraction.hpp  62-117   BSCEncoder constructor (sample pattern from rand(), or sample_pattern.txt)
//                                          196-373  constructCubicGrid (the Gaussian-weighted 3 x 7 x 7 cells; its 2-D radius searches go to the
//                                                   stand-in exact KdTreeFLANN of oracle/ref_stubs: ascending distance, ties by index)
//                                          463-565  computeFeatureProjectedGridAndCompareFeature2D (occupancy + depth / density bits)
//                                          678-758  ReArrangeGrid / ReArrange_2D and the three re-arrangements
//                                          839-872  getVoxelNum, getVoxelIndex, contain2DPair
//   include/filter.hpp                     18-88    CFilter::IDPair, CFilter::voxelfilter (std::sort, Q1 phantom entries)
//   include/pca.h                          16-45    eigenValue, eigenVector, pcaFeature
//   include/keypoint_detect.hpp            132-147  CKeypointDetect::pruneUnstablePoints
//                                          119-130, 149-191  cmpBasedOnCurvature, nonMaximaSuppression (std::sort + std::set logic; the
//                                          radius search behind it is the stand-in KdTreeFLANN of oracle/ref_stubs: exact, d^2 < r^2)
// Output only into oracle/_ref/.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <algorithm>
#include <set>
#include <vector>

#include "stereo_binary_feature.h"  // the reference's own (through the stand-in PCL / Eigen / boost headers)
#include <pcl/point_types.h>

using namespace std;

namespace ghicp {
template <typename PointT>
class BSCEncoder : public StereoBinaryFeature {
 public:
#include "bfe_fields.inc"
#include "bfe_ctor.inc"
#include "bfe_grid.inc"
#include "bfe_binarize.inc"
#include "bfe_rearrange.inc"
#include "bfe_private.inc"
};

template <typename PointT>
class CFilter {
 public:
#include "filter_voxel.inc"
};

#include "pca_structs.inc"

template <typename PointT>
class CKeypointDetect {
 public:
  explicit CKeypointDetect(int min_n, float r_nms = 0.f) : _min_point_num_neighborhood(min_n), _curvature_non_max_radius(r_nms) {}
#include "kd_prune.inc"
#include "kd_cmp.inc"
#include "kd_nms.inc"
  int _min_point_num_neighborhood;
  float _curvature_non_max_radius;
};
}  // namespace ghicp

namespace {
struct Quiet {
  std::streambuf* old;
  std::ostringstream sink;
  Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { std::cout.rdbuf(old); }
};
typedef ghicp::BSCEncoder<pcl::PointXYZ> Enc;
}  // namespace

extern "C" {
// The 49 (a, b) pairs `BSCEncoder(R, 7, true)` draws; srand(1) first = the state of a fresh process (the reference never seeds).
// Writes sample_pattern.txt into the cwd like the reference.
void ref_bsc_pattern(int* out98) {
  srand(1);
  Enc e(1.5f, 7, true);
  for (int i = 0; i < 49; i++) { out98[2 * i] = e.grid_index_pairs_2d_[i].first; out98[2 * i + 1] = e.grid_index_pairs_2d_[i].second; }
}

// The strings of one keypoint from its 147 cells (normalized_point_weight, average_depth), built by the reference's own
// computeFeatureProjectedGridAndCompareFeature2D / ReArrangeGrid in the call pattern of extractBinaryFeatureOfKeypoint (bfe:782-828).
// out: 4 x 56 bytes (unused variants zero).
void ref_bsc_strings(const float* weight147, const float* depth147, int dof, const int* pattern98, unsigned char* out) {
  Quiet q;
  Enc e(1.5f, 7, false);  // reads a (missing) sample_pattern.txt: pairs are then set explicitly below
  e.grid_index_pairs_2d_.resize(49);
  for (int i = 0; i < 49; i++) e.grid_index_pairs_2d_[i] = std::pair<int, int>(pattern98[2 * i], pattern98[2 * i + 1]);
  std::memset(out, 0, 4 * 56);
  std::vector<Enc::GridVoxel> grid_1(e.gridFeatureDimension_);
  for (int i = 0; i < 147; i++) { grid_1[i].normalized_point_weight = weight147[i]; grid_1[i].average_depth = depth147[i]; }
  ghicp::StereoBinaryFeature f = e.computeFeatureProjectedGridAndCompareFeature2D(grid_1);
  std::memcpy(out, f.feature_, 56);
  static const int TR[4][3] = {{0, 0, 0}, {1, 2, 2}, {3, 2, 1}, {2, 1, 3}};  // bfe:795, 808, 817
  const int nvar = dof > 4 ? 4 : (dof > 0 ? 2 : 1);
  for (int v = 1; v < nvar; v++) {
    std::vector<Enc::GridVoxel> grid_k(e.gridFeatureDimension_);  // 147 zero cells; ReArrangeGrid APPENDS (Q3)
    e.ReArrangeGrid(grid_1, grid_k, TR[v][0], TR[v][1], TR[v][2]);
    f = e.computeFeatureProjectedGridAndCompareFeature2D(grid_k);
    std::memcpy(out + v * 56, f.feature_, 56);
  }
}

// constructCubicGrid (bfe:196-373) on a neighbourhood already rotated into the LCS: 147 normalized weights + 147 average depths.
void ref_cubic_grid(const float* loc, int n, float R, float* cells294) {
  Quiet q;
  Enc e(R, 7, false);
  pcl::PointCloud<pcl::PointXYZ>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZ>);
  cloud->points.resize(n);
  for (int i = 0; i < n; i++) { cloud->points[i].x = loc[(size_t)i * 3]; cloud->points[i].y = loc[(size_t)i * 3 + 1]; cloud->points[i].z = loc[(size_t)i * 3 + 2]; }
  std::vector<Enc::GridVoxel> grid(e.gridFeatureDimension_);
  e.constructCubicGrid(cloud, grid);
  for (int i = 0; i < 147; i++) { cells294[i] = grid[i].normalized_point_weight; cells294[147 + i] = grid[i].average_depth; }
}

// CFilter::voxelfilter (filter.hpp:28-88).  Returns the number of kept points; out (capacity n + 1) receives their xyz.
int ref_voxelfilter(const float* xyz, int n, int stride, float voxel, float* out_xyz) {
  Quiet q;
  pcl::PointCloud<pcl::PointXYZ>::Ptr in(new pcl::PointCloud<pcl::PointXYZ>), out(new pcl::PointCloud<pcl::PointXYZ>);
  in->points.resize(n);
  for (int i = 0; i < n; i++) { in->points[i].x = xyz[(size_t)i * stride]; in->points[i].y = xyz[(size_t)i * stride + 1]; in->points[i].z = xyz[(size_t)i * stride + 2]; }
  ghicp::CFilter<pcl::PointXYZ> f;
  f.voxelfilter(in, out, voxel);
  for (size_t i = 0; i < out->points.size(); i++) { out_xyz[i * 3] = out->points[i].x; out_xyz[i * 3 + 1] = out->points[i].y; out_xyz[i * 3 + 2] = out->points[i].z; }
  return (int)out->points.size();
}

// CKeypointDetect::pruneUnstablePoints (keypoint_detect.hpp:132-147).  lam: m x 3 (lamada1..3, stored as double like pca.h:243-245).
int ref_prune(const float* lam, const int* count, int m, float ratio_max, int min_n, int* out_idx) {
  std::vector<ghicp::pcaFeature> feats((size_t)m);
  for (int i = 0; i < m; i++) {
    feats[i].values.lamada1 = lam[(size_t)i * 3]; feats[i].values.lamada2 = lam[(size_t)i * 3 + 1]; feats[i].values.lamada3 = lam[(size_t)i * 3 + 2];
    feats[i].ptNum = count[i];
  }
  pcl::PointIndicesPtr idx(new pcl::PointIndices);
  ghicp::CKeypointDetect<pcl::PointXYZ> kd(min_n);
  kd.pruneUnstablePoints(feats, ratio_max, idx);
  for (size_t i = 0; i < idx->indices.size(); i++) out_idx[i] = idx->indices[i];
  return (int)idx->indices.size();
}

// CKeypointDetect::nonMaximaSuppression (keypoint_detect.hpp:149-191) over candidates (xyz f32, curvature f64, original ids).
int ref_nms(const float* xyz, const double* curvature, const int* ids, int c, float radius, int* out_ids) {
  std::vector<ghicp::pcaFeature> feats((size_t)c);
  for (int i = 0; i < c; i++) {
    feats[i].pt.x = xyz[(size_t)i * 3]; feats[i].pt.y = xyz[(size_t)i * 3 + 1]; feats[i].pt.z = xyz[(size_t)i * 3 + 2];
    feats[i].curvature = curvature[i];
    feats[i].ptId = ids[i];
  }
  pcl::PointIndicesPtr idx(new pcl::PointIndices);
  ghicp::CKeypointDetect<pcl::PointXYZ> kd(20, radius);
  if (c > 0) kd.nonMaximaSuppression(feats, idx);
  for (size_t i = 0; i < idx->indices.size(); i++) out_ids[i] = idx->indices[i];
  return (int)idx->indices.size();
}
}  // extern "C"
