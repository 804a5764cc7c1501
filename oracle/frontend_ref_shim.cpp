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
//                                          947-989  computeEigenVectorsByWeightPCA up to its Eigen::EigenSolver call: f64 centroid, weights
//                                                   sqrt(2) R - distance, the float 3 x 3 running sums, the division by the weight sum
//                                          939-1035, 119-155  the whole computeEigenVectorsByWeightPCA and computeLocalCoordinateSystem over a stand-in
//                                                   Eigen::EigenSolver that returns INJECTED eigenpairs: the selection of the principal / normal
//                                                   directions (strict compares, first index on ties), middle = principal x normal, x / y / z
//                                                   axes, their normalisation
//                                          1156-1165 Comput3DDistanceBetweenPoints
//   include/filter.hpp                     18-88    CFilter::IDPair, CFilter::voxelfilter (std::sort, Q1 phantom entries)
//   include/pca.h                          16-45    eigenValue, eigenVector, pcaFeature
//   include/keypoint_detect.hpp            132-147  CKeypointDetect::pruneUnstablePoints
//                                          119-130, 149-191  cmpBasedOnCurvature, nonMaximaSuppression (std::sort + std::set logic; the
//                                          radius search behind it is the stand-in KdTreeFLANN of oracle/ref_stubs: exact, d^2 < r^2)
//                                          60-107   keypointDetectionBasedOnCurvature_adaptive after its PCA call (the threshold loop)
//   include/pca.h                          228-239  CalculatePcaFeature: eigenvalues -> lamada1..3 and the curvature formula
//   include/fpfh.hpp                       93-115   FPFHfeature::keyfpfh (the keypoints' histogram rows)
//   src/common_reg.cpp                     302-313  CRegistration::calOverlap (counting + ratio; the radius search is the stand-in)
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
 public:
#include "bfe_wpca.inc"
#include "bfe_lcs.inc"
  // computeEigenVectorsByWeightPCA (bfe:940-1035) up to the Eigen::EigenSolver call: returns the covariance the solver would be given
  bool weightedCovariance(const typename pcl::PointCloud<PointT>::Ptr& input_cloud, const vector<int>& search_indices, int test_index,
                          Eigen::Matrix<float, 3, 3>& out) {
#include "bfe_wcov.inc"
    out = covariance;
    return true;
  }
  float Comput3DDistanceBetweenPoints(const PointT& pt1, const PointT& pt2) {  // bfe:1156-1165 (closing brace of the class follows it there)
    float dertax, dertay, dertaz, dis;
    dertax = pt1.x - pt2.x;
    dertay = pt1.y - pt2.y;
    dertaz = pt1.z - pt2.z;
    dis = (dertax) * (dertax) + (dertay) * (dertay) + (dertaz) * (dertaz);
    dis = sqrt(dis);
    return dis;
  }
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
  // keypointDetectionBasedOnCurvature_adaptive (kd:53-111) after its PCA call: `features` come from the caller
  // (the literal thresholds 50000 / 5000 of kd:83-106 become kUpper / kLower at extraction, so that the loop can be entered at test scale)
  int kUpper = 50000, kLower = 5000;
  bool adaptiveTail(std::vector<pcaFeature>& features, pcl::PointIndicesPtr& keypointIndices) {
#include "kd_adaptive_tail.inc"
    return true;
  }
  int _min_point_num_neighborhood;
  float _curvature_non_max_radius;
  float _ratio_unstable_thre = 0.65f;
};

// FPFHfeature::keyfpfh (fpfh.hpp:93-115) in a class that declares nothing else
struct FpfhRow { float histogram[33]; };
struct FpfhCloud { unsigned width = 0, height = 0; std::vector<FpfhRow> points; };
typedef std::shared_ptr<FpfhCloud> fpfhFeaturePtr;
struct FPFHfeatureKey {
#include "fpfh_keyfpfh.inc"
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

// computeEigenVectorsByWeightPCA's covariance (bfe:947-989): neighbours = idx[0..cnt) (the radius search's order), test point = test_index.
int ref_weighted_cov(const float* xyz, int n, const int* idx, int cnt, int test_index, float R, float* out9) {
  Quiet q;
  Enc e(R, 7, false);
  pcl::PointCloud<pcl::PointXYZ>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZ>);
  cloud->points.resize(n);
  for (int i = 0; i < n; i++) { cloud->points[i].x = xyz[(size_t)i * 3]; cloud->points[i].y = xyz[(size_t)i * 3 + 1]; cloud->points[i].z = xyz[(size_t)i * 3 + 2]; }
  std::vector<int> si(idx, idx + cnt);
  Eigen::Matrix<float, 3, 3> c;
  if (!e.weightedCovariance(cloud, si, test_index, c)) return 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) out9[i * 3 + j] = c(i, j);
  return 1;
}

// computeLocalCoordinateSystem (bfe:121-155) = computeEigenVectorsByWeightPCA (bfe:940-1035) + axis construction, with the eigen solver's
// output injected by the caller (values[3], vectors 3 x 3 row-major with eigenvectors in columns).  out12: x, y, z axes and the origin.
int ref_lcs(const float* xyz, int n, const int* idx, int cnt, int test_index, float R, const float* values3, const float* vectors9, float* out12) {
  Quiet q;
  Enc e(R, 7, false);
  pcl::PointCloud<pcl::PointXYZ>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZ>);
  cloud->points.resize(n);
  for (int i = 0; i < n; i++) { cloud->points[i].x = xyz[(size_t)i * 3]; cloud->points[i].y = xyz[(size_t)i * 3 + 1]; cloud->points[i].z = xyz[(size_t)i * 3 + 2]; }
  std::vector<int> si(idx, idx + cnt);
  auto& inj = Eigen::EigenSolver<Eigen::Matrix3f>::injected();
  for (int i = 0; i < 3; i++) inj.values[i] = values3[i];
  for (int i = 0; i < 9; i++) inj.vectors[i] = vectors9[i];
  Enc::CoordinateSystem cs;
  if (!e.computeLocalCoordinateSystem(cloud, test_index, si, cs)) return 0;
  const Eigen::Vector3f* ax[4] = {&cs.xAxis, &cs.yAxis, &cs.zAxis, &cs.origin};
  for (int a = 0; a < 4; a++)
    for (int d = 0; d < 3; d++) out12[a * 3 + d] = (*ax[a])(d);
  return 1;
}

// pca.h:233-244: eigenvalues -> pcaFeature values + curvature
double ref_pca_curvature(float l1, float l2, float l3) {
  ghicp::pcaFeature feature;
  Eigen::Vector3f eigen_values(l1, l2, l3);
#include "pca_curvature.inc"
  return feature.curvature;
}

// keypointDetectionBasedOnCurvature_adaptive (kd:53-111) on given PCA features (lam m x 3, count, curvature, xyz); returns the keypoints
int ref_adaptive_tail(const float* xyz, const float* lam, const int* count, const double* curvature, int m, int min_n, float r_nms, float ratio, int upper,
                      int lower, int* out_ids) {
  std::vector<ghicp::pcaFeature> feats((size_t)m);
  for (int i = 0; i < m; i++) {
    feats[i].pt.x = xyz[(size_t)i * 3]; feats[i].pt.y = xyz[(size_t)i * 3 + 1]; feats[i].pt.z = xyz[(size_t)i * 3 + 2];
    feats[i].values.lamada1 = lam[(size_t)i * 3]; feats[i].values.lamada2 = lam[(size_t)i * 3 + 1]; feats[i].values.lamada3 = lam[(size_t)i * 3 + 2];
    feats[i].ptNum = count[i]; feats[i].curvature = curvature[i]; feats[i].ptId = i;
  }
  pcl::PointIndicesPtr idx(new pcl::PointIndices);
  ghicp::CKeypointDetect<pcl::PointXYZ> kd(min_n, r_nms);
  kd._ratio_unstable_thre = ratio; kd.kUpper = upper; kd.kLower = lower;
  kd.adaptiveTail(feats, idx);
  for (size_t i = 0; i < idx->indices.size(); i++) out_ids[i] = idx->indices[i];
  return (int)idx->indices.size();
}

// FPFHfeature::keyfpfh (fpfh.hpp:93-115): rows of the keypoints (source side only is looked at; the target side gets the same input)
void ref_keyfpfh(const float* hist, int m, const int* kp, int k, float* out) {
  ghicp::fpfhFeaturePtr s(new ghicp::FpfhCloud), t(new ghicp::FpfhCloud), ks(new ghicp::FpfhCloud), kt(new ghicp::FpfhCloud);
  s->points.resize(m);
  for (int i = 0; i < m; i++) std::memcpy(s->points[i].histogram, hist + (size_t)i * 33, 33 * sizeof(float));
  *t = *s;
  pcl::PointIndicesPtr si(new pcl::PointIndices), ti(new pcl::PointIndices);
  si->indices.assign(kp, kp + k);
  ti->indices.assign(kp, kp + k);
  ghicp::FPFHfeatureKey f;
  f.keyfpfh(s, t, si, ti, ks, kt);
  for (int i = 0; i < k; i++) std::memcpy(out + (size_t)i * 33, ks->points[i].histogram, 33 * sizeof(float));
}

// CRegistration::calOverlap (common_reg.cpp:294-317): counting loop and ratio over the stand-in exact radius search
float ref_cal_overlap(const float* c1, int n1, const float* c2, int n2, float thre_dis) {
  typedef pcl::PointXYZ PointT;
  pcl::PointCloud<PointT>::Ptr Cloud1(new pcl::PointCloud<PointT>), Cloud2(new pcl::PointCloud<PointT>);
  Cloud1->points.resize(n1); Cloud2->points.resize(n2);
  for (int i = 0; i < n1; i++) { Cloud1->points[i].x = c1[(size_t)i * 3]; Cloud1->points[i].y = c1[(size_t)i * 3 + 1]; Cloud1->points[i].z = c1[(size_t)i * 3 + 2]; }
  for (int i = 0; i < n2; i++) { Cloud2->points[i].x = c2[(size_t)i * 3]; Cloud2->points[i].y = c2[(size_t)i * 3 + 1]; Cloud2->points[i].z = c2[(size_t)i * 3 + 2]; }
  int overlap_point_num = 0;
  float overlap_ratio;
#include "reg_overlap.inc"
  return overlap_ratio;
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
