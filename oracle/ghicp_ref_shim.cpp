// TEST INFRASTRUCTURE: the reference's OWN code behind C entry points, so that tests can pin the oracle's restatement against it.
// Compiled (oracle/Makefile, target `ref`) from the sources where they lie under /root/reference:
//   * src/stereo_binary_feature.cpp           -- StereoBinaryFeature::hammingDistance (16-104), the feature dump format (107-148)
//   * include/fpfh.hpp                        -- FPFHfeature::compute_fpfh_distance (135-165)
//   * src/ghicp_reg.cpp, lines 114-341, 343-604 and 605-789 only -- calED, calFD_BSC, calFD_FPFH, calCD_NF/BSC/FPFH, findcorrespondenceKM
//     (graph build + Km::output filter), findcorrespondenceNNR/NN, adjustweight: plain C++ on vectors and an Eigen::MatrixX3d.  The
//     Makefile extracts those line ranges into oracle/_ref/ghicp_reg_members.inc at build time (a build intermediate in the
//     git-ignored output directory); the rest of that file needs PCL / VTK.
// PCL / Eigen / boost are the stand-in headers of oracle/ref_stubs (interface only).  Output only into oracle/_ref/.
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#define private public  // the pinned functions are private members; the shim sets and reads the state they work on
#define protected public
// one extra member is DECLARED in the reference's class so that the scalar tail of transformestimation (below) can be compiled in member
// scope: the declaration rides on the class's own `bool adjustweight();`
#define adjustweight() adjustweight(); bool transformestimation_scalar_tail(const Eigen::Matrix3d &R, const Eigen::Vector3d &t)
#include "ghicp_reg.h"
#undef adjustweight
#undef private
#undef protected

namespace ghicp {
#include "ghicp_reg_members.inc"

// The plain-C++ statements of GHRegistration::transformestimation (src/ghicp_reg.cpp:791-927) around its PCL call: min_cor / IoU
// (795-799), translation + Euler angles in degrees with pi = 3.1415926 (870-882), RMSE after the update (890, 899-907) and the convergence
// test (909-914), extracted by the Makefile.  R, t stand for what lines 860-864 take out of PCL's float 4x4; Spoint must already hold the
// transformed correspondences (lines 893-898 are Eigen expressions and stay with the restatement).
bool GHRegistration::transformestimation_scalar_tail(const Eigen::Matrix3d &R, const Eigen::Vector3d &t) {
#include "ghicp_reg_te_tail.inc"
  return converge;
}
}  // namespace ghicp

namespace {
struct Quiet {  // the reference prints from every function
  std::streambuf* old;
  std::ostringstream sink;
  Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { std::cout.rdbuf(old); }
};
ghicp::GHRegistration make(const double* kpS, int ks, const double* kpT, int kt, float bbx, int feature, int corr, int dof, float est_iou,
                           float ratio, float step) {
  Eigen::MatrixX3d S(ks, 3), T(kt, 3);
  for (int i = 0; i < ks; i++) for (int d = 0; d < 3; d++) S(i, d) = kpS[(size_t)i * 3 + d];
  for (int j = 0; j < kt; j++) for (int d = 0; d < 3; d++) T(j, d) = kpT[(size_t)j * 3 + d];
  ghicp::Keypoints kp;
  kp.setCoordinate(S, T);
  ghicp::Energyfunction ef;
  ef.init(ks, kt, bbx);  // ghicp_reg.h:26-41
  return ghicp::GHRegistration(kp, ef, (ghicp::FeatureType)feature, (ghicp::CorrespondenceType)corr, 1.5f, ratio, step, dof, est_iou);
}
int index_of(const Eigen::MatrixX3d& M, const Eigen::MatrixX3d& P, long r) {
  for (long i = 0; i < M.rows(); i++)
    if (M(i, 0) == P(r, 0) && M(i, 1) == P(r, 1) && M(i, 2) == P(r, 2)) return (int)i;
  return -1;
}
}  // namespace

extern "C" {
// One GH-ICP iteration up to the correspondences, with the reference's own members (ghicp_reg.cpp:64-85 order):
// calED -> calCD_{NF,BSC,FPFH} -> findcorrespondence{KM,NN,NNR}.  feature: 0 BSC, 2 FPFH, 3 None; corr: 0 NN, 1 NNR, 2 KM (utility.h).
// Returns the number of correspondences; SP/TP (capacity max(ks,kt)) in the reference's emission order.
int ref_iter_step(const double* kpS, int ks, const double* kpT, int kt, const double* FD, int feature, int corr, float bbx, int it, double RMS,
                  double FDM, double FDstd, double para1, double para2, double* penalty, double* CD_out, int* SP, int* TP, double* rmse,
                  double* fdm, double* fdstd, double* energy) {
  Quiet q;
  ghicp::GHRegistration reg = make(kpS, ks, kpT, kt, bbx, feature, corr, 6, 0.6f, 1.1f, 0.1f);
  reg.iteration_number = it;
  reg.RMS = RMS; reg.FDM = FDM; reg.FDstd = FDstd;
  reg.EF.para1_penalty = para1; reg.EF.para2_penalty = para2;
  if (FD)
    for (int i = 0; i < ks; i++) for (int j = 0; j < kt; j++) reg.EF.FD[i][j] = FD[(size_t)i * kt + j];
  reg.calED();
  if (feature == ghicp::BSC) reg.calCD_BSC();
  else if (feature == ghicp::FPFH) reg.calCD_FPFH();
  else reg.calCD_NF();
  if (corr == ghicp::KM) reg.findcorrespondenceKM();  // writes Corres.txt into the cwd (km.cpp:147-148)
  else if (corr == ghicp::NNR) reg.findcorrespondenceNNR();
  else reg.findcorrespondenceNN();
  *penalty = reg.EF.penalty;
  if (CD_out)
    for (int i = 0; i < ks; i++) for (int j = 0; j < kt; j++) CD_out[(size_t)i * kt + j] = reg.EF.CD[i][j];
  const int cor = (int)reg.Spoint.rows();
  for (int c = 0; c < cor; c++) { SP[c] = index_of(reg.KP.kpSXYZ, reg.Spoint, c); TP[c] = index_of(reg.KP.kpTXYZ, reg.Tpoint, c); }
  *rmse = reg.RMS; *fdm = reg.FDM; *fdstd = reg.FDstd;
  if (energy) *energy = reg.energy.empty() ? 0.0 : reg.energy.back();
  return cor;
}

void ref_adjustweight(float est_iou, double IoU, float ratio, float step, double* para1, double* para2) {
  Quiet q;
  const double z[3] = {0, 0, 0};
  ghicp::GHRegistration reg = make(z, 1, z, 1, 1.f, ghicp::None, ghicp::NN, 6, est_iou, ratio, step);
  reg.IoU = IoU;
  reg.EF.para1_penalty = *para1; reg.EF.para2_penalty = *para2;
  reg.adjustweight();
  *para1 = reg.EF.para1_penalty; *para2 = reg.EF.para2_penalty;
}

// calFD_BSC (ghicp_reg.cpp:143-200): featS = V x ks x 56 bytes, featT = kt x 56 bytes, 441-bit strings.
void ref_fd_bsc(const unsigned char* featS, int ks, int V, const unsigned char* featT, int kt, int dof, double* FD) {
  Quiet q;
  const double z[3] = {0, 0, 0};
  std::vector<double> a((size_t)ks * 3, 0.0), b((size_t)kt * 3, 0.0);
  ghicp::GHRegistration reg = make(a.data(), ks, b.data(), kt, 1.f, ghicp::BSC, ghicp::NN, dof, 0.6f, 1.1f, 0.1f);
  (void)z;
  ghicp::doubleVectorSBF bS(4), bT(4);
  for (int v = 0; v < 4; v++) {
    for (int i = 0; i < ks; i++) {
      ghicp::StereoBinaryFeature f(441);
      if (v < V) std::memcpy(f.feature_, featS + ((size_t)v * ks + i) * 56, 56);
      bS[v].push_back(f);
    }
    if (v == 0)
      for (int j = 0; j < kt; j++) {
        ghicp::StereoBinaryFeature f(441);
        std::memcpy(f.feature_, featT + (size_t)j * 56, 56);
        bT[0].push_back(f);
      }
  }
  reg.KP.setBSCfeature(bS, bT);
  reg.calFD_BSC();
  for (int i = 0; i < ks; i++) for (int j = 0; j < kt; j++) FD[(size_t)i * kt + j] = reg.EF.FD[i][j];
}

int ref_hamming(const unsigned char* a, const unsigned char* b, int nbits) {
  ghicp::StereoBinaryFeature fa(nbits), fb(nbits), tool;
  std::memcpy(fa.feature_, a, fa.byte_);
  std::memcpy(fb.feature_, b, fb.byte_);
  return tool.hammingDistance(fa, fb);
}

float ref_fpfh_distance(const float* h1, const float* h2) {
  float a[33], b[33];
  std::memcpy(a, h1, sizeof(a));
  std::memcpy(b, h2, sizeof(b));
  ghicp::FPFHfeature<pcl::PointXYZ> f(1.0);
  return f.compute_fpfh_distance(a, b);
}

// See transformestimation_scalar_tail above.  Rt16: the iteration's 4x4 (row-major, f32 values); SpA: correspondences AFTER the update,
// Tp: their targets (c x 3).  Returns converge; out3 = {IoU, RMSE after, converge}.
int ref_te_tail(const double* Rt16, const double* SpA, const double* Tp, int c, int ks, int kt, int min_cor, float conv_t, float conv_r, double* out3) {
  Quiet q;
  std::vector<double> a((size_t)ks * 3, 0.0), b((size_t)kt * 3, 0.0);
  ghicp::GHRegistration reg = make(a.data(), ks, b.data(), kt, 1.f, ghicp::None, ghicp::NN, 6, 0.6f, 1.1f, 0.1f);
  reg.EF.min_cor = min_cor;
  reg.converge_t_ = conv_t; reg.converge_r_ = conv_r;
  reg.Spoint.resize(c, 3); reg.Tpoint.resize(c, 3);
  for (int i = 0; i < c; i++) for (int d = 0; d < 3; d++) { reg.Spoint(i, d) = SpA[(size_t)i * 3 + d]; reg.Tpoint(i, d) = Tp[(size_t)i * 3 + d]; }
  Eigen::Matrix3d R;
  Eigen::Vector3d t;
  for (int r = 0; r < 3; r++) { for (int q2 = 0; q2 < 3; q2++) R(r, q2) = Rt16[r * 4 + q2]; t(r) = Rt16[r * 4 + 3]; }
  const bool cv = reg.transformestimation_scalar_tail(R, t);
  out3[0] = reg.IoU; out3[1] = reg.rmseafter.back(); out3[2] = cv ? 1.0 : 0.0;
  return cv ? 1 : 0;
}

// CloudUtility::getCloudBound (utility.h:153-183) and the bbx magnitude of test/ghicp_main.cpp:93 (that one line is extracted by the
// Makefile into _ref/main_bbx.inc): the scale of the Euclidean term, 0.005 * bbx (ghicp_reg.h:40).
float ref_bbx_magnitude(const float* xyz, int n, int stride) {
  pcl::PointCloud<pcl::PointXYZ> cloud;
  cloud.points.resize(n);
  for (int i = 0; i < n; i++) { cloud.points[i].x = xyz[(size_t)i * stride]; cloud.points[i].y = xyz[(size_t)i * stride + 1]; cloud.points[i].z = xyz[(size_t)i * stride + 2]; }
  ghicp::CloudUtility<pcl::PointXYZ> cu;
  ghicp::Bounds s_cloud_bbx;
  cu.getCloudBound(cloud, s_cloud_bbx);
#include "main_bbx.inc"
  return bbx_magnitude;
}

// StereoBinaryFeature::writeFeatures / readFeatures (stereo_binary_feature.cpp:107-148): the on-disk dump format
int ref_sbf_write(const char* path, const unsigned char* feat, int k) {
  std::vector<ghicp::StereoBinaryFeature> v;
  for (int i = 0; i < k; i++) {
    ghicp::StereoBinaryFeature f(441);
    std::memcpy(f.feature_, feat + (size_t)i * 56, 56);
    v.push_back(f);
  }
  ghicp::StereoBinaryFeature tool;
  tool.writeFeatures(v, path);
  return 0;
}
}  // extern "C"
