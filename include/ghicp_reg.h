// Drop-in for include/ghicp_reg.h + src/ghicp_reg.cpp: same Energyfunction / Keypoints / GHRegistration
// surface (ctor arguments, set_raw_pointcloud, set_viewer, ghicp_reg, public result members); the loop runs on
// the GPU through ghicp_fd_bsc / ghicp_register.  No PCLVisualizer is created (ghicp_reg.cpp:28).
#ifndef GHICP_DROPIN_GHICP_REG_H_
#define GHICP_DROPIN_GHICP_REG_H_
#include <cstring>
#include <vector>

#include "km.h"
#include "stereo_binary_feature.h"
#include "utility.h"

namespace ghicp {
struct Energyfunction {  // ghicp_reg.h:15-42
  std::vector<std::vector<double>> ED, FD, CD;  // declared as in the reference; the GPU path never materialises them (init() leaves them empty)
  int weight_changing_rate;
  double penalty, para1_penalty, para2_penalty, penalty_initial;
  int min_cor;
  double KM_eps;
  float scale;
  float bbx_magnitude_;
  Energyfunction() : weight_changing_rate(6), penalty(0), para1_penalty(1), para2_penalty(1), penalty_initial(2), min_cor(10), KM_eps(0.01), scale(0), bbx_magnitude_(0) {}
  void init(int, int, float bbx_magnitude) {
    penalty_initial = 2.0; para1_penalty = 1.0; para2_penalty = 1.0; min_cor = 10; weight_changing_rate = 6; KM_eps = 0.01;
    scale = 0.005 * bbx_magnitude;
    bbx_magnitude_ = bbx_magnitude;
  }
};

struct Keypoints {  // ghicp_reg.h:44-72
  int kps_num = 0, kpt_num = 0;
  Eigen::MatrixX3d kpSXYZ, kpTXYZ;
  doubleVectorSBF bscS, bscT;
  fpfhFeaturePtr fpfhS, fpfhT;
  void setCoordinate(Eigen::MatrixX3d& kps, Eigen::MatrixX3d& kpt) { kpSXYZ = kps; kpTXYZ = kpt; kps_num = (int)kpSXYZ.rows(); kpt_num = (int)kpTXYZ.rows(); }
  void setBSCfeature(const doubleVectorSBF& s, const doubleVectorSBF& t) { bscS = s; bscT = t; }
  void setFPFHfeature(const fpfhFeaturePtr& fpfh_S, const fpfhFeaturePtr& fpfh_T) { fpfhS = fpfh_S; fpfhT = fpfh_T; }
};

class GHRegistration {
 public:
  GHRegistration(Keypoints Kp, Energyfunction Ef, FeatureType Ft, CorrespondenceType Ct, float radiusNonMax, float weight_adjustment_ratio,
                 float weight_adjustment_step, int dof_type, float estimated_IoU, float converge_tran = 0.02, float converge_rot = 0.02,
                 int ite = 0, int ite2 = 0)
      : KP(Kp), EF(Ef), Ft_(Ft), Ct_(Ct) {
    (void)ite; (void)ite2;
    ghicp_params_default(&P);
    P.feature = (int)Ft; P.corr = (int)Ct; P.dof = dof_type; P.radius_nonmax = radiusNonMax; P.adjust_ratio = weight_adjustment_ratio;
    P.adjust_step = weight_adjustment_step; P.est_iou = estimated_IoU; P.converge_t = converge_tran; P.converge_r = converge_rot;
    P.bbx_magnitude = Ef.bbx_magnitude_; P.penalty_initial = Ef.penalty_initial; P.para1 = Ef.para1_penalty; P.para2 = Ef.para2_penalty;
    P.km_eps = Ef.KM_eps; P.min_cor = Ef.min_cor; P.weight_changing_rate = Ef.weight_changing_rate;
    Rt_tillnow = Eigen::Matrix4d::Identity();
    RMS = 99999;
    gt_maxdis = radiusNonMax / 3;
  }
  void set_raw_pointcloud(const pcl::PointCloud<pcl::PointXYZI>::Ptr&, const pcl::PointCloud<pcl::PointXYZI>::Ptr&) {}  // viewer-only in the reference (:97)
  void set_viewer(bool) {}
  void set_max_iterations(int n) { P.max_iter = n; }  // the reference has no guard (ghicp_reg.cpp:49); default 200

  bool ghicp_reg(Eigen::Matrix4d& Rt_final) {  // ghicp_reg.cpp:24-112
    ghicp_ctx* c = detail::ctx();
    const int64_t ks = KP.kps_num, kt = KP.kpt_num;
    std::vector<uint16_t> fd16;
    std::vector<float> fd32;
    const void* FD = nullptr;
    if (Ft_ == BSC) {  // calFD_BSC :143-200
      const int V = P.dof == 6 ? 4 : 2;
      std::vector<uint8_t> fS((size_t)V * ks * 56), fT((size_t)kt * 56);
      for (int v = 0; v < V; v++) for (int64_t i = 0; i < ks; i++) std::memcpy(&fS[((size_t)v * ks + i) * 56], KP.bscS[v][i].feature_, 56);
      for (int64_t j = 0; j < kt; j++) std::memcpy(&fT[(size_t)j * 56], KP.bscT[0][j].feature_, 56);
      fd16.resize((size_t)ks * kt);
      detail::check(ghicp_fd_bsc(c, fS.data(), ks, V, fT.data(), kt, fd16.data()));
      FD = fd16.data();
    } else if (Ft_ == FPFH) {  // calFD_FPFH :202-214
      std::vector<float> hS((size_t)ks * 33), hT((size_t)kt * 33);  // pcl::FPFHSignature33 rows
      for (int64_t i = 0; i < ks; i++) std::memcpy(&hS[(size_t)i * 33], KP.fpfhS->points[(size_t)i].histogram, 33 * sizeof(float));
      for (int64_t j = 0; j < kt; j++) std::memcpy(&hT[(size_t)j * 33], KP.fpfhT->points[(size_t)j].histogram, 33 * sizeof(float));
      fd32.resize((size_t)ks * kt);
      detail::check(ghicp_fd_fpfh(c, hS.data(), ks, hT.data(), kt, fd32.data()));
      FD = fd32.data();
    }
    std::vector<ghicp_iter> trace((size_t)P.max_iter);
    std::vector<int32_t> ml((size_t)P.max_iter * (ks > 0 ? ks : 1));
    int32_t n_iter = 0;
    double Rt[16];
    // Eigen::MatrixX3d is column-major: the C ABI takes k x 3 row-major copies (the public interface only: operator())
    std::vector<double> kS((size_t)ks * 3), kT((size_t)kt * 3);
    for (int64_t i = 0; i < ks; i++) for (int d = 0; d < 3; d++) kS[(size_t)i * 3 + d] = KP.kpSXYZ(i, d);
    for (int64_t j = 0; j < kt; j++) for (int d = 0; d < 3; d++) kT[(size_t)j * 3 + d] = KP.kpTXYZ(j, d);
    detail::check(ghicp_register(c, &P, kS.data(), ks, kT.data(), kt, FD, Rt, trace.data(), &n_iter, ml.data()));
    for (int r = 0; r < 4; r++) for (int q = 0; q < 4; q++) Rt_tillnow(r, q) = Rt[r * 4 + q];
    Rt_final = Rt_tillnow;
    matchlist.assign((size_t)ks, std::vector<int>((size_t)n_iter));
    for (int it = 0; it < n_iter; it++) {
      energy.push_back(trace[it].energy); rmse.push_back(trace[it].rmse); rmseafter.push_back(trace[it].rmse_after); cor.push_back(trace[it].cor);
      for (int64_t i = 0; i < ks; i++) matchlist[(size_t)i][(size_t)it] = ml[(size_t)it * ks + i];
    }
    if (n_iter > 0) RMS = trace[n_iter - 1].rmse;
    iterations = n_iter;
    return 1;
  }

  double gt_maxdis, PCFD = 0, RMS;
  Eigen::Matrix4d Rt_tillnow, Rt_gt;
  std::vector<std::vector<int>> matchlist;
  std::vector<double> energy, rmse, rmseafter, pre, rec;
  std::vector<int> cor;
  int iterations = 0;

 private:
  Keypoints KP;
  Energyfunction EF;
  FeatureType Ft_;
  CorrespondenceType Ct_;
  ghicp_params P;
};
}  // namespace ghicp
#endif
