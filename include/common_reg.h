// Drop-in for include/common_reg.h (+ src/common_reg.cpp): ghicp::CRegistration<PointT>, the fine-registration
// wrappers the reference builds on PCL's ICP family.  Same class / method names and argument meaning; the work runs on
// the GPU through the C ABI (ghicp_icp, ghicp_cal_overlap, ghicp_transform_cloud_f32, ghicp_inv_transform).
//   icp_reg        common_reg.cpp:45-107    point-to-point, closed-form SVD per iteration
//   ptplicp_reg    common_reg.cpp:122-199   point-to-plane LLS; normals by k-NN PCA (covariance_K <= 20)
//   calOverlap     common_reg.cpp:294-317
//   transformcloud common_reg.cpp:325-349
//   invTransform   common_reg.cpp:357-370   (R^T with the negated translation -- "Not Mathimatically" an inverse)
//   CSTRAN_4DOF / CSTRAN_7DOF / LLS_4DOF / SVD_6DOF   common_reg.cpp:425-888: closed-form fits from a handful of control
//                  points; host arithmetic (normal equations in f64; SVD_6DOF = the float Umeyama of the path through
//                  ghicp_rigid_svd_host), no kernel launch
// Differences a caller can observe: ptplicp_reg and SVD_6DOF return true (the reference falls off the end of a bool
// function); (A^T A)^-1 A^T b is solved by elimination with partial pivoting instead of an explicit inverse (agrees to
// rounding); gicp_reg / Coarsereg_FPFHSAC are not part of the hot path and are not provided.
#ifndef GHICP_DROPIN_COMMON_REG_H_
#define GHICP_DROPIN_COMMON_REG_H_
#include <cmath>
#include <ctime>
#include <iostream>
#include <vector>

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

  // ---- closed-form solvers from control points (common_reg.cpp:425-888).  coordinatesA/B: rows of (x, y, z); the first
  // cp_number rows are fitted, the rest only checked (RMSE printed, kept in last_check_rmse; -1 when nothing to check).
  // X Y yaw scale: B = s R(yaw) A + t in the plane.  transpara = {tx, ty, s, sin, cos}   (common_reg.cpp:425-516)
  bool CSTRAN_4DOF(const std::vector<std::vector<double>>& coordinatesA, const std::vector<std::vector<double>>& coordinatesB,
                   std::vector<double>& transpara, int cp_number) {
    transpara.resize(5);
    if (cp_number < 3) { std::cout << "Error ! Not enough control point number ..." << std::endl; return false; }
    std::vector<double> A((size_t)cp_number * 2 * 4), b((size_t)cp_number * 2), x;
    for (int j = 0; j < cp_number; j++) {
      const double ax = coordinatesA[j][0], ay = coordinatesA[j][1];
      double* r0 = &A[(size_t)(j * 2) * 4];
      double* r1 = r0 + 4;
      r0[0] = 1; r0[1] = 0; r0[2] = ax; r0[3] = -ay;
      r1[0] = 0; r1[1] = 1; r1[2] = ay; r1[3] = ax;
      b[j * 2] = coordinatesB[j][0];
      b[j * 2 + 1] = coordinatesB[j][1];
    }
    if (!least_squares(A, b, cp_number * 2, 4, x)) return false;
    const double s = std::sqrt(x[2] * x[2] + x[3] * x[3]);
    transpara[0] = x[0]; transpara[1] = x[1]; transpara[2] = s; transpara[3] = x[3] / s; transpara[4] = x[2] / s;
    std::cout << "Estimated Transformation From A to B" << std::endl << "tx: " << x[0] << " m" << std::endl << "ty: " << x[1] << " m" << std::endl
              << "scale: " << s << std::endl;
    check(coordinatesA, coordinatesB, cp_number, [&](const std::vector<double>& p, double* o) {
      o[0] = transpara[2] * transpara[4] * p[0] - transpara[2] * transpara[3] * p[1] + transpara[0];
      o[1] = transpara[2] * transpara[3] * p[0] + transpara[2] * transpara[4] * p[1] + transpara[1];
      o[2] = 0;
    }, 2);
    return true;
  }

  // X Y Z roll pitch yaw scale, small-angle model.  transpara = {tx, ty, tz, rx, ry, rz, s}   (common_reg.cpp:518-616)
  bool CSTRAN_7DOF(const std::vector<std::vector<double>>& coordinatesA, const std::vector<std::vector<double>>& coordinatesB,
                   std::vector<double>& transpara, int cp_number) {
    transpara.resize(7);
    if (cp_number < 4) { std::cout << "Error ! Not enough control point number ..." << std::endl; return false; }
    std::vector<double> A((size_t)cp_number * 3 * 7, 0.0), b((size_t)cp_number * 3), x;
    for (int j = 0; j < cp_number; j++) {
      const double ax = coordinatesA[j][0], ay = coordinatesA[j][1], az = coordinatesA[j][2];
      double* r0 = &A[(size_t)(j * 3) * 7];
      double *r1 = r0 + 7, *r2 = r0 + 14;
      r0[0] = 1; r0[4] = -az; r0[5] = ay; r0[6] = ax;
      r1[1] = 1; r1[3] = az; r1[5] = -ax; r1[6] = ay;
      r2[2] = 1; r2[3] = -ay; r2[4] = ax; r2[6] = az;
      for (int d = 0; d < 3; d++) b[j * 3 + d] = coordinatesB[j][d];
    }
    if (!least_squares(A, b, cp_number * 3, 7, x)) return false;
    for (int i = 0; i < 7; i++) transpara[i] = x[i];
    std::cout << "Estimated Transformation From A to B" << std::endl << "tx: " << x[0] << " m" << std::endl << "ty: " << x[1] << " m" << std::endl
              << "tz: " << x[2] << " m" << std::endl << "rx: " << x[3] << std::endl << "ry: " << x[4] << std::endl << "rz: " << x[5] << std::endl
              << "scale: " << x[6] << std::endl;
    check(coordinatesA, coordinatesB, cp_number, [&](const std::vector<double>& p, double* o) {
      o[0] = x[0] + x[6] * p[0] + x[5] * p[1] - x[4] * p[2];
      o[1] = x[1] + x[6] * p[1] - x[5] * p[0] + x[3] * p[2];
      o[2] = x[2] + x[6] * p[2] + x[4] * p[0] - x[3] * p[1];
    }, 3);
    return true;
  }

  // X Y Z yaw by Gauss-Newton from the initial yaw theta0_degree, until |dtheta| <= 1e-9   (common_reg.cpp:619-772)
  bool LLS_4DOF(const std::vector<std::vector<double>>& coordinatesA, const std::vector<std::vector<double>>& coordinatesB,
                Eigen::Matrix4d& TransMatrixA2B, int cp_number, double theta0_degree) {
    if (cp_number < 2) { std::cout << "Error ! Not enough control point number ..." << std::endl; return false; }
    double theta0 = theta0_degree / 180 * M_PI, dtheta = 9999;
    const double eps = 1e-9;
    std::vector<double> A((size_t)cp_number * 3 * 4), b((size_t)cp_number * 3), x(4, 0.0);
    int iter_num = 0;
    while (std::fabs(dtheta) > eps) {
      for (int j = 0; j < cp_number; j++) {
        const double ax = coordinatesA[j][0], ay = coordinatesA[j][1];
        double* r0 = &A[(size_t)(j * 3) * 4];
        double *r1 = r0 + 4, *r2 = r0 + 8;
        r0[0] = -ax * std::sin(theta0) - ay * std::cos(theta0); r0[1] = 1; r0[2] = 0; r0[3] = 0;
        r1[0] = ax * std::cos(theta0) - ay * std::sin(theta0); r1[1] = 0; r1[2] = 1; r1[3] = 0;
        r2[0] = 0; r2[1] = 0; r2[2] = 0; r2[3] = 1;
        b[j * 3] = coordinatesB[j][0] - ax * std::cos(theta0) + ay * std::sin(theta0);
        b[j * 3 + 1] = coordinatesB[j][1] - ax * std::sin(theta0) - ay * std::cos(theta0);
        b[j * 3 + 2] = coordinatesB[j][2] - coordinatesA[j][2];
      }
      if (!least_squares(A, b, cp_number * 3, 4, x)) return false;
      dtheta = x[0];
      theta0 += dtheta;
      if (++iter_num > 1000) return false;  // the reference would spin forever on a degenerate configuration
    }
    const double theta = theta0, tx = x[1], ty = x[2], tz = x[3];
    std::cout << "Calculated by Linear Least Square" << std::endl << "Converged in " << iter_num << " iterations ..." << std::endl;
    const double T[16] = {std::cos(theta), -std::sin(theta), 0, tx, std::sin(theta), std::cos(theta), 0, ty, 0, 0, 1, tz, 0, 0, 0, 1};
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) TransMatrixA2B(r, c) = T[r * 4 + c];
    check(coordinatesA, coordinatesB, cp_number, [&](const std::vector<double>& p, double* o) {
      o[0] = std::cos(theta) * p[0] - std::sin(theta) * p[1] + tx;
      o[1] = std::sin(theta) * p[0] + std::cos(theta) * p[1] + ty;
      o[2] = p[2] + tz;
    }, 3);
    return true;
  }

  // X Y Z roll pitch yaw: pcl TransformationEstimationSVD on the control points (float Umeyama)   (common_reg.cpp:774-888)
  bool SVD_6DOF(const std::vector<std::vector<double>>& coordinatesA, const std::vector<std::vector<double>>& coordinatesB,
                Eigen::Matrix4d& TransMatrixA2B, int cp_number) {
    if (cp_number < 2) { std::cout << "Error ! Not enough control point number ..." << std::endl; return false; }
    std::vector<double> a((size_t)cp_number * 3), b((size_t)cp_number * 3);
    for (int i = 0; i < cp_number; i++)
      for (int d = 0; d < 3; d++) { a[(size_t)i * 3 + d] = coordinatesA[i][d]; b[(size_t)i * 3 + d] = coordinatesB[i][d]; }
    double T[16];
    if (ghicp_rigid_svd_host(a.data(), b.data(), cp_number, T) != GHICP_OK) return false;
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) TransMatrixA2B(r, c) = T[r * 4 + c];
    std::cout << "Calculated by SVD" << std::endl;
    check(coordinatesA, coordinatesB, cp_number, [&](const std::vector<double>& p, double* o) {
      for (int r = 0; r < 3; r++) o[r] = T[r * 4] * p[0] + T[r * 4 + 1] * p[1] + T[r * 4 + 2] * p[2] + T[r * 4 + 3];
    }, 3);
    return true;
  }

  double last_check_rmse = -1.0;  // RMSE over the check points of the last control-point solver call

  // statistics of the last icp_reg / ptplicp_reg call (the reference only logs them)
  ghicp_icp_stats last_stats = {};

 private:
  // x = (A^T A)^-1 A^T b, A row-major rows x n (n <= 7): normal equations in f64, elimination with partial pivoting
  static bool least_squares(const std::vector<double>& A, const std::vector<double>& b, int rows, int n, std::vector<double>& x) {
    double M[7][8];
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) {
        double s = 0;
        for (int r = 0; r < rows; r++) s += A[(size_t)r * n + i] * A[(size_t)r * n + j];
        M[i][j] = s;
      }
      double s = 0;
      for (int r = 0; r < rows; r++) s += A[(size_t)r * n + i] * b[r];
      M[i][n] = s;
    }
    for (int c = 0; c < n; c++) {
      int piv = c;
      for (int r = c + 1; r < n; r++) if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
      if (M[piv][c] == 0.0) return false;
      if (piv != c) for (int q = 0; q <= n; q++) { const double t = M[c][q]; M[c][q] = M[piv][q]; M[piv][q] = t; }
      for (int r = c + 1; r < n; r++) {
        const double f = M[r][c] / M[c][c];
        for (int q = c; q <= n; q++) M[r][q] -= f * M[c][q];
      }
    }
    x.assign((size_t)n, 0.0);
    for (int r = n - 1; r >= 0; r--) {
      double s = M[r][n];
      for (int q = r + 1; q < n; q++) s -= M[r][q] * x[q];
      x[r] = s / M[r][r];
    }
    return true;
  }
  // RMSE of the fitted map over the points after the first cp_number ("Checking", e.g. common_reg.cpp:490-513)
  template <typename F>
  void check(const std::vector<std::vector<double>>& A, const std::vector<std::vector<double>>& B, int cp_number, F&& map, int dims) {
    const int total = (int)(A.size() >= B.size() ? B.size() : A.size());
    last_check_rmse = -1.0;
    if (total <= cp_number) { std::cout << "Not enough points for check ..." << std::endl; return; }
    const int nchk = total - cp_number;
    double sum = 0;
    for (int j = 0; j < nchk; j++) {
      double o[3];
      map(A[(size_t)(j + cp_number)], o);
      for (int d = 0; d < dims; d++) sum += (o[d] - B[(size_t)(j + cp_number)][d]) * (o[d] - B[(size_t)(j + cp_number)][d]);
    }
    last_check_rmse = std::sqrt(sum / nchk);
    std::cout << "Calculated from " << nchk << " points, the RMSE is " << last_check_rmse << std::endl;
  }
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
