// Host-only check of the control-point solvers of include/common_reg.h (CRegistration::CSTRAN_4DOF / CSTRAN_7DOF / LLS_4DOF /
// SVD_6DOF, reference src/common_reg.cpp:425-888): reads "n cp" then n rows "ax ay az bx by bz", prints the fitted parameters.
#include <cstdio>

#include "common_reg.h"

using namespace ghicp;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "r");
  if (!f) return 2;
  int n = 0, cp = 0;
  double theta0 = 0;
  if (fscanf(f, "%d %d %lf", &n, &cp, &theta0) != 3) return 2;
  std::vector<std::vector<double>> A(n, std::vector<double>(3)), B(n, std::vector<double>(3));
  for (int i = 0; i < n; i++)
    if (fscanf(f, "%lf %lf %lf %lf %lf %lf", &A[i][0], &A[i][1], &A[i][2], &B[i][0], &B[i][1], &B[i][2]) != 6) return 2;
  fclose(f);
  CRegistration<pcl::PointXYZ> reg;
  std::vector<double> p4, p7;
  const bool ok4 = reg.CSTRAN_4DOF(A, B, p4, cp);
  printf("C4 %d %.17g %.17g %.17g %.17g %.17g RMSE %.17g\n", ok4, p4[0], p4[1], p4[2], p4[3], p4[4], reg.last_check_rmse);
  const bool ok7 = reg.CSTRAN_7DOF(A, B, p7, cp);
  printf("C7 %d", ok7);
  for (int i = 0; i < 7; i++) printf(" %.17g", p7[i]);
  printf(" RMSE %.17g\n", reg.last_check_rmse);
  Eigen::Matrix4d T;
  const bool okl = reg.LLS_4DOF(A, B, T, cp, theta0);
  printf("LLS %d", okl);
  for (int i = 0; i < 16; i++) printf(" %.17g", T(i / 4, i % 4));
  printf(" RMSE %.17g\n", reg.last_check_rmse);
  const bool oks = reg.SVD_6DOF(A, B, T, cp);
  printf("SVD %d", oks);
  for (int i = 0; i < 16; i++) printf(" %.17g", T(i / 4, i % 4));
  printf(" RMSE %.17g\n", reg.last_check_rmse);
  std::vector<double> tmp;
  printf("FEW %d %d %d %d\n", reg.CSTRAN_4DOF(A, B, tmp, 2), reg.CSTRAN_7DOF(A, B, tmp, 3), reg.LLS_4DOF(A, B, T, 1, 0.0), reg.SVD_6DOF(A, B, T, 1));
  return 0;
}
