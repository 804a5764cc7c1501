// Exercises the reference-named C++ classes (include/*.h) end to end on the GPU: the code below is what
// test/ghicp_main.cpp:95-151 does, minus file I/O.  Prints the final 4x4 and a few counters for the pytest wrapper.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "binary_feature_extraction.hpp"
#include "ghicp_reg.h"
#include "keypoint_detect.hpp"
#include "km.h"

using namespace ghicp;
typedef pcl::PointXYZI Point_T;

static pcl::PointCloud<Point_T>::Ptr load(const char* path) {
  pcl::PointCloud<Point_T>::Ptr c(new pcl::PointCloud<Point_T>());
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  int n = 0;
  if (fread(&n, 4, 1, f) != 1) exit(2);
  c->points.resize(n);
  for (int i = 0; i < n; i++) {
    float p[3];
    if (fread(p, 4, 3, f) != 3) exit(2);
    std::memset(&c->points[i], 0, sizeof(Point_T));
    c->points[i].x = p[0]; c->points[i].y = p[1]; c->points[i].z = p[2];
  }
  fclose(f);
  return c;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  // KM known-answer vector (src/km.cpp:237-259)
  {
    Graph g; g.n = 3; g.sp = 3; g.tp = 3;
    g.GTable = {{-5, -2, -100}, {-4, -2, -6}, {-100, -1, -7}};
    Km km(g, 0.01, 1000.0);
    km.kmsolve();
    printf("KMKAT %d %d %d energy %g\n", km.match()[0], km.match()[1], km.match()[2], km.Calenergy());
  }
  pcl::PointCloud<Point_T>::Ptr T = load(argv[1]), S = load(argv[2]);
  const char corr = argv[3][0];
  float bbx = 0;
  if (ghicp_bbx_magnitude(detail::ctx(), detail::xyz(*S), (int64_t)S->size(), detail::stride<Point_T>(), &bbx) != GHICP_OK) return 3;
  CKeypointDetect<Point_T> ckpd(0.5f, 0.65f, 20, 1.5f);
  pcl::PointIndicesPtr kT, kS;
  ckpd.keypointDetectionBasedOnCurvature(T, kT);
  ckpd.keypointDetectionBasedOnCurvature(S, kS);
  Eigen::MatrixX3d kpS, kpT;
  kpS.resize((long)kS->indices.size(), 3); kpT.resize((long)kT->indices.size(), 3);
  for (size_t i = 0; i < kS->indices.size(); i++) { const Point_T& p = S->points[kS->indices[i]]; kpS(i, 0) = p.x; kpS(i, 1) = p.y; kpS(i, 2) = p.z; }
  for (size_t i = 0; i < kT->indices.size(); i++) { const Point_T& p = T->points[kT->indices[i]]; kpT(i, 0) = p.x; kpT(i, 1) = p.y; kpT(i, 2) = p.z; }
  Keypoints Kp;
  Kp.setCoordinate(kpS, kpT);
  BSCEncoder<Point_T> bsc(1.5f, 7, true);  // glibc rand() sample pattern (Q2)
  doubleVectorSBF bscT, bscS;
  bsc.extractBinaryFeatures(T, kT, 0, bscT);
  bsc.extractBinaryFeatures(S, kS, 6, bscS);
  Kp.setBSCfeature(bscS, bscT);
  Energyfunction Ef;
  Ef.init((int)kS->indices.size(), (int)kT->indices.size(), bbx);
  GHRegistration reg(Kp, Ef, BSC, corr == 'K' ? KM : NN, 1.5f, 1.1f, 0.1f, 6, 0.6f);
  reg.set_max_iterations(80);
  Eigen::Matrix4d Rt;
  reg.ghicp_reg(Rt);
  printf("KP %zu %zu ITER %d\n", kS->indices.size(), kT->indices.size(), reg.iterations);
  printf("RT");
  for (int i = 0; i < 16; i++) printf(" %.17g", Rt.m[i]);
  printf("\n");
  return 0;
}
