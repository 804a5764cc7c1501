// Exercises the reference-named C++ classes (include/*.h) end to end on the GPU: the code below is what
// test/ghicp_main.cpp:86-153 does, minus file I/O -- voxel filter and bounds (CFilter), keypoints, BSC or FPFH features, GHRegistration,
// the final transform of the raw source.  Prints the final 4x4 and a few counters for the pytest wrapper.
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>

#include "binary_feature_extraction.hpp"
#include "common_reg.h"
#include "filter.hpp"
#include "fpfh.hpp"
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

// wall clock of the reference's own call sequence (INTEGRATION.md quotes it): "TIME <stage> <ms>" lines, ignored by the parity checks
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define STAGE(name) do { const double t_ = now_ms(); printf("TIME %s %.3f\n", name, t_ - t_stage); t_stage = t_; } while (0)

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  // KM known-answer vector (src/km.cpp:237-259)
  {
    Graph g; g.n = 3; g.sp = 3; g.tp = 3;
    g.GTable = {{-5, -2, -100}, {-4, -2, -6}, {-100, -1, -7}};
    Km km(g, 0.01, 1000.0);
    km.kmsolve();
    std::vector<int> SP, TP, SPo, TPo;  // Km::output (km.cpp:144-233): every edge differs from -penalty, so TP = 0, 1, 2
    km.output(SP, TP, SPo, TPo);
    printf("KMKAT %d %d %d energy %g\n", SP[0], SP[1], SP[2], km.Calenergy());
  }
  pcl::PointCloud<Point_T>::Ptr Traw = load(argv[1]), Sraw = load(argv[2]);
  {  // context creation + HIP start-up: outside the timed sequence (a process pays it once)
    Bounds warm;
    CFilter<Point_T> f0;
    f0.getCloudBound(*Traw, warm);
  }
  const double t_begin = now_ms();
  double t_stage = t_begin;
  const char corr = argv[3][0];  // N: BSC + NN, K: BSC + KM, R: FPFH + NNR (main:118-127)
  // Downsampling + bbx_magnitude (main:86-93)
  CFilter<Point_T> cfilter;
  pcl::PointCloud<Point_T>::Ptr T(new pcl::PointCloud<Point_T>()), S(new pcl::PointCloud<Point_T>());
  cfilter.voxelfilter(Traw, T, 0.1f);
  cfilter.voxelfilter(Sraw, S, 0.1f);
  Bounds s_cloud_bbx;
  cfilter.getCloudBound(*S, s_cloud_bbx);
  float bbx = s_cloud_bbx.max_x - s_cloud_bbx.min_x + s_cloud_bbx.max_y - s_cloud_bbx.min_y + s_cloud_bbx.max_z - s_cloud_bbx.min_z;
  STAGE("voxelfilter+bounds");
  printf("DS %zu %zu BBX %.9g FIRST %.9g %.9g %.9g\n", T->points.size(), S->points.size(), bbx, S->points[1].x, S->points[1].y, S->points[1].z);
  CKeypointDetect<Point_T> ckpd(0.5f, 0.65f, 20, 1.5f);
  pcl::PointIndicesPtr kT, kS;
  ckpd.keypointDetectionBasedOnCurvature(T, kT);
  ckpd.keypointDetectionBasedOnCurvature(S, kS);
  STAGE("keypoints");
  Eigen::MatrixX3d kpS, kpT;
  kpS.resize((long)kS->indices.size(), 3); kpT.resize((long)kT->indices.size(), 3);
  for (size_t i = 0; i < kS->indices.size(); i++) { const Point_T& p = S->points[kS->indices[i]]; kpS(i, 0) = p.x; kpS(i, 1) = p.y; kpS(i, 2) = p.z; }
  for (size_t i = 0; i < kT->indices.size(); i++) { const Point_T& p = T->points[kT->indices[i]]; kpT(i, 0) = p.x; kpT(i, 1) = p.y; kpT(i, 2) = p.z; }
  Keypoints Kp;
  Kp.setCoordinate(kpS, kpT);
  if (corr == 'R') {  // FPFH feature (main:118-127)
    FPFHfeature<Point_T> fpfh(1.5f);
    fpfhFeaturePtr fpfhT(new fpfhFeature), fpfhS(new fpfhFeature), fpfhT_k(new fpfhFeature), fpfhS_k(new fpfhFeature);
    fpfh.compute_fpfh_feature(T, fpfhT);
    fpfh.compute_fpfh_feature(S, fpfhS);
    fpfh.keyfpfh(fpfhS, fpfhT, kS, kT, fpfhS_k, fpfhT_k);
    Kp.setFPFHfeature(fpfhS_k, fpfhT_k);
    if (!fpfhS_k->points.empty() && !fpfhT_k->points.empty())
      printf("FPFHD %.9g\n", fpfh.compute_fpfh_distance(fpfhS_k->points[0].histogram, fpfhT_k->points[0].histogram));
  } else {
    // default: the reference's READ path (bfe:103-115) -- ./sample_pattern.txt, written by the caller; "rand": the pattern drawn from
    // rand() and written out (bfe:75-101, Q2), whose sequence depends on how often the process called rand() before (the HIP runtime does)
    BSCEncoder<Point_T> bsc(1.5f, 7, argc > 4 && std::strcmp(argv[4], "rand") == 0);
    doubleVectorSBF bscT, bscS;
    bsc.extractBinaryFeatures(T, kT, 0, bscT);
    bsc.extractBinaryFeatures(S, kS, 6, bscS);
    Kp.setBSCfeature(bscS, bscT);
  }
  STAGE("features");
  Energyfunction Ef;
  Ef.init((int)kS->indices.size(), (int)kT->indices.size(), bbx);
  GHRegistration reg(Kp, Ef, corr == 'R' ? FPFH : BSC, corr == 'K' ? KM : (corr == 'R' ? NNR : NN), 1.5f, 1.1f, 0.1f, 6, 0.6f);
  reg.set_max_iterations(80);
  Eigen::Matrix4d Rt;
  reg.ghicp_reg(Rt);
  STAGE("ghicp_reg");
  printf("KP %zu %zu ITER %d\n", kS->indices.size(), kT->indices.size(), reg.iterations);
  printf("RT");
  for (int i = 0; i < 16; i++) printf(" %.17g", Rt(i / 4, i % 4));
  printf("\n");
  {  // pcl::transformPointCloud(*pointCloudS, *pointCloudS_reg, Rt_final.cast<float>()) (main:153) on the RAW source
    std::vector<float> reg3(Sraw->points.size() * 3);
    double Rt16[16];
    for (int i = 0; i < 16; i++) Rt16[i] = Rt(i / 4, i % 4);
    if (ghicp_transform_cloud(detail::ctx(), detail::xyz(*Sraw), (int64_t)Sraw->points.size(), detail::stride<Point_T>(), Rt16, reg3.data()) != GHICP_OK) return 4;
    printf("REG %.9g %.9g %.9g\n", reg3[3 * 11], reg3[3 * 11 + 1], reg3[3 * 11 + 2]);
  }
  STAGE("transformPointCloud");
  printf("TIME main_86_153_total %.3f\n", now_ms() - t_begin);
  {
    int64_t hits = 0, misses = 0, kept = 0;
    ghicp_ctx_stage_stats(detail::ctx(), &hits, &misses, &kept);
    printf("STAGED hits %lld misses %lld bytes_kept %lld\n", (long long)hits, (long long)misses, (long long)kept);
  }
  // fine registration after GH-ICP (CRegistration, common_reg.h): coarse-aligned source -> trimmed point-to-point ICP
  {
    CRegistration<Point_T> creg;
    Eigen::Matrix4f Rf, Ticp, Tinv;
    for (int i = 0; i < 16; i++) Rf(i / 4, i % 4) = (float)Rt(i / 4, i % 4);
    pcl::PointCloud<Point_T>::Ptr S1(new pcl::PointCloud<Point_T>()), S2(new pcl::PointCloud<Point_T>());
    creg.transformcloud(S, S1, Rf);
    printf("OVERLAP %.9g\n", creg.calOverlap(S1, T, 0.3f));
    const bool ok = creg.icp_reg(S1, T, S2, Ticp, 20, false, true, 0.3f, 0.1f);
    creg.invTransform(Ticp, Tinv);
    printf("ICPSTATS done %d overlap %.9g\n", creg.last_stats.done, creg.last_stats.overlap);
    if (!ok) Ticp = Rf;  // a refused registration leaves the caller's matrix alone (common_reg.cpp:66-70): print something defined
    printf("ICP %d %d %d %zu", ok ? 1 : 0, creg.last_stats.iterations, creg.last_stats.reason, S2->points.size());
    for (int i = 0; i < 16; i++) printf(" %.9g", Ticp(i / 4, i % 4));
    printf("\n");
    printf("INV");
    for (int i = 0; i < 16; i++) printf(" %.9g", Tinv(i / 4, i % 4));
    printf("\n");
    printf("S1 %.9g %.9g %.9g\n", S1->points[7].x, S1->points[7].y, S1->points[7].z);
  }
  return 0;
}
