// Host-only check of include/dataio.hpp (drop-in for the reference's DataIo<PointT>): reads the files named on the command
// line, prints what it saw, and writes the cloud back in every supported format for the pytest side to parse independently.
#include <cstdio>

#include "dataio.hpp"

using namespace ghicp;
typedef pcl::PointXYZI Point_T;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string outdir = argv[1];
  DataIo<Point_T> io;
  for (int a = 2; a < argc; a++) {
    pcl::PointCloud<Point_T>::Ptr c(new pcl::PointCloud<Point_T>());
    const bool ok = io.readCloudFile(argv[a], c);
    double sx = 0, sy = 0, sz = 0, si = 0;
    for (size_t i = 0; i < c->points.size(); i++) { sx += c->points[i].x; sy += c->points[i].y; sz += c->points[i].z; si += c->points[i].intensity; }
    printf("READ %s %d %zu %.9g %.9g %.9g %.9g\n", argv[a], ok ? 1 : 0, c->points.size(), sx, sy, sz, si);
    if (ok && a == 2) {
      if (!io.writeCloudFile(outdir + "/out.pcd", c) || !io.writeCloudFile(outdir + "/out.ply", c) || !io.writeCloudFile(outdir + "/out.txt", c)) return 3;
      if (!io.writeTxtFile(outdir + "/out_sub.txt", c, 3)) return 3;
      pcl::PointIndicesPtr kp(new pcl::PointIndices());
      kp->indices = {2, 0, 5};
      if (!io.outputKeypoints(outdir + "/kp.txt", kp, c)) return 3;
      Eigen::MatrixX3d S, T;
      io.savecoordinates(c, c, kp, kp, S, T);
      printf("COORD %ld %.9g %.9g %.9g\n", S.rows(), S(0, 0), S(1, 1), T(2, 2));
      printf("UNDEF %d\n", io.readCloudFile(outdir + "/nothing.xyz", c) ? 1 : 0);
    }
  }
  return 0;
}
