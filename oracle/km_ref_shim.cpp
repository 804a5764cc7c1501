// TEST INFRASTRUCTURE: exposes the reference's own Km (include/km.h + src/km.cpp, compiled from
// where they lie under /root/reference) through a C entry point so tests can pin the oracle's
// restatement of Kuhn-Munkres against the real thing.  Output only into oracle/_ref/.
#define private public  // Km keeps its Graph private; the shim reads gra.match after kmsolve()
#include "km.h"
#undef private
extern "C" int ref_km_solve(const double* w, int n, double eps, double penalty, int* match) {
  ghicp::Graph g;
  g.GTable.assign(n, std::vector<double>(n));
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) g.GTable[i][j] = w[(size_t)i * n + j];
  g.n = n; g.sp = n; g.tp = n;   // same set-up as findcorrespondenceKM (src/ghicp_reg.cpp:416-426)
  g.lx.resize(n); g.ly.resize(n); g.match.resize(n); g.slack.resize(n); g.visx.resize(n); g.visy.resize(n);
  ghicp::Km km(g, eps, penalty);
  km.kmsolve();
  for (int i = 0; i < n; i++) match[i] = km.gra.match[i];
  return 0;
}
