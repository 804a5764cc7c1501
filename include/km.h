// Drop-in for include/km.h + src/km.cpp: same Graph / Km surface, Km::kmsolve runs on the GPU (ghicp_km_solve)
// with the reference's exact traversal order.  No "Corres.txt" is written (km.cpp:147 side effect removed).
#ifndef GHICP_DROPIN_KM_H_
#define GHICP_DROPIN_KM_H_
#include <vector>

#include "utility.h"

namespace ghicp {
struct Graph {  // km.h:15-30
  std::vector<std::vector<double>> GTable;
  int n = 0, sp = 0, tp = 0;
  std::vector<int> match;
  std::vector<double> lx, ly, slack;
  std::vector<bool> visx, visy;
  double energy = 0;
  std::vector<int> min_match;
  int min_n = 0;
};

class Km {
 public:
  Km(Graph graph, double eps0, double penalty0) : penalty(penalty0), precision(0), recall(0), gra(graph), eps(eps0) {}  // km.h:38-43
  void kmsolve() {  // km.cpp:40-126
    const int n = gra.n;
    std::vector<double> w((size_t)n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) w[(size_t)i * n + j] = gra.GTable[i][j];
    gra.match.assign(n, -1);
    detail::check(ghicp_km_solve(detail::ctx(), w.data(), n, eps, gra.match.data()));
  }
  double Calenergy() {  // km.cpp:128-141
    gra.energy = 0;
    for (int i = 0; i < gra.n; i++) if (gra.GTable[gra.match[i]][i] != -10000) gra.energy -= gra.GTable[gra.match[i]][i];
    return gra.energy;
  }
  int output(std::vector<int>& SP, std::vector<int>& TP, std::vector<int>& SPout, std::vector<int>& TPout) {  // km.cpp:144-233
    int cor = 0, exact = 0;
    for (int i = 0; i < gra.n; i++) {
      if (gra.match[i] == i) exact++;
      if (gra.GTable[gra.match[i]][i] != -penalty) { SP.push_back(gra.match[i]); TP.push_back(i); cor++; }
      else if (gra.sp >= gra.tp) { SPout.push_back(gra.match[i]); if (i < gra.tp) TPout.push_back(i); }
      else { TPout.push_back(i); if (gra.match[i] < gra.sp) SPout.push_back(gra.match[i]); }
    }
    precision = 1.0 * exact / cor;
    recall = 1.0 * exact / gra.n;
    return cor;
  }
  bool findpath(int) { return false; }  // internal to the GPU solver
  double penalty, precision, recall;

 private:
  Graph gra;
  double eps;
};
}  // namespace ghicp
#endif
