// Internal: descriptor of one sparse Kuhn-Munkres problem (shared by loop.hip, km_dense_door.hip, km4.hip).  Not part of the ABI.
#pragma once
struct Km2Problem {
  int n, pad_;
  double bg, eps;
  const unsigned* row_ptr;  // n+1
  const int* cols;          // ascending within a row
  const double* vals;       // explicit weights, each > bg
  const double* lx_init;    // row maxima over the full row (km.cpp:56-62)
  int* match_out;           // n: match[y] = x
  int* status;              // 0 = ok
  const int* done;          // optional early-exit flag (device)
  long long* steps;         // optional: counters (profiling)
};
struct ghicp_ctx;
int gh_km4_launch(ghicp_ctx* ctx, const Km2Problem* d_probs, int nprob, int n_max);
bool gh_km4_fits(int n);
size_t gh_km4_lds_bytes(int n);

// A batch of problems of different sizes: problems are grouped into classes of equal LDS occupancy (problems per CU) so that one
// large problem does not lower the occupancy of all the others, and within a class the largest problems start first.
struct Km4Plan {
  int nclass = 0;
  int begin[8] = {0}, count[8] = {0};
  size_t lds[8] = {0};
  int per_cu[8] = {0};      // problems per CU of the class by LDS (capped at four)
  double weight[8] = {0};   // the class's share of the batch's work: sum of the cost hints, or of n^2 without hints
  int* d_order = nullptr;  // device: problem indices, class after class
};
// cost: optional per-problem cost hints (host, nprob floats): within a class the costliest problems are queued first
int gh_km4_plan(ghicp_ctx* ctx, const int* h_n, int nprob, Km4Plan* plan, const float* cost = nullptr);
int gh_km4_launch_plan(ghicp_ctx* ctx, const Km2Problem* d_probs, const Km4Plan& plan);
