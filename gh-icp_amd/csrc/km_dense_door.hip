// Dense front door of the sparse Kuhn-Munkres solver (gfx950): the public ghicp_km_solve takes the reference's n x n f64 matrix
// (Km::kmsolve, src/km.cpp:40-126); the solver (km4_dev.h) works on "background + CSR of the explicit entries".  GH-ICP's own
// matrices (src/ghicp_reg.cpp:348-365) are -penalty everywhere except where CD < penalty, and the loop builds their CSR directly
// (loop.hip); for an arbitrary dense matrix the background is the matrix MINIMUM and everything above it is explicit, which is exact
// for the same reasons (E1-E3 in km4_dev.h's header: a background entry is evaluated as fl(fl(lx+ly) - bg) from a register).
// (Round 1's one-wave DFS emulation k_km2 / k_km3 lived here; k_km4 replaced it in round 2 and covers n <= 3700 since round 3, larger
// graphs go to the dense solver of km.hip.)
#include "ctx.h"
#include "devmath.h"

#include <climits>
#include <cstdlib>

#include "km_prob.h"

namespace {

__global__ __launch_bounds__(256) void k_dense_min(const double* __restrict__ w, size_t total, unsigned long long* __restrict__ out) {
  // order-preserving key of a double (ascending)
  double m = INFINITY;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) m = fmin(m, w[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) {
    unsigned long long b = (unsigned long long)__double_as_longlong(m);
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    atomicMin(out, b);
  }
}

__global__ __launch_bounds__(256) void k_dense_rows(const double* __restrict__ w, int n, const unsigned long long* __restrict__ minkey,
                                                    unsigned* __restrict__ cnt, double* __restrict__ lx, const unsigned* __restrict__ row_ptr,
                                                    int* __restrict__ cols, double* __restrict__ vals, double* __restrict__ bg_out, int fill) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  unsigned long long k = *minkey;
  k = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  const double bg = __longlong_as_double((long long)k);
  if (row == 0 && lane == 0) *bg_out = bg;
  const double* r = w + (size_t)row * n;
  unsigned c = 0;
  double mx = r[0];
  const unsigned base = fill ? row_ptr[row] : 0u;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    double v = bg;
    if (j < n) { v = r[j]; mx = fmax(mx, v); }
    const bool e = (j < n) && (v != bg);
    const unsigned long long b = __ballot(e);
    if (fill && e) {
      const unsigned off = c + __popcll(b & ((1ull << lane) - 1ull));
      cols[base + off] = j;
      vals[base + off] = v;
    }
    c += __popcll(b);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if (lane == 0 && !fill) { cnt[row] = c; lx[row] = mx; }
}

}  // namespace

// exclusive scan of n row counts -> row_ptr[0..n] (single block; n <= ~10^5)
__global__ __launch_bounds__(1024) void k_gh_scan_rows(const unsigned* __restrict__ cnt, int n, unsigned* __restrict__ row_ptr) {
  __shared__ int sc[17];
  int carry = 0;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n ? (int)cnt[i] : 0;
    int tot;
    const int ex = gh_block_excl_scan(v, sc, &tot);
    if (i < n) row_ptr[i] = (unsigned)(carry + ex);
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) row_ptr[n] = (unsigned)carry;
}

// dense front door: background = the matrix minimum, everything above it explicit.  n must satisfy gh_km4_fits(n).
int gh_km_solve_sparse_from_dense(ghicp_ctx* ctx, const double* w, int n, double eps, int32_t* match, int* status_dev) {
  hipStream_t s = ctx->stream;
  unsigned long long* mk;
  unsigned *cnt, *rptr;
  double *lx, *vals, *bgd;
  int* cols;
  Km2Problem* dp;
  GH_TRY(ctx->reserve(B_KM_MISC, 64, (int**)&mk));  // [0..1] min key, [2] status, [4..] bg
  GH_TRY(ctx->reserve(B_KM_LX, (size_t)n + 1, &lx));
  char* scr;
  GH_TRY(ctx->reserve(B_LOOP_KMSCR, (size_t)(n + 2) * 8 + 256 + sizeof(Km2Problem), &scr));
  cnt = (unsigned*)scr;
  rptr = cnt + n + 1;
  dp = (Km2Problem*)(scr + (size_t)(n + 2) * 8 + 64);
  GH_TRY(ctx->reserve(B_LOOP_SP, (size_t)n * n + 1, &cols));
  GH_TRY(ctx->reserve(B_LOOP_KMW, (size_t)n * n + 1, &vals));
  bgd = (double*)((char*)mk + 32);
  GH_HIP(hipMemsetAsync(mk, 0xff, 8, s));
  hipLaunchKernelGGL(k_dense_min, dim3(min(cdiv((long long)n * n, 256), 1024)), dim3(256), 0, s, w, (size_t)n * n, mk);
  hipLaunchKernelGGL(k_dense_rows, dim3(cdiv(n, 4)), dim3(256), 0, s, w, n, mk, cnt, lx, (const unsigned*)nullptr, cols, vals, bgd, 0);
  hipLaunchKernelGGL(k_gh_scan_rows, dim3(1), dim3(1024), 0, s, cnt, n, rptr);
  hipLaunchKernelGGL(k_dense_rows, dim3(cdiv(n, 4)), dim3(256), 0, s, w, n, mk, cnt, lx, rptr, cols, vals, bgd, 1);
  double bg_h = 0;
  GH_HIP(hipMemcpyAsync(&bg_h, bgd, sizeof(double), hipMemcpyDeviceToHost, s));
  GH_HIP(hipStreamSynchronize(s));
  Km2Problem hp;
  memset(&hp, 0, sizeof(hp));
  hp.n = n; hp.bg = bg_h; hp.eps = eps; hp.row_ptr = rptr; hp.cols = cols; hp.vals = vals; hp.lx_init = lx; hp.match_out = match;
  hp.status = status_dev;
  long long* dstats = nullptr;
  if (ctx->km_stats) {  // GHICP_KM_STATS=1 (read once per context): stage counters of the solve
    GH_TRY(ctx->reserve(B_P_PATTERN, 40, &dstats));
    GH_HIP(hipMemsetAsync(dstats, 0, 32 * sizeof(long long), s));
    hp.steps = dstats;
  }
  GH_HIP(hipMemcpyAsync(dp, &hp, sizeof(hp), hipMemcpyHostToDevice, s));
  GH_TRY(gh_km4_launch(ctx, dp, 1, n));
  if (dstats) {
    long long h[32];
    GH_HIP(hipMemcpyAsync(h, dstats, sizeof(h), hipMemcpyDeviceToHost, s));
    GH_HIP(hipStreamSynchronize(s));
    fprintf(stderr, "[km4 stats] n=%d activations=%lld phases=%lld failed=%lld pull_rounds=%lld dfs_iterations=%lld flood_rows(failed)=%lld rebuilt_rows=%lld | cycles: flood=%lld failed=%lld pull=%lld dfs=%lld total=%lld | hazard=%lld | flood: levels=%lld flagged_rows=%lld cyc_rows=%lld sweeps=%lld cyc_sweeps=%lld\n",
            n, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], h[16], h[17]);
    fprintf(stderr, "[km4 dfs] iterations by kind (count / cycles): flagged row %lld / %lld, listed-only row %lld / %lld, background-tight row with a list %lld / %lld, march %lld / %lld, pop %lld / %lld\n",
            h[18], h[23], h[19], h[24], h[20], h[25], h[21], h[26], h[22], h[27]);
  }
  return GHICP_OK;
}
