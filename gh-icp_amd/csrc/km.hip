// Kuhn-Munkres on gfx950 with the reference's exact semantics (src/km.cpp:13-126):
// labels lx = row max, ly = 0; per root the slack array is reset to INF2 = 1000; a root is
// grown by repeated DFS (ascending y, eps-tight test `lx+ly-w < eps`) and, on failure,
// relabelled by delta = min slack over unvisited y.
//
// How the emulation is organised (DESIGN.md "KM"):
//  * A FAILED findpath() visits exactly the alternating-reachable set, and every visited x ends up
//    scanning its whole row, so visx, visy and slack[y] (y unvisited) do not depend on the DFS order.
//    Those phases run as a wave-parallel BFS (one wave per frontier row, LDS atomics on slack/visy).
//  * The DFS order only decides WHICH augmenting path the one successful findpath() per root takes.
//    When the BFS meets a free y the phase is re-run as an exact DFS emulation (explicit stack,
//    wave ballots pick the lowest tight unvisited y), which yields the reference's match[] bit for bit.
// One persistent workgroup per problem: the solve is a dependency chain, throughput comes from
// running many pairs' solves on different CUs (streams), not from spreading one solve over the chip.
#include "ctx.h"
#include "devmath.h"

#include <cstdlib>

namespace {

constexpr int KM_THREADS = 1024;
constexpr int KM_WAVES = KM_THREADS / 64;
constexpr double KM_INF2 = 1000.0;  // km.cpp:42

__global__ __launch_bounds__(256) void k_km_rowmax(const int* __restrict__ done, const double* __restrict__ w, int n, double* __restrict__ lx) {
  if (done && *done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  const double* r = w + (size_t)row * n;
  double m = r[0];  // km.cpp:56-62
  for (int j = lane; j < n; j += 64) m = fmax(m, r[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if (lane == 0) lx[row] = m;
}

__device__ inline unsigned long long d2u(double d) { return (unsigned long long)__double_as_longlong(d); }

// bytes of per-problem state kept next to the CU (LDS when it fits, else global scratch):
//   lx, ly, slack (f64) + match, stack x, stack y (i32) + visx/visy bitsets
__host__ __device__ inline size_t km_hot_bytes(int n) { return (size_t)n * 36 + 2 * (size_t)((n + 31) / 32) * 4 + 64; }

template <bool IN_LDS>
__global__ __launch_bounds__(KM_THREADS) void k_km_solve(const int* __restrict__ done, const double* __restrict__ W, int n, double eps,
                                                         const double* __restrict__ lx_init, int* __restrict__ match_out,
                                                         char* __restrict__ gscratch, int* __restrict__ status) {
  if (done && *done) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[16];
  __shared__ int s_nnext, s_free, s_flag;
  const int nw32 = (n + 31) / 32;
  char* base = IN_LDS ? smem : gscratch;
  double* lx = (double*)base;
  double* ly = lx + n;
  double* slack = ly + n;
  int* match = (int*)(slack + n);
  int* stx = match + n;
  int* sty = stx + n;  // y taken at this level (resume scanning at sty+1), -1 = none yet
  unsigned* visx = (unsigned*)(sty + n);
  unsigned* visy = visx + nw32;
  int* fr0 = (int*)(gscratch + (IN_LDS ? 0 : km_hot_bytes(n)));
  int* fr1 = fr0 + n;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n; i += KM_THREADS) { lx[i] = lx_init[i]; ly[i] = 0.0; match[i] = -1; }
  __syncthreads();

  bool bad = false;
  for (int root = 0; root < n && !bad; ++root) {
    for (int i = tid; i < n; i += KM_THREADS) slack[i] = KM_INF2;
    __syncthreads();
    for (int phase = 0; !bad; ++phase) {
      // ---------------- order-independent reachability sweep
      for (int i = tid; i < nw32; i += KM_THREADS) { visx[i] = 0u; visy[i] = 0u; }
      __syncthreads();
      if (tid == 0) { fr0[0] = root; visx[root >> 5] = 1u << (root & 31); s_free = 0; s_nnext = 0; }
      __syncthreads();
      int nf = 1;
      int* cur = fr0;
      int* nxt = fr1;
      while (nf > 0) {
        for (int f = wave; f < nf; f += KM_WAVES) {
          const int x = cur[f];
          const double lxv = lx[x];
          const double* row = W + (size_t)x * n;
          for (int y0 = 0; y0 < n; y0 += 256) {
            double wv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int y = y0 + u * 64 + lane; wv[u] = (y < n) ? row[y] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int y = y0 + u * 64 + lane;
              if (y < n) {
                const double d = lxv + ly[y] - wv[u];  // km.cpp:21
                if (d < eps) {
                  const unsigned bit = 1u << (y & 31);
                  if ((atomicOr(&visy[y >> 5], bit) & bit) == 0u) {
                    const int m = match[y];
                    if (m == -1) s_free = 1;
                    else { atomicOr(&visx[m >> 5], 1u << (m & 31)); nxt[atomicAdd(&s_nnext, 1)] = m; }
                  }
                } else {
                  atomicMin((unsigned long long*)&slack[y], d2u(d));  // km.cpp:33 (d >= eps > 0)
                }
              }
            }
          }
        }
        __syncthreads();
        nf = s_free ? 0 : s_nnext;
        __syncthreads();
        if (tid == 0) s_nnext = 0;
        int* t = cur; cur = nxt; nxt = t;
        __syncthreads();
      }
      if (!s_free) {
        // ---------------- failed phase: relabel (km.cpp:80-98)
        double dl = KM_INF2;
        for (int j = tid; j < n; j += KM_THREADS)
          if (!((visy[j >> 5] >> (j & 31)) & 1u)) dl = fmin(dl, slack[j]);
        dl = gh_block_min(dl, red);
        for (int i = tid; i < n; i += KM_THREADS) {
          if ((visx[i >> 5] >> (i & 31)) & 1u) lx[i] -= dl;
          if ((visy[i >> 5] >> (i & 31)) & 1u) ly[i] += dl;
          else slack[i] -= dl;
        }
        __syncthreads();
        if (phase > 4 * n + 16) {  // cannot happen for finite weights; never spin forever on NaNs
          if (tid == 0) *status = 2;
          bad = true;
        }
        continue;
      }
      // ---------------- successful phase: exact DFS emulation (km.cpp:13-37), wave 0 only
      for (int i = tid; i < nw32; i += KM_THREADS) { visx[i] = 0u; visy[i] = 0u; }
      __syncthreads();
      if (wave == 0) {
        int sp = 0;
        if (lane == 0) { stx[0] = root; sty[0] = -1; }
        int x = root, ystart = 0;
        bool ok = false;
        for (;;) {
          const double lxv = lx[x];
          const double* row = W + (size_t)x * n;
          int ystar = -1;
          for (int y0 = ystart; y0 < n && ystar < 0; y0 += 256) {
            double wv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int y = y0 + u * 64 + lane; wv[u] = (y < n) ? row[y] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int y = y0 + u * 64 + lane;
              bool tight = false;
              if (y < n && !((visy[y >> 5] >> (y & 31)) & 1u)) tight = (lxv + ly[y] - wv[u]) < eps;
              const unsigned long long b = __ballot(tight);
              if (b != 0ull && ystar < 0) ystar = y0 + u * 64 + (int)__ffsll((long long)b) - 1;
            }
          }
          if (ystar >= 0) {
            const int m = match[ystar];
            if (lane == 0) { visy[ystar >> 5] |= 1u << (ystar & 31); sty[sp] = ystar; }
            if (!IN_LDS) __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            if (m == -1) {
              // unwind: match[y] = x on every level (km.cpp:26-29); all y distinct
              for (int f = lane; f <= sp; f += 64) match[sty[f]] = stx[f];
              ok = true;
              break;
            }
            sp++;
            if (lane == 0) { stx[sp] = m; sty[sp] = -1; }
            x = m; ystart = 0;
          } else {
            sp--;
            if (sp < 0) break;
            if (!IN_LDS) __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            x = stx[sp]; ystart = sty[sp] + 1;
          }
          if (!IN_LDS) __threadfence_block();
          __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) s_flag = ok ? 1 : 0;
      }
      __syncthreads();
      if (!s_flag) {
        if (tid == 0) *status = 3;  // reachability said "free y reachable" but the DFS found none
        bad = true;
      }
      break;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += KM_THREADS) match_out[i] = match[i];
}

}  // namespace

int gh_km_solve_dev(ghicp_ctx* ctx, const double* w, int n, double eps, int32_t* match, const int* done_flag) {
  if (n <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  double* lx;
  char* scratch;
  int* status;
  GH_TRY(ctx->reserve(B_KM_LX, (size_t)n + 1, &lx));
  const size_t hot = km_hot_bytes(n);
  GH_TRY(ctx->reserve(B_LOOP_KMSCR, hot + (size_t)n * 8 + 64, &scratch));
  GH_TRY(ctx->reserve(B_KM_MISC, 4, &status));
  GH_HIP(hipMemsetAsync(status, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_km_rowmax, dim3(cdiv(n, 4)), dim3(256), 0, s, done_flag, w, n, lx);
  const size_t lds_limit = 160 * 1024 - 512;
  hipEvent_t kt = ctx->kt_begin(KT_KM_SOLVE);
  if (hot <= lds_limit) {
    {  // per device: set before every launch
      GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
    }
    hipLaunchKernelGGL(k_km_solve<true>, dim3(1), dim3(KM_THREADS), hot, s, done_flag, w, n, eps, lx, match, scratch, status);
  } else {
    hipLaunchKernelGGL(k_km_solve<false>, dim3(1), dim3(KM_THREADS), 0, s, done_flag, w, n, eps, lx, match, scratch, status);
  }
  ctx->kt_end(KT_KM_SOLVE, kt);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

#include "km_prob.h"
int gh_km_solve_sparse_from_dense(ghicp_ctx* ctx, const double* w, int n, double eps, int32_t* match, int* status_dev);

extern "C" int ghicp_km_solve(ghicp_ctx* ctx, const double* w, int64_t n, double eps, int32_t* match) {
  GH_ENTER(ctx);
  GH_ARG(n >= 0 && n < 46000 && (n == 0 || (w != nullptr && match != nullptr)));
  Stager sg(ctx);
  const double* dw;
  int32_t* dm;
  GH_TRY(sg.in(w, (size_t)n * n, &dw));
  GH_TRY(sg.out(match, (size_t)n, &dm));
  const bool v2 = n > 0 && gh_km4_fits((int)n);  // LDS-resident sparse solver (n <= ~3700); beyond: the dense solver below
  if (v2) {
    int* stv;
    GH_TRY(ctx->reserve(B_P_MISC, 16, &stv));
    GH_HIP(hipMemsetAsync(stv, 0, sizeof(int), ctx->stream));
    GH_TRY(gh_km_solve_sparse_from_dense(ctx, dw, (int)n, eps, dm, stv));
  } else {
    GH_TRY(gh_km_solve_dev(ctx, dw, (int)n, eps, dm, nullptr));
  }
  GH_TRY(sg.finish());
  GH_HIP(hipStreamSynchronize(ctx->stream));
  if (n > 0) {
    int st = 0;
    GH_HIP(hipMemcpy(&st, v2 ? ctx->buf[B_P_MISC].p : ctx->buf[B_KM_MISC].p, sizeof(int), hipMemcpyDeviceToHost));
    if ((st & 0xFFFF) != 0) return ctx->fail(GHICP_ERR_INTERNAL, "km_solve: solver status %d (non-finite weights?)", st & 0xFFFF);  // (bits 16+: a count of literal-fallback solves, diagnostics)
  }
  return GHICP_OK;
}
