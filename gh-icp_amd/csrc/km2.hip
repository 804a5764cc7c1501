// Kuhn-Munkres, second-generation kernel (gfx950): the reference's exact traversal (src/km.cpp:13-126)
// on a SPARSE + BACKGROUND representation of the weight matrix, all solver state in LDS, one wave per
// problem (wave-synchronous: no barriers on the dependency chain), many problems per launch.
//
// GH-ICP's KM matrix (src/ghicp_reg.cpp:348-365) is -penalty everywhere except where CD < penalty.
// Rows are therefore stored as "background bg = -penalty + sorted explicit entries (col, w > bg)".
// What makes the emulation exact (all by monotonicity of IEEE-754 rounding; ly >= 0 always):
//   E1  for a background entry the reference evaluates d = fl(fl(lx+ly[y]) - bg): same expression here,
//       bg comes from a register instead of memory.
//   E2  d >= fl(lx - bg) for every background entry, so a row whose fl(lx-bg) >= eps has no tight
//       background edge and only its explicit entries are looked at.
//   E3  for an explicit entry w >= bg: fl(t - w) <= fl(t - bg), so "background-tight" implies tight;
//       the first tight unvisited y of a row is min(first background-tight y, first tight explicit y).
//   E4  in a FAILED phase every visited row ends up scanning all its columns, so for an unvisited y
//       slack[y] = min(old, min_x d(x,y)); over background entries the minimum is attained by the visited
//       row with the smallest lx (monotone), hence ONE O(n) pass with lxmin replaces |visx| row sweeps;
//       explicit entries update slack individually.  In a SUCCESSFUL phase slack is dead (reset per root).
//   E5  the DFS itself (explicit stack, lowest tight unvisited y first) decides the augmenting path: it is
//       emulated step by step, one wave-parallel row probe per findpath() call.
#include "ctx.h"
#include "devmath.h"

#include <climits>

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
  long long* steps;         // optional: number of findpath() calls (profiling)
};

size_t gh_km2_lds_bytes(int n) { return (size_t)n * 40 + 4 + 2 * (size_t)((n + 31) / 32) * 4 + 64; }

namespace {

constexpr double KM_INF2 = 1000.0;  // km.cpp:42

__device__ inline bool bit_get(const unsigned* b, int i) { return (b[i >> 5] >> (i & 31)) & 1u; }

// lowest y >= y0 whose bit is clear, or n
__device__ inline int first_clear(const unsigned* __restrict__ bits, int y0, int n, int nw, int lane) {
  const int w0 = y0 >> 5;
  for (int wb = w0; wb < nw; wb += 64) {
    const int w = wb + lane;
    unsigned inv = 0u;
    if (w < nw) {
      inv = ~bits[w];
      if (w == w0) inv &= ~0u << (y0 & 31);
      if (w == nw - 1 && (n & 31)) inv &= (1u << (n & 31)) - 1u;
    }
    const unsigned long long b = __ballot(inv != 0u);
    if (b) {
      const int l = (int)__ffsll((long long)b) - 1;
      const unsigned iv = (unsigned)__shfl((int)inv, l, 64);
      return (wb + l) * 32 + (__ffs((int)iv) - 1);
    }
  }
  return n;
}

__global__ __launch_bounds__(64) void k_km2(const Km2Problem* __restrict__ probs) {
  const Km2Problem P = probs[blockIdx.x];
  if (P.n <= 0 || (P.done && *P.done)) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = P.n, nw = (n + 31) / 32, lane = threadIdx.x;
  double* lx = (double*)smem;
  double* ly = lx + n;
  double* slack = ly + n;
  int* match = (int*)(slack + n);
  int* stx = match + n;
  int* sty = stx + n;
  unsigned* rptr = (unsigned*)(sty + n);
  unsigned* visx = rptr + n + 1;
  unsigned* visy = visx + nw;
  const double bg = P.bg, eps = P.eps;

  for (int i = lane; i < n; i += 64) { lx[i] = P.lx_init[i]; ly[i] = 0.0; match[i] = -1; rptr[i] = P.row_ptr[i]; }
  if (lane == 0) rptr[n] = P.row_ptr[n];
  __syncthreads();

  long long nsteps = 0;
  int bad = 0;
  for (int root = 0; root < n && !bad; ++root) {
    for (int i = lane; i < n; i += 64) slack[i] = KM_INF2;
    for (int phase = 0;; ++phase) {
      for (int i = lane; i < nw; i += 64) { visx[i] = 0u; visy[i] = 0u; }
      __syncthreads();
      if (lane == 0) { stx[0] = root; sty[0] = -1; visx[root >> 5] = 1u << (root & 31); }
      __syncthreads();
      int sp = 0, x = root, ystart = 0;
      double lxmin = lx[root];
      bool ok = false;
      for (;;) {  // one iteration == one findpath() activation or resumption (km.cpp:13-37)
        nsteps++;
        const double lxv = lx[x];
        int best = INT_MAX;
        // ---- explicit entries of row x (E3); non-tight ones feed slack (km.cpp:33)
        const unsigned cb = rptr[x], ce = rptr[x + 1];
        for (unsigned c0 = cb; c0 < ce; c0 += 64) {
          const unsigned c = c0 + lane;
          int col = INT_MAX;
          bool tight = false;
          if (c < ce) {
            col = P.cols[c];
            const double d = (lxv + ly[col]) - P.vals[c];
            if (d < eps) tight = col >= ystart && !bit_get(visy, col);
            else slack[col] = fmin(slack[col], d);
          }
          const unsigned long long b = __ballot(tight);
          if (b) { best = __shfl(col, (int)__ffsll((long long)b) - 1, 64); break; }
        }
        // ---- background entries (E1, E2)
        if ((lxv - bg) < eps) {
          int y0 = ystart;
          for (;;) {
            y0 = first_clear(visy, y0, n, nw, lane);
            if (y0 >= n || y0 >= best) break;
            const int y = y0 + lane;
            bool t = false;
            if (y < n && !bit_get(visy, y)) t = ((lxv + ly[y]) - bg) < eps;
            const unsigned long long b = __ballot(t);
            if (b) { best = min(best, y0 + (int)__ffsll((long long)b) - 1); break; }
            y0 += 64;
          }
        }
        if (best != INT_MAX) {
          const int ystar = best;
          const int m = match[ystar];
          if (lane == 0) { visy[ystar >> 5] |= 1u << (ystar & 31); sty[sp] = ystar; }
          __builtin_amdgcn_wave_barrier();
          if (m == -1) {  // augment: match[y] = x on every level of the recursion (km.cpp:26-29)
            for (int f = lane; f <= sp; f += 64) match[sty[f]] = stx[f];
            ok = true;
            break;
          }
          sp++;
          if (lane == 0) { stx[sp] = m; sty[sp] = -1; visx[m >> 5] |= 1u << (m & 31); }
          lxmin = fmin(lxmin, lx[m]);
          x = m; ystart = 0;
        } else {
          sp--;
          if (sp < 0) break;
          x = stx[sp]; ystart = sty[sp] + 1;
        }
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      if (ok) break;
      // ---- failed phase: deferred background slack (E4) + relabel (km.cpp:80-98)
      double dl = KM_INF2;
      for (int y = lane; y < n; y += 64)
        if (!bit_get(visy, y)) {
          const double s = fmin(slack[y], (lxmin + ly[y]) - bg);
          slack[y] = s;
          dl = fmin(dl, s);
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dl = fmin(dl, __shfl_xor(dl, o, 64));
      for (int i = lane; i < n; i += 64) {
        if (bit_get(visx, i)) lx[i] -= dl;
        if (bit_get(visy, i)) ly[i] += dl;
        else slack[i] -= dl;
      }
      __syncthreads();
      if (phase > 4 * n + 16) { bad = 2; break; }  // only reachable with non-finite weights
    }
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) P.match_out[i] = match[i];
  if (lane == 0) {
    if (bad && P.status) *P.status = bad;
    if (P.steps) *P.steps = nsteps;
  }
}

// ---- dense matrix -> background + CSR (for the public ghicp_km_solve entry point)
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

// launches one k_km2 block per problem; descriptors already on device
int gh_km2_launch(ghicp_ctx* ctx, const Km2Problem* d_probs, int nprob, int n_max) {
  const size_t lds = gh_km2_lds_bytes(n_max);
  static size_t attr_done = 0;
  if (lds > attr_done) {
    const size_t want = 160 * 1024;
    GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
    attr_done = want;
  }
  hipEvent_t kt = ctx->kt_begin(KT_KM_SOLVE);
  hipLaunchKernelGGL(k_km2, dim3(nprob), dim3(64), lds, ctx->stream, d_probs);
  ctx->kt_end(KT_KM_SOLVE, kt);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

bool gh_km2_fits(int n) { return gh_km2_lds_bytes(n) <= 160 * 1024 - 256; }

// dense front door: background = the matrix minimum, everything above it explicit
int gh_km2_solve_dense(ghicp_ctx* ctx, const double* w, int n, double eps, int32_t* match, int* status_dev) {
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
  GH_HIP(hipMemcpyAsync(dp, &hp, sizeof(hp), hipMemcpyHostToDevice, s));
  return gh_km2_launch(ctx, dp, 1, n);
}
