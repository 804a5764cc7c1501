// GH-ICP iteration loop on gfx950: replaces GHRegistration::ghicp_reg and its private helpers
// (reference src/ghicp_reg.cpp:24-112, 114-139, 216-341, 343-460, 548-578, 605-927).
//
// Independent scan pairs are the parallel axis of this problem: the per-pair Kuhn-Munkres solve is one wave's dependency
// chain for most of its time, so throughput comes from many pairs in flight.  Every stage is a device function over a per-pair
// descriptor (LoopProb), used in two ways:
//   * Kuhn-Munkres batches whose graphs fit the LDS-resident solver: the PERSISTENT pair loop k_pair_loop -- one 256-thread workgroup is
//     one solve slot, pops a pair from its class queue and runs the pair's whole ghicp_reg loop, iteration after iteration, before it
//     pops the next;
//   * NN / NNR batches and graphs beyond LDS: one launch per stage advances all pairs of the batch by one stage (thin kernel wrappers
//     around the same device functions); a pair that has converged makes its blocks exit at once, the host polls the `done` flags.
// Stages of one iteration:
//   dev_cd_rowmin    fused calED + calCD_* + row arg-min (+ column arg-min sweep for NNR) + sum / sum^2 over
//                    K_S x K_T; no f64 ED/CD matrix is ever materialised                        (S5, HBM-bound)
//   dev_penalty      CDmean / CDstd -> penalty (calCD_* tails)                                    (scalar)
//   [KM] dev_km_csr x2 + dev_km_scan_desc + k4_solve_block (sparse exact Kuhn-Munkres, km4_dev.h)  (S5 KM)
//   dev_solve        accept correspondences, RMSE/FDM/FDstd, float-Umeyama rigid solve, apply to all source
//                    keypoints, RMSE-after, Euler convergence test, adjustweight, Rt product       (S6)
#include "ctx.h"
#include "devmath.h"

#include <cmath>

#include "km_prob.h"
#include "km4_dev.h"
int gh_km_solve_dev(ghicp_ctx* ctx, const double* w, int n, double eps, int32_t* match, const int* done_flag);

namespace {

struct LoopState {
  int it, done, cor, converged_flag;
  double RMS, FDM, FDstd, IoU, para1, para2, penalty, CDmean, CDstd, energy;
  double Rt_till[16];
  double rmse_after;
  unsigned long long t_begin, t_end;  // persistent pair loop: when a slot took the pair and when it let go (s_memrealtime, 100 MHz)
  // diagnostics of the persistent loop (kernel timing on; round 6, the stragglers of DESIGN.md §6): the pair's longest Kuhn-Munkres solve, the
  // iteration it belongs to, and where the slot ran (HW_ID: compute unit / shader array / engine, XCC_ID: the die)
  unsigned long long t_solve_max;
  int it_solve_max;
  unsigned hw_id;
};

struct LoopConst {
  int ks, kt, n, feature, corr, max_iter, min_cor, nchunk_a, nchunk_b, chunk_a, chunk_b, nparts;
  float scale, est_iou, adjust_ratio, adjust_step;
  double converge_t, converge_r, penalty_initial, km_eps;
};

// one registration job of the batch; every pointer is device memory
struct LoopProb {
  LoopConst C;
  LoopState* st;
  double* kpS;            // the pair's own copy (the loop transforms it in place)
  const double* kpS_src;  // where it is copied from when the batch starts
  const double* kpT;
  const void* FD;   // [ks][kt]
  const void* FDt;  // [kt][ks]
  int fdt_given;    // the job came with its transposed matrix (k_pairs_transpose skips it)
  const double* wfd;
  double *pminA, *pminB, *psum;
  int *pidxA, *pidxB, *SP, *TP, *SVs, *TVs;
  ghicp_iter* trace;
  int* matchlist;
  int ml_row0;  // matchlist row of iteration `it` is it - ml_row0 (a resumed loop hands over one row per call)
  // KM
  unsigned *km_cnt, *km_rptr;
  int *km_cols, *kmmatch, *km_status;
  double *km_vals, *km_lx, *kmw;
  Km2Problem* km_desc;
};

constexpr int ROWS = 256;       // threads per block in the sweep = rows handled per block
constexpr int CHUNK_MAX = 512;  // columns staged in LDS per block

template <int FT>
__device__ inline double combined_distance(double ed, const void* fd, size_t idx, double wed, double wfd, double inv_k) {
  if (FT == GHICP_FEATURE_BSC) return wed * ed + wfd * (double)reinterpret_cast<const uint16_t*>(fd)[idx];     // ghicp_reg.cpp:259
  if (FT == GHICP_FEATURE_FPFH) return 1.0 * ed / pow((double)reinterpret_cast<const float*>(fd)[idx], inv_k);  // ghicp_reg.cpp:308
  return ed;                                                                                                     // ghicp_reg.cpp:224
}

template <int FT>
__device__ inline double cd_pivot(const LoopProb& P, double wed, double wfd, double inv_k) {
  const double dx = P.kpS[0] - P.kpT[0], dy = P.kpS[1] - P.kpT[1], dz = P.kpS[2] - P.kpT[2];
  return combined_distance<FT>((double)P.C.scale * sqrt(dx * dx + dy * dy + dz * dz), P.FD, 0, wed, wfd, inv_k);
}

// One sweep: thread = "row" a (keypoint of set A), loop over a chunk of set B staged in LDS.
// The feature matrix is read as [b][a] so that lanes (consecutive a) touch consecutive addresses.
// Row arg-min semantics = ghicp_reg.cpp:715-724 / 622-650: start (9e20, 0), strict '<', ascending index.
// (bx, by) = the block coordinates of the stand-alone kernel; sB: CHUNK_MAX * 3 doubles, red: 16 doubles of LDS.  The persistent
// pair loop calls the same body for every (bx, by) in turn, so both paths produce the same partial sums in the same order.
template <int FT, bool COLS>
__device__ inline void dev_cd_rowmin(const LoopProb& P, const int bx, const int by, double* sB, double* red) {
  const int ka = COLS ? P.C.kt : P.C.ks, kb = COLS ? P.C.ks : P.C.kt;
  const int chunk = COLS ? P.C.chunk_a : P.C.chunk_b, nchunk = COLS ? P.C.nchunk_a : P.C.nchunk_b;
  if (by >= nchunk || bx * ROWS >= ka) return;
  const double* A = COLS ? P.kpT : P.kpS;
  const double* B = COLS ? P.kpS : P.kpT;
  const void* F = COLS ? P.FD : P.FDt;
  const int it = P.st->it;
  const int jb = by * chunk;
  const int je = min(kb, jb + chunk);
  for (int t = threadIdx.x; t < (je - jb) * 3; t += ROWS) sB[t] = B[(size_t)jb * 3 + t];
  __syncthreads();
  const int a = bx * ROWS + threadIdx.x;
  const bool live = a < ka;
  double ax = 0, ay = 0, az = 0;
  if (live) { ax = A[(size_t)a * 3]; ay = A[(size_t)a * 3 + 1]; az = A[(size_t)a * 3 + 2]; }
  double wfd = 0, wed = 1;
  if (FT == GHICP_FEATURE_BSC) { wfd = P.wfd[it]; wed = 1.0 - wfd; }
  const double inv_k = 1.0 / (double)(it + 1);
  const double dscale = (double)P.C.scale;
  double best = 9e20, s = 0, s2 = 0;
  int bidx = 0;
  // CDmean / CDstd are accumulated around a pivot (the CD of keypoint pair (0,0), the same value in every block and in
  // k_penalty): sum^2 / n - mean^2 would cancel when the spread of CD is small against its mean (ghicp_reg.cpp:266-273 is two-pass)
  double piv = 0;
  if (!COLS) piv = cd_pivot<FT>(P, wed, wfd, inv_k);
  if (live) {
    for (int j = jb; j < je; j++) {
      const double dx = ax - sB[(j - jb) * 3], dy = ay - sB[(j - jb) * 3 + 1], dz = az - sB[(j - jb) * 3 + 2];
      const double ed = dscale * sqrt(dx * dx + dy * dy + dz * dz);  // ghicp_reg.cpp:122
      const double cd = combined_distance<FT>(ed, F, (size_t)j * ka + a, wed, wfd, inv_k);
      if (cd < best) { best = cd; bidx = j; }
      if (!COLS) { const double c0 = cd - piv; s += c0; s2 += c0 * c0; }
    }
    (COLS ? P.pminB : P.pminA)[(size_t)by * ka + a] = best;
    (COLS ? P.pidxB : P.pidxA)[(size_t)by * ka + a] = bidx;
  }
  if (!COLS) {
    const double bs = gh_block_sum(s, red);
    const double bs2 = gh_block_sum(s2, red);
    if (threadIdx.x == 0) {
      const size_t b = (size_t)by * cdiv_dev(ka, ROWS) + bx;
      P.psum[b * 2] = bs;
      P.psum[b * 2 + 1] = bs2;
    }
  }
}

template <int FT, bool COLS>
__global__ __launch_bounds__(ROWS) void k_cd_rowmin(const LoopProb* __restrict__ probs) {
  const LoopProb& P = probs[blockIdx.z];
  if (P.st->done) return;
  __shared__ double sB[CHUNK_MAX * 3];
  __shared__ double red[16];
  dev_cd_rowmin<FT, COLS>(P, (int)blockIdx.x, (int)blockIdx.y, sB, red);
}

// calCD_* tails: CDmean, CDstd, penalty (ghicp_reg.cpp:228-239, 264-287, 317-335)
__device__ inline void dev_penalty(const LoopProb& P, double* red) {
  LoopState* st = P.st;
  const LoopConst& C = P.C;
  double s = 0, s2 = 0;
  for (int i = threadIdx.x; i < C.nparts; i += blockDim.x) { s += P.psum[i * 2]; s2 += P.psum[i * 2 + 1]; }
  s = gh_block_sum(s, red);
  s2 = gh_block_sum(s2, red);
  if (threadIdx.x == 0) {
    const int it = st->it;
    const double cnt = (double)C.ks * (double)C.kt;
    double piv, wfd0 = 0, wed0 = 1;
    if (C.feature == GHICP_FEATURE_BSC) { wfd0 = P.wfd[it]; wed0 = 1.0 - wfd0; }
    const double inv_k0 = 1.0 / (double)(it + 1);
    if (C.feature == GHICP_FEATURE_BSC) piv = cd_pivot<GHICP_FEATURE_BSC>(P, wed0, wfd0, inv_k0);
    else if (C.feature == GHICP_FEATURE_FPFH) piv = cd_pivot<GHICP_FEATURE_FPFH>(P, wed0, wfd0, inv_k0);
    else piv = cd_pivot<GHICP_FEATURE_NONE>(P, wed0, wfd0, inv_k0);
    const double dm = s / (double)C.kt / (double)C.ks;  // mean of (CD - pivot)
    const double mean = piv + dm;
    double var = s2 / cnt - dm * dm;
    if (var < 0) var = 0;
    const double sd = sqrt(var);
    double pen;
    if (C.feature == GHICP_FEATURE_NONE) {
      pen = fmax(mean, 1.0);  // Q6: line 239 overrides 230-237
      st->CDstd = 0;
    } else if (C.feature == GHICP_FEATURE_BSC) {
      const double wfd = P.wfd[it], wed = 1.0 - wfd;
      if (it > 1) pen = st->RMS * st->para1 * (double)C.scale * wed + (st->FDM + st->para2 * st->FDstd) * wfd;
      else pen = mean - C.penalty_initial * sd;
      pen = fmax(pen, 5.0);
      st->CDstd = sd;
    } else {
      if (it > 1) pen = st->RMS * st->para1 * (double)C.scale * st->para2;
      else pen = mean / C.penalty_initial;
      st->CDstd = 0;
    }
    st->CDmean = mean;
    st->penalty = pen;
  }
}

__global__ __launch_bounds__(256) void k_penalty(const LoopProb* __restrict__ probs) {
  const LoopProb& P = probs[blockIdx.x];
  if (P.st->done) return;
  __shared__ double red[16];
  dev_penalty(P, red);
}

// Dense KM weights for the large-n fallback (ghicp_reg.cpp:348-365).
template <int FT>
__global__ __launch_bounds__(256) void k_km_weights(const LoopProb* __restrict__ probs) {
  const LoopProb& P = probs[blockIdx.z];
  if (P.st->done || P.kmw == nullptr) return;
  const LoopConst& C = P.C;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= C.n || i >= C.n) return;
  const double pen = P.st->penalty;
  double out = -pen;
  if (i < C.ks && j < C.kt) {
    const int it = P.st->it;
    const double dx = P.kpS[(size_t)i * 3] - P.kpT[(size_t)j * 3], dy = P.kpS[(size_t)i * 3 + 1] - P.kpT[(size_t)j * 3 + 1],
                 dz = P.kpS[(size_t)i * 3 + 2] - P.kpT[(size_t)j * 3 + 2];
    const double ed = (double)C.scale * sqrt(dx * dx + dy * dy + dz * dz);
    double wfd = 0, wed = 1;
    if (FT == GHICP_FEATURE_BSC) { wfd = P.wfd[it]; wed = 1.0 - wfd; }
    const double cd = combined_distance<FT>(ed, P.FD, (size_t)i * C.kt + j, wed, wfd, 1.0 / (double)(it + 1));
    if (cd < pen) out = -cd;
  }
  P.kmw[(size_t)i * C.n + j] = out;
}

// Sparse KM input (km_prob.h): per row the explicit entries (j, -CD) with CD < penalty (ghicp_reg.cpp:358-365);
// every other entry of the n x n graph is the background -penalty.  One wave per row, two passes (count, fill).
template <int FT, int FILL>
__device__ inline void dev_km_csr(const LoopProb& P, const int bx) {
  const LoopConst& C = P.C;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = bx * 4 + wave;
  if (i >= C.n) return;
  const double pen = P.st->penalty;
  if (i >= C.ks) {  // padding rows: all background
    if (!FILL && lane == 0) { P.km_cnt[i] = 0u; P.km_lx[i] = -pen; }
    return;
  }
  const int it = P.st->it;
  double wfd = 0, wed = 1;
  if (FT == GHICP_FEATURE_BSC) { wfd = P.wfd[it]; wed = 1.0 - wfd; }
  const double inv_k = 1.0 / (double)(it + 1);
  const double sx = P.kpS[(size_t)i * 3], sy = P.kpS[(size_t)i * 3 + 1], sz = P.kpS[(size_t)i * 3 + 2];
  const unsigned base = FILL ? P.km_rptr[i] : 0u;
  unsigned c = 0;
  double mx = -pen;  // km.cpp:56-62 row maximum; every explicit entry is > -penalty
  for (int j0 = 0; j0 < C.kt; j0 += 64) {
    const int j = j0 + lane;
    bool e = false;
    double wv = 0;
    if (j < C.kt) {
      const double dx = sx - P.kpT[(size_t)j * 3], dy = sy - P.kpT[(size_t)j * 3 + 1], dz = sz - P.kpT[(size_t)j * 3 + 2];
      const double ed = (double)C.scale * sqrt(dx * dx + dy * dy + dz * dz);
      const double cd = combined_distance<FT>(ed, P.FD, (size_t)i * C.kt + j, wed, wfd, inv_k);
      e = cd < pen;
      wv = -cd;
    }
    const unsigned long long b = __ballot(e);
    if (FILL && e) {
      const unsigned off = c + __popcll(b & ((1ull << lane) - 1ull));
      P.km_cols[base + off] = j;
      P.km_vals[base + off] = wv;
    }
    if (!FILL && e) mx = fmax(mx, wv);
    c += __popcll(b);
  }
  if (!FILL) {
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) { P.km_cnt[i] = c; P.km_lx[i] = mx; }
  }
}

template <int FT, int FILL>
__global__ __launch_bounds__(256) void k_km_csr(const LoopProb* __restrict__ probs) {
  const LoopProb& P = probs[blockIdx.y];
  if (P.st->done || P.km_rptr == nullptr) return;
  dev_km_csr<FT, FILL>(P, (int)blockIdx.x);
}

// exclusive scan of the row counts (one block per pair) + the km2 problem descriptor
__device__ inline void dev_km_scan_desc(const LoopProb& P, int* sc) {
  const int n = P.C.n;
  int carry = 0;
  for (int base = 0; base < n; base += (int)blockDim.x) {
    const int i = base + threadIdx.x;
    const int v = i < n ? (int)P.km_cnt[i] : 0;
    int tot;
    const int ex = gh_block_excl_scan(v, sc, &tot);
    if (i < n) P.km_rptr[i] = (unsigned)(carry + ex);
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    P.km_rptr[n] = (unsigned)carry;
    Km2Problem p;
    p.n = n; p.pad_ = 0; p.bg = -P.st->penalty; p.eps = P.C.km_eps; p.row_ptr = P.km_rptr; p.cols = P.km_cols; p.vals = P.km_vals;
    p.lx_init = P.km_lx; p.match_out = P.kmmatch; p.status = P.km_status; p.done = &P.st->done; p.steps = nullptr;
    *P.km_desc = p;
  }
}

__global__ __launch_bounds__(1024) void k_km_scan_desc(const LoopProb* __restrict__ probs) {
  const LoopProb& P = probs[blockIdx.x];
  if (P.st->done || P.km_rptr == nullptr) return;
  __shared__ int sc[17];
  dev_km_scan_desc(P, sc);
}

// Everything after the sweep, one 1024-thread workgroup per pair.
// red: 16 doubles, ired: 17 ints, sh: 32 doubles of LDS
template <int FT>
__device__ inline void dev_solve(const LoopProb& P, double* red, int* ired, double* sh) {
  LoopState* st = P.st;
  const LoopConst& C = P.C;
  double* kpS = P.kpS;
  const double* kpT = P.kpT;
  const void* FD = P.FD;
  int* SP = P.SP;
  int* TP = P.TP;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int it = st->it;
  const double penalty = st->penalty;
  int* matchlist = P.matchlist;
  if (matchlist)
    for (int i = tid; i < C.ks; i += nt) matchlist[(size_t)(it - P.ml_row0) * C.ks + i] = -1;

  // ---- correspondences, in the reference's emission order
  int cor = 0;
  if (C.corr == GHICP_CORR_KM) {
    // Km::output (km.cpp:157-171): ascending y, kept iff w[match[y]][y] != -penalty (exact compare)
    double e = 0;
    for (int base = 0; base < C.n; base += nt) {
      const int y = base + tid;
      int flag = 0, x = -1;
      if (y < C.n) x = P.kmmatch[y];
      if (y < C.n && x >= 0) {  // x < 0: the solver gave up (non-finite weights); no correspondence, status reported by the host
        double g = -penalty;
        if (P.km_rptr) {  // sparse graph: (x,y) carries a weight != -penalty iff it is an explicit entry
          unsigned lo = P.km_rptr[x], hi = P.km_rptr[x + 1];
          const unsigned end = hi;
          while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (P.km_cols[mid] < y) lo = mid + 1; else hi = mid; }
          if (lo < end && P.km_cols[lo] == y) g = P.km_vals[lo];
        } else {
          g = P.kmw[(size_t)x * C.n + y];
        }
        flag = (g != -penalty) ? 1 : 0;
        if (g != -10000.0) e -= g;  // Calenergy (km.cpp:128-141): INF = 10000 never matches
      }
      int tot;
      const int pos = gh_block_excl_scan(flag, ired, &tot);
      if (flag) { SP[cor + pos] = x; TP[cor + pos] = y; }
      cor += tot;
      __syncthreads();
    }
    e = gh_block_sum(e, red);
    if (tid == 0) st->energy = e;
  } else {
    // row arg-min over chunks (ascending chunk == ascending column)
    for (int i = tid; i < C.ks; i += nt) {
      double best = 9e20; int bi = 0;
      for (int c = 0; c < C.nchunk_b; c++) {
        const double v = P.pminA[(size_t)c * C.ks + i];
        if (v < best) { best = v; bi = P.pidxA[(size_t)c * C.ks + i]; }
      }
      P.SVs[i] = bi;
      P.TVs[C.kt + i] = (best < penalty) ? 1 : 0;  // NN acceptance flag (ghicp_reg.cpp:725)
    }
    if (C.corr == GHICP_CORR_NNR) {
      for (int j = tid; j < C.kt; j += nt) {
        double best = 9e20; int bi = 0;
        for (int c = 0; c < C.nchunk_a; c++) {
          const double v = P.pminB[(size_t)c * C.kt + j];
          if (v < best) { best = v; bi = P.pidxB[(size_t)c * C.kt + j]; }
        }
        P.TVs[j] = bi;
      }
    }
    __syncthreads();
    for (int base = 0; base < C.ks; base += nt) {
      const int i = base + tid;
      int flag = 0, sv = 0;
      if (i < C.ks) {
        sv = P.SVs[i];
        if (C.corr == GHICP_CORR_NN) flag = P.TVs[C.kt + i];
        else flag = (C.kt > 0 && P.TVs[sv] == i) ? 1 : 0;  // Q7: reciprocal test only (ghicp_reg.cpp:654)
      }
      int tot;
      const int pos = gh_block_excl_scan(flag, ired, &tot);
      if (flag) { SP[cor + pos] = i; TP[cor + pos] = sv; }
      cor += tot;
      __syncthreads();
    }
  }
  __syncthreads();
  if (matchlist)
    for (int c = tid; c < cor; c += nt) matchlist[(size_t)(it - P.ml_row0) * C.ks + SP[c]] = TP[c];

  // ---- RMSE, FDM, FDstd (ghicp_reg.cpp:548-578)
  double rm = 0, fm = 0;
  for (int c = tid; c < cor; c += nt) {
    const int i = SP[c], j = TP[c];
    const double dx = kpS[(size_t)i * 3] - kpT[(size_t)j * 3], dy = kpS[(size_t)i * 3 + 1] - kpT[(size_t)j * 3 + 1],
                 dz = kpS[(size_t)i * 3 + 2] - kpT[(size_t)j * 3 + 2];
    rm += dx * dx + dy * dy + dz * dz;
    if (FT == GHICP_FEATURE_BSC) fm += (double)reinterpret_cast<const uint16_t*>(FD)[(size_t)i * C.kt + j];
    if (FT == GHICP_FEATURE_FPFH) fm += (double)reinterpret_cast<const float*>(FD)[(size_t)i * C.kt + j];
  }
  rm = gh_block_sum(rm, red);
  fm = gh_block_sum(fm, red);
  const double FDM = fm / (double)cor;
  double fc = 0;
  if (FT != GHICP_FEATURE_NONE)
    for (int c = tid; c < cor; c += nt) {
      const int i = SP[c], j = TP[c];
      double f = (FT == GHICP_FEATURE_BSC) ? (double)reinterpret_cast<const uint16_t*>(FD)[(size_t)i * C.kt + j]
                                           : (double)reinterpret_cast<const float*>(FD)[(size_t)i * C.kt + j];
      f -= FDM;
      fc += f * f;
    }
  fc = gh_block_sum(fc, red);
  const double FDstd = sqrt(fc / (double)cor);
  const double RMSE = sqrt(rm / (double)cor);

  // ---- float Umeyama (ghicp_reg.cpp:839-866): inputs cast to f32, means/cross-covariance in f64, matrix rounded once (N2)
  double m[6] = {0, 0, 0, 0, 0, 0};
  for (int c = tid; c < cor; c += nt) {
    const int i = SP[c], j = TP[c];
    for (int d = 0; d < 3; d++) { m[d] += (double)(float)kpS[(size_t)i * 3 + d]; m[3 + d] += (double)(float)kpT[(size_t)j * 3 + d]; }
  }
  for (int d = 0; d < 6; d++) m[d] = gh_block_sum(m[d], red);
  float msf[3], mtf[3];
  for (int d = 0; d < 3; d++) { msf[d] = (float)(m[d] / (double)cor); mtf[d] = (float)(m[3 + d] / (double)cor); }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = tid; c < cor; c += nt) {
    const int i = SP[c], j = TP[c];
    double a[3], b[3];
    for (int d = 0; d < 3; d++) {
      a[d] = (double)(float)kpT[(size_t)j * 3 + d] - (double)mtf[d];
      b[d] = (double)(float)kpS[(size_t)i * 3 + d] - (double)msf[d];
    }
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) H[r * 3 + q] += a[r] * b[q];
  }
  for (int d = 0; d < 9; d++) H[d] = gh_block_sum(H[d], red);
  if (tid == 0) {
    double A[9], R[9];
    for (int d = 0; d < 9; d++) A[d] = H[d] / (double)cor;
    gh_quant_grid(A, 9);  // N2: umeyama's sigma is a Matrix3f
    gh_kabsch(A, R);
    float Rf[9], tf[3];
    for (int d = 0; d < 9; d++) Rf[d] = (float)R[d];
    for (int r = 0; r < 3; r++)
      tf[r] = (float)((double)mtf[r] -
                      (((double)Rf[r * 3] * (double)msf[0] + (double)Rf[r * 3 + 1] * (double)msf[1]) + (double)Rf[r * 3 + 2] * (double)msf[2]));
    for (int r = 0; r < 3; r++) {
      for (int q = 0; q < 3; q++) sh[r * 4 + q] = (double)Rf[r * 3 + q];
      sh[r * 4 + 3] = (double)tf[r];
    }
  }
  __syncthreads();
  double Rt[12];
  for (int d = 0; d < 12; d++) Rt[d] = sh[d];

  // ---- RMSE after (on the correspondences, before kpS is overwritten) and update of ALL source keypoints
  double ra = 0;
  for (int c = tid; c < cor; c += nt) {
    const int i = SP[c], j = TP[c];
    const double x = kpS[(size_t)i * 3], y = kpS[(size_t)i * 3 + 1], z = kpS[(size_t)i * 3 + 2];
    const double nx = ((Rt[0] * x + Rt[1] * y) + Rt[2] * z) + Rt[3];
    const double ny = ((Rt[4] * x + Rt[5] * y) + Rt[6] * z) + Rt[7];
    const double nz = ((Rt[8] * x + Rt[9] * y) + Rt[10] * z) + Rt[11];
    const double dx = nx - kpT[(size_t)j * 3], dy = ny - kpT[(size_t)j * 3 + 1], dz = nz - kpT[(size_t)j * 3 + 2];
    ra += dx * dx + dy * dy + dz * dz;
  }
  ra = gh_block_sum(ra, red);
  __syncthreads();
  for (int i = tid; i < C.ks; i += nt) {
    const double x = kpS[(size_t)i * 3], y = kpS[(size_t)i * 3 + 1], z = kpS[(size_t)i * 3 + 2];
    kpS[(size_t)i * 3] = ((Rt[0] * x + Rt[1] * y) + Rt[2] * z) + Rt[3];
    kpS[(size_t)i * 3 + 1] = ((Rt[4] * x + Rt[5] * y) + Rt[6] * z) + Rt[7];
    kpS[(size_t)i * 3 + 2] = ((Rt[8] * x + Rt[9] * y) + Rt[10] * z) + Rt[11];
  }

  if (tid == 0) {
    const double RMSEafter = sqrt(ra / (double)cor);
    bool conv = false;
    if (cor < C.min_cor) conv = true;  // ghicp_reg.cpp:796
    const double IoU = 1.0 * (double)cor / (double)(C.ks + C.kt - cor);
    const double dx = Rt[3], dy = Rt[7], dz = Rt[11];
    double ax = atan2(Rt[9], Rt[10]);
    double ay = atan2(-Rt[8], sqrt(Rt[9] * Rt[9] + Rt[10] * Rt[10]));
    double az = atan2(Rt[1], Rt[0]);
    const double pi = 3.1415926;
    ax = ax / pi * 180; ay = ay / pi * 180; az = az / pi * 180;
    if (fabs(dx) < C.converge_t && fabs(dy) < C.converge_t && fabs(dz) < C.converge_t && fabs(ax) < C.converge_r &&
        fabs(ay) < C.converge_r && fabs(az) < C.converge_r)
      conv = true;
    double p1 = st->para1, p2 = st->para2;  // adjustweight ghicp_reg.cpp:771-789
    if ((double)C.est_iou / IoU > (double)C.adjust_ratio) { p1 += (double)C.adjust_step; p2 += (double)C.adjust_step; }
    else if (IoU / (double)C.est_iou > (double)C.adjust_ratio) { p1 -= (double)C.adjust_step; p2 -= (double)C.adjust_step; }
    double nt16[16];
    const double Rt16[16] = {Rt[0], Rt[1], Rt[2], Rt[3], Rt[4], Rt[5], Rt[6], Rt[7], Rt[8], Rt[9], Rt[10], Rt[11], 0, 0, 0, 1};
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += Rt16[r * 4 + k] * st->Rt_till[k * 4 + c];
        nt16[r * 4 + c] = s;
      }
    for (int d = 0; d < 16; d++) st->Rt_till[d] = nt16[d];
    ghicp_iter rec;
    rec.cor = cor; rec.converged = conv ? 1 : 0;
    rec.penalty = penalty; rec.cdmean = st->CDmean; rec.cdstd = st->CDstd; rec.rmse = RMSE; rec.rmse_after = RMSEafter;
    rec.fdm = FDM; rec.fdstd = FDstd; rec.iou = IoU; rec.para1 = p1; rec.para2 = p2;
    rec.energy = (C.corr == GHICP_CORR_KM) ? st->energy : 0.0;
    for (int d = 0; d < 16; d++) rec.Rt[d] = Rt16[d];
    P.trace[it] = rec;
    st->RMS = RMSE; st->FDM = FDM; st->FDstd = FDstd; st->IoU = IoU; st->para1 = p1; st->para2 = p2; st->cor = cor;
    st->rmse_after = RMSEafter;
    st->it = it + 1;
    if (conv || it + 1 >= C.max_iter) { st->done = 1; st->converged_flag = conv ? 1 : 0; }
  }
}

template <int FT>
__global__ __launch_bounds__(1024) void k_solve(const LoopProb* __restrict__ probs) {
  const LoopProb& P = probs[blockIdx.x];
  if (P.st->done) return;
  __shared__ double red[16];
  __shared__ int ired[17];
  __shared__ double sh[32];
  dev_solve<FT>(P, red, ired, sh);
}

// Hand-over of a batch in two launches instead of two per pair (5376 pairs a step: the per-pair copies, memsets and transposes were a
// launch-bound tail of every step, profiles/r03_kernel_stats_bench_default.txt): every pair's source keypoints into its own buffer ...
__global__ __launch_bounds__(256) void k_pairs_copy_kps(const LoopProb* __restrict__ probs, int pair0) {
  const LoopProb& P = probs[pair0 + blockIdx.y];
  const int n = P.C.ks * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) P.kpS[i] = P.kpS_src[i];
}
// ... and every pair's feature-distance matrix transposed once so that the row sweep reads it coalesced (tiles beyond a pair's extent exit)
template <typename T> __global__ void k_pairs_transpose(const LoopProb* __restrict__ probs, int pair0) {
  const LoopProb& P = probs[pair0 + blockIdx.z];
  const int rows = P.C.ks, cols = P.C.kt;
  if (rows <= 0 || cols <= 0 || (int)blockIdx.x * 32 >= cols || (int)blockIdx.y * 32 >= rows || P.fdt_given) return;
  const T* __restrict__ in = reinterpret_cast<const T*>(P.FD);
  T* __restrict__ out = const_cast<T*>(reinterpret_cast<const T*>(P.FDt));
  __shared__ T tile[32][33];
  const int x = blockIdx.x * 32 + threadIdx.x, y0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y)
    if (x < cols && y0 + r < rows) tile[r][threadIdx.x] = in[(size_t)(y0 + r) * cols + x];
  __syncthreads();
  const int ox = blockIdx.y * 32 + threadIdx.x, oy0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y)
    if (ox < rows && oy0 + r < cols) out[(size_t)(oy0 + r) * rows + ox] = tile[threadIdx.x][r];
}


__global__ void k_collect_done(const LoopProb* __restrict__ probs, int n, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out[i * 2] = probs[i].st->it; out[i * 2 + 1] = probs[i].st->done; }
}

// ---- Persistent pair loop (Kuhn-Munkres configurations).  One 256-thread workgroup = one SOLVE SLOT: it pops a pair from the class
// queue and runs that pair's whole GH-ICP loop (ghicp_reg.cpp:49-103: calED + calCD_* sweep, penalty, graph build, Kuhn-Munkres
// solve, transformestimation, adjustweight, iterate until converged), then pops the next pair.  No kernel boundary, no host poll
// and no other pair stands between two iterations of a pair, so a slot is never idle while its queue holds work: converged pairs
// free their slot at once and the next pair is admitted at once (continuous batching at pair granularity; per pair the order of
// ghicp_reg.cpp:49-103 is kept).  The stages are the SAME device functions the stand-alone kernels run, called for every block
// coordinate in turn.  (The sweep's column chunks are sized per path -- one workgroup sweeps a pair here, many workgroups a batch in the
// per-stage path -- so the f64 sums CDmean / CDstd may differ in the last bits between the two paths: N6 of DESIGN.md §2; everything
// that is compared bit for bit -- matches, solver, rigid solve -- is the same code on the same values.)  Stage scratch (13 KB) overlays
// the solver's LDS.
constexpr int PL_SCRATCH = (CHUNK_MAX * 3 + 16 + 32) * 8 + 20 * 4;

// The stages as out-of-line calls: the persistent kernel's register budget is then the LARGEST stage's, not what the register
// allocator makes of all of them inlined into one loop (256 VGPRs + scratch, one workgroup per CU, when everything is inlined).
template <int FT>
__device__ __noinline__ void pl_sweep(const LoopProb& P, double* sB, double* red) {
  const int rbA = cdiv_dev(P.C.ks > 0 ? P.C.ks : 1, ROWS);
  for (int by = 0; by < P.C.nchunk_b; by++)
    for (int bx = 0; bx < rbA; bx++) {
      __syncthreads();
      dev_cd_rowmin<FT, false>(P, bx, by, sB, red);
    }
  __syncthreads();
  dev_penalty(P, red);
  __syncthreads();
}
template <int FT>
__device__ __noinline__ void pl_graph(const LoopProb& P, int* ired) {
  const int rb4 = cdiv_dev(P.C.n, 4);
  for (int bx = 0; bx < rb4; bx++) dev_km_csr<FT, 0>(P, bx);
  __syncthreads();
  dev_km_scan_desc(P, ired);
  __syncthreads();
  for (int bx = 0; bx < rb4; bx++) dev_km_csr<FT, 1>(P, bx);
  __syncthreads();
}
template <bool PROF>
__device__ __noinline__ void pl_km(const Km2Problem* desc, int km_flags, char* smem, int lds_bytes) {
  const Km2Problem KP = *desc;
  k4_solve_block<PROF>(KP, km_flags, smem, lds_bytes, nullptr);
  __syncthreads();
}
template <int FT>
__device__ __noinline__ void pl_solve(const LoopProb& P, double* red, int* ired, double* sh) {
  dev_solve<FT>(P, red, ired, sh);
  __syncthreads();
}

template <int FT, bool PROF>
__global__ __launch_bounds__(K4_T, 4) void k_pair_loop(const LoopProb* __restrict__ probs, const int* __restrict__ order, const int npairs, int* qhead,
                                                  const int km_flags, const int lds_bytes, unsigned long long* lstat, int* progress,
                                                  const int* __restrict__ order2, const int npairs2, int* qhead2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sB = reinterpret_cast<double*>(smem);
  double* red = sB + CHUNK_MAX * 3;
  double* sh = red + 16;
  int* ired = reinterpret_cast<int*>(sh + 32);
  volatile int* s_idx = ired + 18;  // (no static LDS in this kernel: the launch may ask for all 160 KB as dynamic LDS)
  const unsigned long long t_slot0 = lstat ? __builtin_amdgcn_s_memrealtime() : 0ull;
  unsigned long long t_solve = 0ull, t_solve_max = 0ull, n_solve = 0ull;
  // A slot is one wave's dependent instruction stream for most of its life (the solver's flood and DFS), and in the tail of a batch it
  // shares its SIMD with the throughput kernels of the next batch's front end: behind 7 ALU-bound waves it would get every eighth issue
  // slot.  The heaviest matrices of the 64 bench scenes are the LATE iterations of the slowest pairs -- exactly the tail --, ~240 ms
  // alone by the model (scripts/km_hazard_survey.py; none of 2220 solves takes the hazard fallback), yet default runs show single solves
  // of 1-4 s (pair_loop_stats.longest_solve_ms).  Highest wave priority: the slot wins the arbitration whenever it can issue at all.
  __builtin_amdgcn_s_setprio(3);
  // (round 6) a slot whose own queue is dry goes on with the queue of the class of SMALLER graphs (order2: they fit its LDS): the slots of the
  // confined three-per-CU class used to leave one by one while the class's last pairs finished, and the launch that re-used their CUs for the
  // other class waited behind the whole kernel in stream order -- 110-180 of 924 slots idle for ~2.2 s of a 9.5 s batch
  // (profiles/r06_call10_log.txt: 924 -> 816 -> 744 before the 988 of the re-launch)
  for (int qsel = 0; qsel < 2; qsel++) {
  const int* const ord = qsel == 0 ? order : order2;
  const int np = qsel == 0 ? npairs : npairs2;
  int* const qh = qsel == 0 ? qhead : qhead2;
  if (ord == nullptr) break;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) *s_idx = atomicAdd(qh, 1);
    __syncthreads();
    const int q = *s_idx;
    if (q >= np) break;
    const LoopProb& P = probs[ord[q]];
    if (threadIdx.x == 0 && lstat) P.st->t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_pair_max = 0ull;
    int it_pair_max = 0;
    while (*(volatile int*)&P.st->done == 0) {
      pl_sweep<FT>(P, sB, red);   // calED + calCD_* + sums + penalty (ghicp_reg.cpp:114-139, 216-341)
      pl_graph<FT>(P, ired);      // the sparse graph of findcorrespondenceKM (ghicp_reg.cpp:348-365): count, scan, fill
      const unsigned long long t0 = lstat ? __builtin_amdgcn_s_memrealtime() : 0ull;
      pl_km<PROF>(P.km_desc, km_flags, smem, lds_bytes);  // Km::kmsolve (km.cpp:40-126)
      if (lstat) {
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
        t_solve += dt; t_solve_max = dt > t_solve_max ? dt : t_solve_max; n_solve++;
        if (dt > t_pair_max) { t_pair_max = dt; it_pair_max = *(volatile int*)&P.st->it; }
      }
      pl_solve<FT>(P, red, ired, sh);  // Km::output, transformestimation, adjustweight (ghicp_reg.cpp:416-460, 605-927)
    }
    if (threadIdx.x == 0 && lstat) {
      P.st->t_end = __builtin_amdgcn_s_memrealtime();
      P.st->t_solve_max = t_pair_max;
      P.st->it_solve_max = it_pair_max;
      // HW_ID (hwreg 4): CU_ID [11:8], SH_ID [12], SE_ID [15:13]; XCC_ID (hwreg 20): [3:0]
      P.st->hw_id = ((unsigned)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) & 0xFFFFu) | (((unsigned)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xFu) << 16);
    }
    if (threadIdx.x == 0 && progress) __hip_atomic_fetch_add(progress, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  }
  if (threadIdx.x == 0 && lstat) {  // launch record: first slot start, last slot end, sum / max of the solve times, solves, sum of slot lifetimes
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    atomicMax(&lstat[0], (1ull << 62) - t_slot0);
    atomicMax(&lstat[1], t1);
    atomicAdd(&lstat[2], t_solve);
    atomicMax(&lstat[3], t_solve_max);
    atomicAdd(&lstat[4], n_solve);
    atomicAdd(&lstat[5], t1 - t_slot0);
    atomicAdd(&lstat[6], 1ull);
  }
}

static void pick_chunks(int ka, int kb, int batch, int* chunk, int* nchunk) {
  // aim for >= ~2048 workgroups per launch (256 CUs x 8) with chunks of at least 32 columns
  const int rowblocks = cdiv(ka > 0 ? ka : 1, ROWS) * (batch > 0 ? batch : 1);
  int want = cdiv(2048, rowblocks);
  int ch = cdiv(kb > 0 ? kb : 1, want);
  if (ch < 32) ch = 32;
  if (ch > CHUNK_MAX) ch = CHUNK_MAX;
  *chunk = ch;
  *nchunk = kb > 0 ? cdiv(kb, ch) : 1;
}

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(char* b) : base(b) {}
  template <typename T> T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

// Launches the persistent pair loop: one launch per LDS-occupancy class of the batch (gh_km4_plan: problems per CU, largest graphs
// first), all classes concurrently -- class 0 on the context's stream, the others on auxiliary streams forked from and joined into
// it -- each with its own queue head.  A launch has at most (slots per CU x CUs) workgroups; every workgroup pops pairs until its
// queue is empty.  Returns when every pair of the batch has converged (or hit max_iter).
// (Error path, round-3 advisor: class launches that are already running write the batch's states; whoever returns early waits for
// every stream first -- gh_join_aux -- so that the caller may reuse the context's buffers.  Auxiliary streams inherit the context's CU
// mask (ghicp_ctx_set_cu_mask); a masked stream handed in through ghicp_ctx_set_stream is not inspected: documented in ghicp_c.h.)
static void gh_join_aux(ghicp_ctx* ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->confine_stream) (void)hipStreamSynchronize(ctx->confine_stream);
  if (ctx->rest_stream) (void)hipStreamSynchronize(ctx->rest_stream);
  for (hipStream_t a : ctx->aux_streams) (void)hipStreamSynchronize(a);
}
#define GH_HIP_JOIN(call)                                                                                     \
  do {                                                                                                        \
    const hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) { gh_join_aux(ctx); return ctx->fail(GHICP_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } \
  } while (0)
template <int FT>
int run_pair_loop(ghicp_ctx* ctx, const LoopProb* dprobs, int nb, const Km4Plan& plan, int* dqheads) {
  hipStream_t s = ctx->stream;
  const int nc = plan.nclass;
  if (nc <= 0) return GHICP_OK;
  while ((int)ctx->aux_streams.size() < nc - 1) {
    hipStream_t a = nullptr;
    if (ctx->cu_mask.empty()) GH_HIP(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    else GH_HIP(hipExtStreamCreateWithCUMask(&a, (uint32_t)ctx->cu_mask.size(), ctx->cu_mask.data()));  // stay on the context's compute units
    ctx->aux_streams.push_back(a);
  }
  while ((int)ctx->aux_events.size() < nc + 2) {
    hipEvent_t e = nullptr;
    GH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->aux_events.push_back(e);
  }
  if (!ctx->progress_host) {
    if (hipHostMalloc((void**)&ctx->progress_host, 64, hipHostMallocMapped) != hipSuccess)
      return ctx->fail(GHICP_ERR_HIP, "pair loop: mapped progress counter allocation failed");
  }
  *(volatile int*)ctx->progress_host = 0;
  ctx->progress_live.store(true, std::memory_order_release);
  const bool prof = ctx->km_stats;
  const void* fn = prof ? reinterpret_cast<const void*>(&k_pair_loop<FT, true>) : reinterpret_cast<const void*>(&k_pair_loop<FT, false>);
  GH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int kflags = ctx->km_force_hazard ? 4 : 0;  // test hook: sends one phase through the hazard fallback
  GH_HIP(hipMemsetAsync(dqheads, 0, 16 * sizeof(int), s));
  // one launch record per BATCH: the classes of a batch share it (first slot start, last slot end, sums over all slots), and the batch's
  // capacity is what can be resident at once: all its workgroups, but not more than the slots of the roomiest class (the classes compete
  // for the same CUs)
  unsigned long long* lstat = nullptr;
  int batch_slots = 0, batch_grid = 0;
  if (ctx->kt_on && ctx->km_launches < ghicp_ctx::KM_LSTAT_MAX) {
    GH_TRY(ctx->reserve(B_KM_LSTAT, (size_t)ghicp_ctx::KM_LSTAT_MAX * ghicp_ctx::KM_LSTAT_W, &lstat));
    lstat += ctx->km_launches * ghicp_ctx::KM_LSTAT_W;
  }
  // ---- two classes, three and four slots per CU: the three-per-CU class on its own CUs (see ghicp_ctx::loop_confine).  Their number follows
  // the class's share w of the batch's work: 3 B slots of 3 B + 4 (CUs - B) should do w of it, with a margin of 15 % on the caller's prior (iterations x n^2 in bench.py: measured, call 6 -- 392.9 -> 421.5 pairs/s on one box; the prior iterations x n with 10-20 % margin, calls 7 and 8, gave the class fewer CUs and the batch a longer span: 375-384) (its queue must not
  // outlast the other one: the four-per-CU slots may use every CU, the confined ones only theirs), spread evenly over the mask's bits.
  int confine_b = 0;
  if (ctx->loop_confine && nc == 2 && plan.per_cu[0] == 3 && plan.per_cu[1] == 4 && plan.count[0] > 0 && plan.count[1] > 0 && ctx->cu_mask.empty() &&
      ctx->num_cu >= 8 && ctx->num_cu <= 2048 && plan.weight[0] > 0 && plan.weight[1] > 0) {
    const double w = std::min(0.9, ctx->loop_confine_margin * plan.weight[0] / (plan.weight[0] + plan.weight[1]));
    int B = (int)std::ceil(w * 4.0 * ctx->num_cu / (3.0 + w));
    B = std::max(B, (plan.count[0] >= 3 ? 1 : 0));
    B = std::min(B, std::min(ctx->num_cu / 2, cdiv(plan.count[0], 3)));
    if (B >= 1) {
      if (ctx->confine_stream == nullptr || ctx->rest_stream == nullptr || ctx->confine_cus != B) {
        // B follows the caller's cost prior from batch to batch: the masked stream pairs are kept per B (a handful of values in practice)
        // instead of being destroyed and re-created -- with a stream synchronisation -- on the launch path (round-5 advisor)
        ctx->confine_stream = ctx->rest_stream = nullptr;
        ctx->confine_cus = 0;
        for (auto& e : ctx->confine_cache)
          if (e.cus == B) { ctx->confine_stream = e.confined; ctx->rest_stream = e.rest; ctx->confine_cus = B; }
        if (ctx->confine_cus != B) {
          if (ctx->confine_cache.size() >= 16) {  // bounded: drop them all (nothing of this context runs on them between two batches)
            for (auto& e : ctx->confine_cache) {
              (void)hipStreamSynchronize(e.confined); (void)hipStreamDestroy(e.confined);
              (void)hipStreamSynchronize(e.rest); (void)hipStreamDestroy(e.rest);
            }
            ctx->confine_cache.clear();
          }
          std::vector<uint32_t> mask((size_t)cdiv(ctx->num_cu, 32), 0u), rest((size_t)cdiv(ctx->num_cu, 32), 0u);
          for (int i = 0; i < ctx->num_cu; i++) {
            const bool in = (long long)(i + 1) * B / ctx->num_cu > (long long)i * B / ctx->num_cu;
            (in ? mask : rest)[(size_t)i >> 5] |= 1u << (i & 31);
          }
          hipStream_t sa = nullptr, sb = nullptr;
          if (hipExtStreamCreateWithCUMask(&sa, (uint32_t)mask.size(), mask.data()) == hipSuccess &&
              hipExtStreamCreateWithCUMask(&sb, (uint32_t)rest.size(), rest.data()) == hipSuccess) {
            ctx->confine_cache.push_back({B, sa, sb});
            ctx->confine_stream = sa; ctx->rest_stream = sb; ctx->confine_cus = B;
          } else {  // no masked streams on this runtime: the batch runs unconfined; the runtime's sticky error must not fail the launch below
            if (sa) (void)hipStreamDestroy(sa);
            if (sb) (void)hipStreamDestroy(sb);
            (void)hipGetLastError();
          }
        }
      }
      if (ctx->confine_cus == B) confine_b = B;
    }
  }
  hipEvent_t kt = ctx->kt_begin(KT_PAIR_LOOP);
  GH_HIP_JOIN(hipEventRecord(ctx->aux_events[0], s));
  bool stole = false;
  for (int c = 0; c < nc; c++) {
    if (plan.count[c] <= 0) continue;
    // confined: the three-per-CU class on its CUs; the four-per-CU class on all the OTHER CUs (if it could use every CU, its slots would
    // take the confined class's CUs first and never leave) and, behind the confined class in stream order, on those CUs as well -- the
    // same queue, so the late slots help drain it
    const bool confined = confine_b > 0;
    hipStream_t sc = confined ? (c == 0 ? ctx->confine_stream : ctx->rest_stream) : (c == 0 ? s : ctx->aux_streams[(size_t)c - 1]);
    if (c > 0 || confined) GH_HIP_JOIN(hipStreamWaitEvent(sc, ctx->aux_events[0], 0));
    const size_t lds = std::max(std::max(plan.lds[c], (size_t)PL_SCRATCH + 64), (size_t)ctx->loop_min_lds);
    int per_cu = 0;
    GH_HIP_JOIN(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, K4_T, lds));
    if (per_cu <= 0) gh_join_aux(ctx);
    if (per_cu <= 0) return ctx->fail(GHICP_ERR_INTERNAL, "pair loop: a workgroup with %zu bytes of LDS does not fit a CU", lds);
    const int slots = per_cu * (confined ? (c == 0 ? confine_b : ctx->num_cu - confine_b) : ctx->num_cu);
    int grid = std::min(plan.count[c], slots);
    if (ctx->loop_slots_cap > 0) grid = std::min(grid, ctx->loop_slots_cap);  // test hook (GHICP_LOOP_SLOTS)
    // capacity of the batch = the most slots the chip can hold at once: the roomiest class on EVERY CU.  (Round 5 added up the confined
    // classes' shares, 3 B + 4 (CUs - B); but once the three-per-CU class has drained, its CUs take four slots of the other class, so the
    // slot lifetimes of a batch could exceed that "capacity" x span: idle_slot_fraction -0.14 in profiles/r05_bench_confine1.json --
    // round-5 verdict, weak #5.  Against this bound the LDS the three-per-CU slots leave unused counts as idle, which it is.)
    batch_slots = std::max(batch_slots, per_cu * ctx->num_cu);
    batch_grid += grid;
    hipEvent_t kd = ctx->kt_begin_on(KT_PAIR_LOOP_DISPATCH, sc);  // this dispatch alone, on its own stream (behind the fork event)
    // the confined class's slots go on with the other class's queue when their own is dry (smaller graphs: they fit)
    const bool steal = confined && c == 0 && plan.lds[1] <= lds;
    stole = stole || steal;
    const int* o2 = steal ? (const int*)(plan.d_order + plan.begin[1]) : (const int*)nullptr;
    const int n2 = steal ? plan.count[1] : 0;
    int* q2 = steal ? dqheads + 1 : (int*)nullptr;
    if (prof)
      hipLaunchKernelGGL((k_pair_loop<FT, true>), dim3(grid), dim3(K4_T), lds, sc, dprobs, (const int*)(plan.d_order + plan.begin[c]), plan.count[c],
                         dqheads + c, kflags, (int)lds, lstat, ctx->progress_host, o2, n2, q2);
    else
      hipLaunchKernelGGL((k_pair_loop<FT, false>), dim3(grid), dim3(K4_T), lds, sc, dprobs, (const int*)(plan.d_order + plan.begin[c]), plan.count[c],
                         dqheads + c, kflags, (int)lds, lstat, ctx->progress_host, o2, n2, q2);
    ctx->kt_end_on(KT_PAIR_LOOP_DISPATCH, kd, sc);
    GH_HIP_JOIN(hipGetLastError());
    if (c > 0 || confined) {
      GH_HIP_JOIN(hipEventRecord(ctx->aux_events[(size_t)c + 1], sc));
      GH_HIP_JOIN(hipStreamWaitEvent(s, ctx->aux_events[(size_t)c + 1], 0));
    }
    if (confined && c == 1 && !stole) {  // ... and the four-per-CU class once more, on the confined CUs, after the three-per-CU class (only when that class's slots could not take the queue over themselves)
      const int grid2 = std::min(plan.count[c], per_cu * confine_b);
      hipEvent_t kd2 = ctx->kt_begin_on(KT_PAIR_LOOP_DISPATCH, ctx->confine_stream);
      if (prof)
        hipLaunchKernelGGL((k_pair_loop<FT, true>), dim3(grid2), dim3(K4_T), lds, ctx->confine_stream, dprobs, (const int*)(plan.d_order + plan.begin[c]),
                           plan.count[c], dqheads + c, kflags, (int)lds, lstat, ctx->progress_host, (const int*)nullptr, 0, (int*)nullptr);
      else
        hipLaunchKernelGGL((k_pair_loop<FT, false>), dim3(grid2), dim3(K4_T), lds, ctx->confine_stream, dprobs, (const int*)(plan.d_order + plan.begin[c]),
                           plan.count[c], dqheads + c, kflags, (int)lds, lstat, ctx->progress_host, (const int*)nullptr, 0, (int*)nullptr);
      ctx->kt_end_on(KT_PAIR_LOOP_DISPATCH, kd2, ctx->confine_stream);
      GH_HIP_JOIN(hipGetLastError());
      batch_grid += grid2;
      GH_HIP_JOIN(hipEventRecord(ctx->aux_events[(size_t)nc + 1], ctx->confine_stream));
      GH_HIP_JOIN(hipStreamWaitEvent(s, ctx->aux_events[(size_t)nc + 1], 0));
    }
  }
  ctx->kt_end(KT_PAIR_LOOP, kt);
  if (lstat) {
    ctx->km_slots.push_back(std::min(batch_slots, batch_grid));
    ctx->km_launches++;
  }
  GH_HIP_JOIN(hipStreamSynchronize(s));
  return GHICP_OK;
}
#undef GH_HIP_JOIN

template <int FT>
int run_loops(ghicp_ctx* ctx, int nb, const gh_loop_job* jobs) {
  hipStream_t s = ctx->stream;
  // ghicp_ctx_set_loop_cost_hints: "consumed by the next registration call of this context, ignored otherwise" -- taken here, whatever
  // path the batch takes (round-4 advisor: cleared only in the Kuhn-Munkres branch, hints survived a batch of another kind)
  std::vector<float> cost_hints;
  cost_hints.swap(ctx->loop_cost_hints);
  const ghicp_params* p0 = jobs[0].p;
  const int corr = p0->corr;
  int max_iter = 1;
  for (int b = 0; b < nb; b++) max_iter = jobs[b].p->max_iter > max_iter ? jobs[b].p->max_iter : max_iter;
  // exp(-it/rate) from the host libm, exactly as calCD_BSC computes it (ghicp_reg.cpp:247); one table per batch
  std::vector<double> wtab(max_iter + 1);
  for (int i = 0; i <= max_iter; i++) wtab[i] = std::exp(-1.0 * i / p0->weight_changing_rate);

  // ---- size pass, then carve every pair's buffers out of one allocation
  std::vector<LoopProb> hp(nb);
  std::vector<LoopState> hst(nb);
  size_t total = 0;
  int max_rowsA = 1, max_chunkB = 1, max_rowsB = 1, max_chunkA = 1, max_n = 1;
  // Kuhn-Munkres batches whose every graph fits the LDS-resident solver run as the persistent pair loop (k_pair_loop)
  bool persistent = corr == GHICP_CORR_KM;
  for (int b = 0; b < nb && persistent; b++) {
    const int n = std::max(jobs[b].ks, jobs[b].kt);
    persistent = (jobs[b].ks <= 0 || jobs[b].kt <= 0) || gh_km4_fits(n);
  }
  const int chunk_batch = persistent ? (1 << 20) : nb;  // one workgroup sweeps a pair: the largest chunks (fewest partial sums)
  for (int pass = 0; pass < 2; pass++) {
    char* arena = nullptr;
    if (pass == 1) GH_TRY(ctx->reserve(B_LOOP_STATE, total + 4096, &arena));
    Carver cv(arena);
    const double* wfd = cv.take<double>(wtab.size());
    LoopProb* dprobs = cv.take<LoopProb>(nb);
    int* dflags = cv.take<int>((size_t)nb * 2);
    int* dqheads = cv.take<int>(16);  // queue heads of the persistent pair loop, one per class
    Km2Problem* d_descs = cv.take<Km2Problem>(nb);  // contiguous: the dense-fallback path launches one solve kernel over all of them
    LoopState* dstates = cv.take<LoopState>(nb);    // contiguous states and solver status words: ONE upload / memset / download per batch
    int* dkmst = cv.take<int>((size_t)nb + 1);
    int max_ks = 1, max_kt = 1;
    bool need_transpose = false;
    for (int b = 0; b < nb; b++) {
      const gh_loop_job& J = jobs[b];
      const ghicp_params* p = J.p;
      LoopProb& L = hp[b];
      memset(&L, 0, sizeof(L));
      LoopConst& C = L.C;
      const int ks = J.ks, kt = J.kt;
      C.ks = ks; C.kt = kt; C.n = ks > kt ? ks : kt; C.feature = p->feature; C.corr = p->corr; C.max_iter = p->max_iter; C.min_cor = p->min_cor;
      C.scale = (float)(0.005 * p->bbx_magnitude);  // ghicp_reg.h:40 (double product stored to float)
      C.est_iou = p->est_iou; C.adjust_ratio = p->adjust_ratio; C.adjust_step = p->adjust_step;
      C.converge_t = (double)p->converge_t; C.converge_r = (double)p->converge_r; C.penalty_initial = p->penalty_initial; C.km_eps = p->km_eps;
      pick_chunks(ks, kt, chunk_batch, &C.chunk_b, &C.nchunk_b);
      pick_chunks(kt, ks, chunk_batch, &C.chunk_a, &C.nchunk_a);
      C.nparts = cdiv(ks > 0 ? ks : 1, ROWS) * C.nchunk_b;
      max_rowsA = std::max(max_rowsA, cdiv(ks > 0 ? ks : 1, ROWS)); max_chunkB = std::max(max_chunkB, C.nchunk_b);
      max_rowsB = std::max(max_rowsB, cdiv(kt > 0 ? kt : 1, ROWS)); max_chunkA = std::max(max_chunkA, C.nchunk_a);
      max_n = std::max(max_n, C.n);
      L.wfd = wfd;
      L.st = dstates + b;
      L.kpS_src = J.kpS;
      max_ks = std::max(max_ks, ks); max_kt = std::max(max_kt, kt);
      L.kpS = cv.take<double>((size_t)ks * 3 + 3);
      L.kpT = J.kpT; L.FD = J.FD;
      L.pminA = cv.take<double>((size_t)C.nchunk_b * ks + 1);
      L.pidxA = cv.take<int>((size_t)C.nchunk_b * ks + 1);
      L.psum = cv.take<double>((size_t)C.nparts * 2 + 2);
      if (corr == GHICP_CORR_NNR) {
        L.pminB = cv.take<double>((size_t)C.nchunk_a * kt + 1);
        L.pidxB = cv.take<int>((size_t)C.nchunk_a * kt + 1);
      }
      L.SP = cv.take<int>((size_t)C.n + 1);
      L.TP = cv.take<int>((size_t)C.n + 1);
      L.SVs = cv.take<int>((size_t)ks + 1);
      L.TVs = cv.take<int>((size_t)kt + ks + 2);
      L.trace = cv.take<ghicp_iter>((size_t)p->max_iter + 1);
      L.matchlist = J.matchlist;
      L.ml_row0 = J.ml_row0;
      if (FT != GHICP_FEATURE_NONE) {
        if (J.FDt) L.FDt = J.FDt;  // the caller's batched feature-distance kernel wrote the transposed copy already
        else { L.FDt = cv.take<char>((size_t)ks * kt * (FT == GHICP_FEATURE_BSC ? 2 : 4) + 16); need_transpose = true; }
        L.fdt_given = J.FDt != nullptr;
      }
      if (corr == GHICP_CORR_KM) {
        L.kmmatch = cv.take<int>((size_t)C.n + 1);
        L.km_status = dkmst + b;
        if (gh_km4_fits(C.n)) {
          L.km_cnt = cv.take<unsigned>((size_t)C.n + 1);
          L.km_rptr = cv.take<unsigned>((size_t)C.n + 2);
          L.km_lx = cv.take<double>((size_t)C.n + 1);
          L.km_cols = cv.take<int>((size_t)ks * kt + 1);
          L.km_vals = cv.take<double>((size_t)ks * kt + 1);
          L.km_desc = d_descs ? d_descs + b : nullptr;
        } else {
          L.kmw = cv.take<double>((size_t)C.n * C.n + 1);
        }
      }
    }
    total = cv.off;
    if (pass == 1) {
      // ---- upload tables, states, descriptors
      GH_HIP(hipMemcpyAsync(const_cast<double*>(wfd), wtab.data(), wtab.size() * sizeof(double), hipMemcpyHostToDevice, s));
      GH_HIP(hipMemsetAsync(d_descs, 0, (size_t)nb * sizeof(Km2Problem), s));  // n = 0: "nothing to solve"
      for (int b = 0; b < nb; b++) {
        LoopState& h = hst[b];
        memset(&h, 0, sizeof(h));
        h.RMS = 99999; h.para1 = jobs[b].p->para1; h.para2 = jobs[b].p->para2;  // ghicp_reg.h:98, 33-34
        for (int d = 0; d < 4; d++) h.Rt_till[d * 5] = 1.0;
        if (jobs[b].resume_in) {  // ghicp_iterate: the loop continues from the state the previous call left
          memcpy(&h, jobs[b].resume_in, sizeof(h));
          h.done = 0;
        }
        if (jobs[b].ks <= 0 || jobs[b].kt <= 0) h.done = 1;
      }
      GH_HIP(hipMemcpyAsync(dstates, hst.data(), (size_t)nb * sizeof(LoopState), hipMemcpyHostToDevice, s));
      GH_HIP(hipMemsetAsync(dkmst, 0, ((size_t)nb + 1) * sizeof(int), s));
      GH_HIP(hipMemcpyAsync(dprobs, hp.data(), (size_t)nb * sizeof(LoopProb), hipMemcpyHostToDevice, s));
      // the hand-over of the whole batch: source keypoints into the pairs' own buffers, feature matrices transposed (k_pairs_*)
      for (int b0 = 0; b0 < nb; b0 += 65535)  // gridDim.y <= 65535
        hipLaunchKernelGGL(k_pairs_copy_kps, dim3(std::min(cdiv(max_ks * 3, 256), 8), std::min(65535, nb - b0)), dim3(256), 0, s, (const LoopProb*)dprobs, b0);
      if (FT != GHICP_FEATURE_NONE && need_transpose) {
        const int tx = cdiv(max_kt, 32), ty = cdiv(max_ks, 32);
        const int zmax = (int)std::max<long long>(1, std::min<long long>(65535, (1ll << 22) / ((long long)tx * ty)));  // <= 4 M workgroups (2^30 threads) a launch
        for (int b0 = 0; b0 < nb; b0 += zmax) {
          const dim3 g(tx, ty, std::min(zmax, nb - b0));
          if (FT == GHICP_FEATURE_BSC) hipLaunchKernelGGL(k_pairs_transpose<uint16_t>, g, dim3(32, 8), 0, s, (const LoopProb*)dprobs, b0);
          else hipLaunchKernelGGL(k_pairs_transpose<float>, g, dim3(32, 8), 0, s, (const LoopProb*)dprobs, b0);
        }
      }
      GH_HIP(hipGetLastError());

      // ---- iterate
      ctx->loop_total.store(nb, std::memory_order_relaxed);
      ctx->loop_active.store(nb, std::memory_order_relaxed);
      std::vector<int> hflags((size_t)nb * 2, 0);
      const int poll_every = 2;
      int launched = 0;
      bool all_done = false;
      bool any_dense = false;
      for (int b = 0; b < nb; b++) any_dense |= (hp[b].kmw != nullptr);
      bool any_sparse = false;
      int max_n_sparse = 1;
      for (int b = 0; b < nb; b++)
        if (hp[b].km_rptr) { any_sparse = true; max_n_sparse = std::max(max_n_sparse, hp[b].C.n); }
      // the Kuhn-Munkres launches of this batch: problems grouped by LDS occupancy, largest first (km4.hip)
      Km4Plan km_plan;
      bool use_plan = false;
      if (any_sparse && !any_dense && gh_km4_fits(max_n_sparse)) {
        std::vector<int> hn((size_t)nb);
        bool all_sparse = true;
        for (int b = 0; b < nb; b++) { hn[b] = hp[b].C.n; all_sparse &= (hp[b].km_rptr != nullptr) || jobs[b].ks <= 0 || jobs[b].kt <= 0; }
        if (all_sparse) {
          // cost hints of the caller for exactly this batch (ghicp_ctx_set_loop_cost_hints): consumed once
          const bool hinted = (int)cost_hints.size() == nb;
          GH_TRY(gh_km4_plan(ctx, hn.data(), nb, &km_plan, hinted ? cost_hints.data() : nullptr));
          use_plan = true;
        }
      }
      if (persistent && use_plan) {
        const int rc = run_pair_loop<FT>(ctx, dprobs, nb, km_plan, dqheads);
        if (rc != GHICP_OK) {
          ctx->progress_live.store(false, std::memory_order_release);
          return rc;
        }
        all_done = true;
      }
      while (!all_done && launched < max_iter) {
        for (int r = 0; r < poll_every && launched < max_iter; r++, launched++) {
          hipEvent_t kev = ctx->kt_begin(KT_CD_ROWMIN);
          hipLaunchKernelGGL((k_cd_rowmin<FT, false>), dim3(max_rowsA, max_chunkB, nb), dim3(ROWS), 0, s, dprobs);
          ctx->kt_end(KT_CD_ROWMIN, kev);
          if (corr == GHICP_CORR_NNR) hipLaunchKernelGGL((k_cd_rowmin<FT, true>), dim3(max_rowsB, max_chunkA, nb), dim3(ROWS), 0, s, dprobs);
          hipLaunchKernelGGL(k_penalty, dim3(nb), dim3(256), 0, s, dprobs);
          if (corr == GHICP_CORR_KM) {
            hipEvent_t kw = ctx->kt_begin(KT_KM_WEIGHTS);
            hipLaunchKernelGGL((k_km_csr<FT, 0>), dim3(cdiv(max_n, 4), nb), dim3(256), 0, s, dprobs);
            hipLaunchKernelGGL(k_km_scan_desc, dim3(nb), dim3(1024), 0, s, dprobs);
            hipLaunchKernelGGL((k_km_csr<FT, 1>), dim3(cdiv(max_n, 4), nb), dim3(256), 0, s, dprobs);
            ctx->kt_end(KT_KM_WEIGHTS, kw);
            if (any_sparse) GH_TRY(use_plan ? gh_km4_launch_plan(ctx, d_descs, km_plan) : gh_km4_launch(ctx, d_descs, nb, max_n_sparse));
            if (any_dense) {  // matrices too large for the LDS-resident solver: dense fallback, one pair at a time
              hipLaunchKernelGGL(k_km_weights<FT>, dim3(cdiv(max_n, 256), max_n, nb), dim3(256), 0, s, dprobs);
              for (int b = 0; b < nb; b++)
                if (hp[b].kmw) GH_TRY(gh_km_solve_dev(ctx, hp[b].kmw, hp[b].C.n, hp[b].C.km_eps, hp[b].kmmatch, &hp[b].st->done));
            }
          }
          // 256 threads: four waves of this kernel (108 VGPRs) fit next to the Kuhn-Munkres waves of other batches on a CU;
          // a 1024-thread block needs a CU with no resident solve wave and stalls for a whole solve launch when batches overlap
          hipLaunchKernelGGL(k_solve<FT>, dim3(nb), dim3(256), 0, s, dprobs);
        }
        GH_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_collect_done, dim3(cdiv(nb, 256)), dim3(256), 0, s, dprobs, nb, dflags);
        GH_HIP(hipMemcpyAsync(hflags.data(), dflags, hflags.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        GH_HIP(hipStreamSynchronize(s));
        all_done = true;
        long long still = 0;
        for (int b = 0; b < nb; b++) { all_done &= (hflags[(size_t)b * 2 + 1] != 0); still += hflags[(size_t)b * 2 + 1] == 0; }
        ctx->loop_active.store(still, std::memory_order_relaxed);
      }
      ctx->loop_active.store(0, std::memory_order_relaxed);
      ctx->progress_live.store(false, std::memory_order_release);
      // ---- results
      std::vector<int> hkmst((size_t)nb + 1, 0);
      GH_HIP(hipMemcpyAsync(hst.data(), dstates, (size_t)nb * sizeof(LoopState), hipMemcpyDeviceToHost, s));
      GH_HIP(hipMemcpyAsync(hkmst.data(), dkmst, ((size_t)nb + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
      GH_HIP(hipStreamSynchronize(s));
      if (ctx->kt_on && persistent) {  // slot timeline of the batch (diagnostics, ghicp_ctx_loop_timeline): per pair begin / end / iterations
        ctx->loop_timeline.resize((size_t)nb * 3);
        for (int b = 0; b < nb; b++) {
          // iterations [15:0] | iteration of the pair's longest solve [31:16] | that solve in units of 16 ticks = 160 ns [63:32]; where the slot
          // ran rides in the top bits of `begin` (ticks since boot need 48 bits): CU [55:52], shader array [56], engine [59:57], die [63:60]
          const unsigned hw = hst[b].hw_id;
          const unsigned long long where = (unsigned long long)((hw >> 8) & 0xFu) | ((unsigned long long)((hw >> 12) & 1u) << 4) | ((unsigned long long)((hw >> 13) & 7u) << 5) |
                                           ((unsigned long long)((hw >> 16) & 0xFu) << 8);
          ctx->loop_timeline[(size_t)b * 3] = (long long)((hst[b].t_begin & 0x000FFFFFFFFFFFFFull) | (where << 52));
          ctx->loop_timeline[(size_t)b * 3 + 1] = (long long)(hst[b].t_end & 0x000FFFFFFFFFFFFFull);
          ctx->loop_timeline[(size_t)b * 3 + 2] = (long long)(((unsigned long long)(hst[b].it & 0xFFFF)) | ((unsigned long long)(hst[b].it_solve_max & 0xFFFF) << 16) |
                                                              (std::min<unsigned long long>(hst[b].t_solve_max >> 4, 0xFFFFFFFFull) << 32));
        }
      }
      for (int b = 0; b < nb; b++) {
        const gh_loop_job& J = jobs[b];
        for (int d = 0; d < 16; d++) J.Rt16[d] = hst[b].Rt_till[d];
        if (J.n_iter) *J.n_iter = hst[b].it;
        if (J.converged) *J.converged = hst[b].converged_flag;
        if (J.rmse_after) *J.rmse_after = hst[b].rmse_after;
        if (J.trace && hst[b].it > 0) GH_HIP(hipMemcpyAsync(J.trace, hp[b].trace, (size_t)hst[b].it * sizeof(ghicp_iter), hipMemcpyDeviceToHost, s));
        if (J.trace_last && hst[b].it > 0) GH_HIP(hipMemcpyAsync(J.trace_last, hp[b].trace + (hst[b].it - 1), sizeof(ghicp_iter), hipMemcpyDeviceToHost, s));
        if (J.kpS_out && J.ks > 0) GH_HIP(hipMemcpyAsync(J.kpS_out, hp[b].kpS, (size_t)J.ks * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (J.resume_out) memcpy(J.resume_out, &hst[b], sizeof(LoopState));
      }
      GH_HIP(hipStreamSynchronize(s));
      int kmst = 0;
      for (int b = 0; b < nb; b++)
        if (hp[b].km_status) {
          kmst |= hkmst[(size_t)b] & 0xFFFF;
          ctx->loop_hazards += (long long)((unsigned)hkmst[(size_t)b] >> 16);  // solves through the literal fallback (diagnostics)
        }
      if (any_dense && ctx->buf[B_KM_MISC].p) {  // the dense fallback reports through the context's own status word
        int v = 0;
        GH_HIP(hipMemcpy(&v, ctx->buf[B_KM_MISC].p, sizeof(int), hipMemcpyDeviceToHost));
        kmst |= v;
      }
      if (kmst) return ctx->fail(GHICP_ERR_INTERNAL, "KM solver status %d (non-finite energy?)", kmst);
    }
  }
  return GHICP_OK;
}

}  // namespace

int gh_register_batch_dev(ghicp_ctx* ctx, int nb, const gh_loop_job* jobs) {
  if (nb <= 0) return GHICP_OK;
  for (int b = 0; b < nb; b++) {
    const gh_loop_job& J = jobs[b];
    GH_ARG(J.p != nullptr && J.Rt16 != nullptr);
    GH_ARG(J.ks >= 0 && J.kt >= 0 && J.p->max_iter > 0 && J.p->max_iter <= 100000);
    GH_ARG(J.p->corr == GHICP_CORR_NN || J.p->corr == GHICP_CORR_NNR || J.p->corr == GHICP_CORR_KM);
    GH_ARG(J.p->feature == jobs[0].p->feature && J.p->corr == jobs[0].p->corr && J.p->weight_changing_rate == jobs[0].p->weight_changing_rate);
    if (J.p->feature == GHICP_FEATURE_BSC || J.p->feature == GHICP_FEATURE_FPFH) GH_ARG(J.FD != nullptr || J.ks == 0 || J.kt == 0);
  }
  switch (jobs[0].p->feature) {
    case GHICP_FEATURE_BSC: return run_loops<GHICP_FEATURE_BSC>(ctx, nb, jobs);
    case GHICP_FEATURE_FPFH: return run_loops<GHICP_FEATURE_FPFH>(ctx, nb, jobs);
    case GHICP_FEATURE_NONE:
    case GHICP_FEATURE_ROPS:  // test/ghicp_main.cpp:130-134 falls through with no feature
      return run_loops<GHICP_FEATURE_NONE>(ctx, nb, jobs);
    default: return ctx->fail(GHICP_ERR_ARG, "unknown feature type %d", jobs[0].p->feature);
  }
}

int gh_register_dev(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS, int ks, const double* kpT, int kt, const void* FD,
                    double* Rt16, ghicp_iter* trace, int32_t* n_iter, int32_t* matchlist) {
  gh_loop_job J;
  memset(&J, 0, sizeof(J));
  J.p = p; J.kpS = kpS; J.ks = ks; J.kpT = kpT; J.kt = kt; J.FD = FD; J.Rt16 = Rt16; J.trace = trace; J.n_iter = n_iter; J.matchlist = matchlist;
  return gh_register_batch_dev(ctx, 1, &J);
}

extern "C" int ghicp_register(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS, int64_t ks, const double* kpT, int64_t kt,
                              const void* FD, double* Rt16, ghicp_iter* trace, int32_t* n_iter, int32_t* matchlist) {
  GH_ENTER(ctx);
  GH_ARG(p != nullptr && ks >= 0 && kt >= 0 && ks < (1 << 24) && kt < (1 << 24));
  Stager sg(ctx);
  const double *dS, *dT;
  GH_TRY(sg.in(kpS, (size_t)ks * 3, &dS));
  GH_TRY(sg.in(kpT, (size_t)kt * 3, &dT));
  const size_t esz = p->feature == GHICP_FEATURE_BSC ? 2 : 4;
  const char* dFD = nullptr;
  if (p->feature == GHICP_FEATURE_BSC || p->feature == GHICP_FEATURE_FPFH) GH_TRY(sg.in((const char*)FD, (size_t)ks * kt * esz, &dFD));
  int32_t* dml;
  GH_TRY(sg.out(matchlist, (size_t)p->max_iter * ks, &dml));
  GH_TRY(gh_register_dev(ctx, p, dS, (int)ks, dT, (int)kt, dFD, Rt16, trace, n_iter, dml));
  return sg.finish();
}

// ---- One step at a time (ghicp_iterate = one pass of calED ... adjustweight, src/ghicp_reg.cpp:49-103).  The state object owns
// the moving source keypoints and the scalar loop state (Rt_tillnow, RMS, FDM, FDstd, IoU, para1/2, iteration number) between calls; an
// iteration runs through the SAME batch path as ghicp_register (a batch of one pair, max_iter = it + 1), so a sequence of ghicp_iterate calls
// reproduces ghicp_register's trace bit for bit (tests/test_gpu_loop.py::test_iterate_equals_register).
struct ghicp_loop {
  ghicp_ctx* ctx;
  ghicp_params p;
  int ks, kt;
  DevBuf kpS, kpT, FD;  // own device copies: the caller's arrays may go away between two calls
  LoopState st;
  bool started, finished;
};

extern "C" int ghicp_loop_create(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS, int64_t ks, const double* kpT, int64_t kt, const void* FD,
                                 ghicp_loop** out) {
  GH_ENTER(ctx);
  GH_ARG(p != nullptr && out != nullptr && ks >= 0 && kt >= 0 && ks < (1 << 24) && kt < (1 << 24));
  GH_ARG(p->corr == GHICP_CORR_NN || p->corr == GHICP_CORR_NNR || p->corr == GHICP_CORR_KM);
  const bool has_fd = p->feature == GHICP_FEATURE_BSC || p->feature == GHICP_FEATURE_FPFH;
  if (has_fd) GH_ARG(FD != nullptr || ks == 0 || kt == 0);
  *out = nullptr;
  ghicp_loop* L = new ghicp_loop();
  L->ctx = ctx; L->p = *p; L->ks = (int)ks; L->kt = (int)kt; L->started = false; L->finished = false;
  memset(&L->st, 0, sizeof(L->st));
  const hipMemcpyKind kind = ctx->host_ptrs ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  const size_t esz = p->feature == GHICP_FEATURE_BSC ? 2 : 4;
  hipError_t e = L->kpS.reserve((size_t)ks * 24 + 32);
  if (e == hipSuccess) e = L->kpT.reserve((size_t)kt * 24 + 32);
  if (e == hipSuccess && has_fd) e = L->FD.reserve((size_t)ks * kt * esz + 32);
  if (e == hipSuccess && ks > 0) e = hipMemcpyAsync(L->kpS.p, kpS, (size_t)ks * 24, kind, ctx->stream);
  if (e == hipSuccess && kt > 0) e = hipMemcpyAsync(L->kpT.p, kpT, (size_t)kt * 24, kind, ctx->stream);
  if (e == hipSuccess && has_fd && ks > 0 && kt > 0) e = hipMemcpyAsync(L->FD.p, FD, (size_t)ks * kt * esz, kind, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    L->kpS.release(); L->kpT.release(); L->FD.release();
    delete L;
    return ctx->fail(GHICP_ERR_HIP, "ghicp_loop_create: %s", hipGetErrorString(e));
  }
  *out = L;
  return GHICP_OK;
}

extern "C" int ghicp_iterate(ghicp_ctx* ctx, ghicp_loop* L, ghicp_iter* out, int32_t* match_row) {
  GH_ENTER(ctx);
  GH_ARG(L != nullptr && L->ctx == ctx && out != nullptr);
  if (L->finished) return ctx->fail(GHICP_ERR_ARG, "ghicp_iterate: the loop has converged (or has no keypoints); create a new one");
  ghicp_params p = L->p;
  p.max_iter = (L->started ? L->st.it : 0) + 1;  // exactly one more iteration
  Stager sg(ctx);
  int32_t* dml;
  GH_TRY(sg.out(match_row, (size_t)L->ks, &dml));
  double Rt16[16];
  int32_t n_iter = 0, conv = 0;
  gh_loop_job J;
  memset(&J, 0, sizeof(J));
  J.p = &p; J.kpS = L->kpS.as<double>(); J.ks = L->ks; J.kpT = L->kpT.as<double>(); J.kt = L->kt; J.FD = L->FD.p; J.Rt16 = Rt16;
  J.n_iter = &n_iter; J.converged = &conv; J.matchlist = dml; J.ml_row0 = p.max_iter - 1;
  J.resume_in = L->started ? &L->st : nullptr; J.resume_out = &L->st; J.kpS_out = L->kpS.as<double>(); J.trace_last = out;
  memset(out, 0, sizeof(*out));
  GH_TRY(gh_register_batch_dev(ctx, 1, &J));
  GH_HIP(hipStreamSynchronize(ctx->stream));
  L->started = true;
  if (L->ks <= 0 || L->kt <= 0 || conv) L->finished = true;
  if (L->ks <= 0 || L->kt <= 0) out->converged = 1;
  return sg.finish();
}

extern "C" int ghicp_loop_result(const ghicp_loop* L, double* Rt16, int32_t* n_iter, int32_t* converged, double* rmse_after) {
  if (!L || !Rt16) return GHICP_ERR_ARG;
  for (int d = 0; d < 16; d++) Rt16[d] = L->started ? L->st.Rt_till[d] : (d % 5 == 0 ? 1.0 : 0.0);
  if (n_iter) *n_iter = L->started ? L->st.it : 0;
  if (converged) *converged = L->started ? L->st.converged_flag : 0;
  if (rmse_after) *rmse_after = L->started ? L->st.rmse_after : 0.0;
  return GHICP_OK;
}

extern "C" void ghicp_loop_destroy(ghicp_loop* L) {
  if (!L) return;
  (void)hipSetDevice(L->ctx->device);
  L->kpS.release(); L->kpT.release(); L->FD.release();
  delete L;
}
