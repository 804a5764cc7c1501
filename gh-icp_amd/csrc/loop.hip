// GH-ICP iteration loop on gfx950: replaces GHRegistration::ghicp_reg and its private helpers
// (reference src/ghicp_reg.cpp:24-112, 114-139, 216-341, 343-460, 548-578, 605-927).
//
// Per iteration (no host round-trip; the host only polls `done` every few iterations):
//   k_cd_rowmin      fused calED + calCD_* + row arg-min (+ column arg-min pass for NNR) + sum/sum^2
//                    over K_S x K_T; no f64 ED/CD matrix is ever materialised           (S5, HBM-bound)
//   k_penalty        CDmean/CDstd -> penalty (calCD_* tail)                               (scalar)
//   [KM] k_km_weights + km_solve (km.hip)                                                 (S5 KM)
//   k_solve          accept correspondences, RMSE/FDM/FDstd, float-Umeyama rigid solve, apply to all
//                    source keypoints, RMSE-after, Euler convergence test, adjustweight, Rt product (S6)
#include "ctx.h"
#include "devmath.h"

#include <cmath>

namespace {

struct LoopState {
  int it, done, cor, converged_flag;
  double RMS, FDM, FDstd, IoU, para1, para2, penalty, CDmean, CDstd, energy;
  double Rt_till[16];
};

struct LoopConst {
  int ks, kt, n, feature, corr, max_iter, min_cor, nchunk_a, nchunk_b, chunk_a, chunk_b;
  float scale, est_iou, adjust_ratio, adjust_step;
  double converge_t, converge_r, penalty_initial, km_eps;
};

constexpr int ROWS = 256;      // threads per block in the sweep = rows handled per block
constexpr int CHUNK_MAX = 512;  // columns staged in LDS per block

// One sweep: thread = "row" a (keypoint of set A), loop over a chunk of set B staged in LDS.
// FDt is [kb][ka] so that lanes (consecutive a) read consecutive addresses.
// Row arg-min semantics = ghicp_reg.cpp:715-724 / 622-650: start (9e20, 0), strict '<', ascending index.
template <int FT, bool SUMS>
__global__ __launch_bounds__(ROWS) void k_cd_rowmin(const LoopState* __restrict__ st, const double* __restrict__ A, int ka,
                                                     const double* __restrict__ B, int kb, const void* __restrict__ FDt, int chunk,
                                                     float scale, const double* __restrict__ wfd_tab, double* __restrict__ part_min,
                                                     int* __restrict__ part_idx, double* __restrict__ part_sum) {
  if (st->done) return;
  __shared__ double sB[CHUNK_MAX * 3];
  __shared__ double red[16];
  const int it = st->it;
  const int jb = blockIdx.y * chunk;
  const int je = min(kb, jb + chunk);
  for (int t = threadIdx.x; t < (je - jb) * 3; t += ROWS) sB[t] = B[(size_t)jb * 3 + t];
  __syncthreads();
  const int a = blockIdx.x * ROWS + threadIdx.x;
  const bool live = a < ka;
  double ax = 0, ay = 0, az = 0;
  if (live) { ax = A[(size_t)a * 3]; ay = A[(size_t)a * 3 + 1]; az = A[(size_t)a * 3 + 2]; }
  double wfd = 0, wed = 1, inv_k = 1;
  if (FT == GHICP_FEATURE_BSC) { wfd = wfd_tab[it]; wed = 1.0 - wfd; }
  if (FT == GHICP_FEATURE_FPFH) inv_k = 1.0 / (double)(it + 1);
  const double dscale = (double)scale;
  double best = 9e20, s = 0, s2 = 0;
  int bidx = 0;
  if (live) {
    for (int j = jb; j < je; j++) {
      const double dx = ax - sB[(j - jb) * 3], dy = ay - sB[(j - jb) * 3 + 1], dz = az - sB[(j - jb) * 3 + 2];
      const double ed = dscale * sqrt(dx * dx + dy * dy + dz * dz);
      double cd;
      if (FT == GHICP_FEATURE_BSC) {
        const double fd = (double)reinterpret_cast<const uint16_t*>(FDt)[(size_t)j * ka + a];
        cd = wed * ed + wfd * fd;  // ghicp_reg.cpp:259
      } else if (FT == GHICP_FEATURE_FPFH) {
        const double fd = (double)reinterpret_cast<const float*>(FDt)[(size_t)j * ka + a];
        cd = 1.0 * ed / pow(fd, inv_k);  // ghicp_reg.cpp:308
      } else {
        cd = ed;  // ghicp_reg.cpp:224
      }
      if (cd < best) { best = cd; bidx = j; }
      if (SUMS) { s += cd; s2 += cd * cd; }
    }
    part_min[(size_t)blockIdx.y * ka + a] = best;
    part_idx[(size_t)blockIdx.y * ka + a] = bidx;
  }
  if (SUMS) {
    const double bs = gh_block_sum(s, red);
    const double bs2 = gh_block_sum(s2, red);
    if (threadIdx.x == 0) {
      const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
      part_sum[b * 2] = bs;
      part_sum[b * 2 + 1] = bs2;
    }
  }
}

// calCD_* tails: CDmean, CDstd, penalty (ghicp_reg.cpp:228-239, 264-287, 317-335)
__global__ __launch_bounds__(256) void k_penalty(LoopState* st, LoopConst C, const double* __restrict__ part_sum, int nparts,
                                                 const double* __restrict__ wfd_tab) {
  if (st->done) return;
  __shared__ double red[16];
  double s = 0, s2 = 0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) { s += part_sum[i * 2]; s2 += part_sum[i * 2 + 1]; }
  s = gh_block_sum(s, red);
  s2 = gh_block_sum(s2, red);
  if (threadIdx.x == 0) {
    const int it = st->it;
    const double cnt = (double)C.ks * (double)C.kt;
    const double mean = s / (double)C.kt / (double)C.ks;
    double var = s2 / cnt - mean * mean;
    if (var < 0) var = 0;
    const double sd = sqrt(var);
    double pen;
    if (C.feature == GHICP_FEATURE_NONE) {
      pen = fmax(mean, 1.0);  // Q6: line 239 overrides 230-237
      st->CDstd = 0;
    } else if (C.feature == GHICP_FEATURE_BSC) {
      const double wfd = wfd_tab[it], wed = 1.0 - wfd;
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

// KM weights (ghicp_reg.cpp:348-365): w[i][j] = -CD if CD < penalty else -penalty, padded to n x n.
template <int FT>
__global__ __launch_bounds__(256) void k_km_weights(const LoopState* __restrict__ st, LoopConst C, const double* __restrict__ kpS,
                                                    const double* __restrict__ kpT, const void* __restrict__ FD,
                                                    const double* __restrict__ wfd_tab, double* __restrict__ w) {
  if (st->done) return;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= C.n) return;
  const double pen = st->penalty;
  double out = -pen;
  if (i < C.ks && j < C.kt) {
    const int it = st->it;
    const double dx = kpS[(size_t)i * 3] - kpT[(size_t)j * 3], dy = kpS[(size_t)i * 3 + 1] - kpT[(size_t)j * 3 + 1],
                 dz = kpS[(size_t)i * 3 + 2] - kpT[(size_t)j * 3 + 2];
    const double ed = (double)C.scale * sqrt(dx * dx + dy * dy + dz * dz);
    double cd;
    if (FT == GHICP_FEATURE_BSC) {
      const double wfd = wfd_tab[it], wed = 1.0 - wfd;
      cd = wed * ed + wfd * (double)reinterpret_cast<const uint16_t*>(FD)[(size_t)i * C.kt + j];
    } else if (FT == GHICP_FEATURE_FPFH) {
      cd = 1.0 * ed / pow((double)reinterpret_cast<const float*>(FD)[(size_t)i * C.kt + j], 1.0 / (double)(it + 1));
    } else {
      cd = ed;
    }
    if (cd < pen) out = -cd;
  }
  w[(size_t)i * C.n + j] = out;
}

// Everything after the sweep, one 1024-thread workgroup.
template <int FT>
__global__ __launch_bounds__(1024) void k_solve(LoopState* st, LoopConst C, double* __restrict__ kpS, const double* __restrict__ kpT,
                                                const void* __restrict__ FD, const double* __restrict__ pminA,
                                                const int* __restrict__ pidxA, const int* __restrict__ pidxB,
                                                const double* __restrict__ pminB, const double* __restrict__ kmw,
                                                const int* __restrict__ kmmatch, int* __restrict__ SP, int* __restrict__ TP,
                                                int* __restrict__ SVs, int* __restrict__ TVs, ghicp_iter* __restrict__ trace,
                                                int* __restrict__ matchlist) {
  if (st->done) return;
  __shared__ double red[16];
  __shared__ int ired[17];
  __shared__ double sh[32];
  __shared__ int s_cor;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int it = st->it;
  const double penalty = st->penalty;
  if (matchlist)
    for (int i = tid; i < C.ks; i += nt) matchlist[(size_t)it * C.ks + i] = -1;

  // ---- correspondences, in the reference's emission order
  int cor = 0;
  if (C.corr == GHICP_CORR_KM) {
    // Km::output (km.cpp:157-171): ascending y, kept iff w[match[y]][y] != -penalty (exact compare)
    for (int base = 0; base < C.n; base += nt) {
      const int y = base + tid;
      int flag = 0, x = -1;
      if (y < C.n) { x = kmmatch[y]; flag = (kmw[(size_t)x * C.n + y] != -penalty) ? 1 : 0; }
      int tot;
      const int pos = gh_block_excl_scan(flag, ired, &tot);
      if (flag) { SP[cor + pos] = x; TP[cor + pos] = y; }
      cor += tot;
      __syncthreads();
    }
    // Calenergy (km.cpp:128-141): INF = 10000 never matches, so energy = -sum of matched weights
    double e = 0;
    for (int y = tid; y < C.n; y += nt) { const double g = kmw[(size_t)kmmatch[y] * C.n + y]; if (g != -10000.0) e -= g; }
    e = gh_block_sum(e, red);
    if (tid == 0) st->energy = e;
  } else {
    // row arg-min over chunks (ascending chunk == ascending column)
    for (int i = tid; i < C.ks; i += nt) {
      double best = 9e20; int bi = 0;
      for (int c = 0; c < C.nchunk_b; c++) {
        const double v = pminA[(size_t)c * C.ks + i];
        if (v < best) { best = v; bi = pidxA[(size_t)c * C.ks + i]; }
      }
      SVs[i] = bi;
      TVs[C.kt + i] = (best < penalty) ? 1 : 0;  // NN acceptance flag (ghicp_reg.cpp:725)
    }
    if (C.corr == GHICP_CORR_NNR) {
      for (int j = tid; j < C.kt; j += nt) {
        double best = 9e20; int bi = 0;
        for (int c = 0; c < C.nchunk_a; c++) {
          const double v = pminB[(size_t)c * C.kt + j];
          if (v < best) { best = v; bi = pidxB[(size_t)c * C.kt + j]; }
        }
        TVs[j] = bi;
      }
    }
    __syncthreads();
    for (int base = 0; base < C.ks; base += nt) {
      const int i = base + tid;
      int flag = 0, sv = 0;
      if (i < C.ks) {
        sv = SVs[i];
        if (C.corr == GHICP_CORR_NN) flag = TVs[C.kt + i];
        else flag = (C.kt > 0 && TVs[sv] == i) ? 1 : 0;  // Q7: reciprocal test only (ghicp_reg.cpp:654)
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
    for (int c = tid; c < cor; c += nt) matchlist[(size_t)it * C.ks + SP[c]] = TP[c];

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

  // ---- float Umeyama (ghicp_reg.cpp:839-866): inputs cast to f32, means/cross-covariance in f64 rounded to f32
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
    s_cor = cor;
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
    trace[it] = rec;
    st->RMS = RMSE; st->FDM = FDM; st->FDstd = FDstd; st->IoU = IoU; st->para1 = p1; st->para2 = p2; st->cor = cor;
    st->it = it + 1;
    if (conv || it + 1 >= C.max_iter) { st->done = 1; st->converged_flag = conv ? 1 : 0; }
  }
}

template <typename T> __global__ void k_transpose(const T* __restrict__ in, int rows, int cols, T* __restrict__ out) {
  __shared__ T tile[32][33];
  const int x = blockIdx.x * 32 + threadIdx.x, y0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y)
    if (x < cols && y0 + r < rows) tile[r][threadIdx.x] = in[(size_t)(y0 + r) * cols + x];
  __syncthreads();
  const int ox = blockIdx.y * 32 + threadIdx.x, oy0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y)
    if (ox < rows && oy0 + r < cols) out[(size_t)(oy0 + r) * rows + ox] = tile[threadIdx.x][r];
}

static void pick_chunks(int ka, int kb, int* chunk, int* nchunk) {
  // aim for >= ~2048 workgroups (256 CUs x 8) with chunks of at least 32 columns
  const int rowblocks = cdiv(ka, ROWS);
  int want = cdiv(2048, rowblocks);
  int ch = cdiv(kb, want);
  if (ch < 32) ch = 32;
  if (ch > CHUNK_MAX) ch = CHUNK_MAX;
  if (kb <= 0) ch = 32;
  *chunk = ch;
  *nchunk = kb > 0 ? cdiv(kb, ch) : 1;
}

template <int FT>
int run_loop(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS_in, int ks, const double* kpT, int kt, const void* FD, double* Rt16,
             ghicp_iter* trace_host, int32_t* n_iter, int32_t* matchlist) {
  hipStream_t s = ctx->stream;
  LoopConst C;
  memset(&C, 0, sizeof(C));
  C.ks = ks; C.kt = kt; C.n = ks > kt ? ks : kt; C.feature = p->feature; C.corr = p->corr; C.max_iter = p->max_iter; C.min_cor = p->min_cor;
  C.scale = (float)(0.005 * p->bbx_magnitude);  // ghicp_reg.h:40 (double product stored to float)
  C.est_iou = p->est_iou; C.adjust_ratio = p->adjust_ratio; C.adjust_step = p->adjust_step;
  C.converge_t = (double)p->converge_t; C.converge_r = (double)p->converge_r; C.penalty_initial = p->penalty_initial; C.km_eps = p->km_eps;
  pick_chunks(ks, kt, &C.chunk_b, &C.nchunk_b);
  pick_chunks(kt, ks, &C.chunk_a, &C.nchunk_a);

  LoopState* st; double* kpS; double *pminA, *pminB, *psum, *wfd; int *pidxA, *pidxB, *SP, *TP, *SVs, *TVs; ghicp_iter* trace;
  GH_TRY(ctx->reserve(B_LOOP_STATE, 1, &st));
  GH_TRY(ctx->reserve(B_LOOP_KPS, (size_t)ks * 3 + 3, &kpS));
  GH_TRY(ctx->reserve(B_LOOP_PARTMIN, (size_t)C.nchunk_b * ks + 1, &pminA));
  GH_TRY(ctx->reserve(B_LOOP_PARTIDX, (size_t)C.nchunk_b * ks + 1, &pidxA));
  GH_TRY(ctx->reserve(B_LOOP_PARTMIN2, (size_t)C.nchunk_a * kt + 1, &pminB));
  GH_TRY(ctx->reserve(B_LOOP_PARTIDX2, (size_t)C.nchunk_a * kt + 1, &pidxB));
  const int nparts = cdiv(ks, ROWS) * C.nchunk_b;
  GH_TRY(ctx->reserve(B_LOOP_PARTSUM, (size_t)nparts * 2 + 2, &psum));
  GH_TRY(ctx->reserve(B_LOOP_SP, (size_t)C.n + 1, &SP));
  GH_TRY(ctx->reserve(B_LOOP_TP, (size_t)C.n + 1, &TP));
  GH_TRY(ctx->reserve(B_LOOP_ACC, (size_t)ks + (size_t)kt + ks + 2, &SVs));
  TVs = SVs + ks;  // TVs[0..kt) column arg-min, TVs[kt..kt+ks) NN acceptance flags
  GH_TRY(ctx->reserve(B_LOOP_TRACE, (size_t)p->max_iter + 1, &trace));
  GH_TRY(ctx->reserve(B_LOOP_WFD, (size_t)p->max_iter + 1, &wfd));

  // exp(-it/rate) from the host libm, exactly as calCD_BSC computes it (ghicp_reg.cpp:247)
  std::vector<double> wtab(p->max_iter + 1);
  for (int i = 0; i <= p->max_iter; i++) wtab[i] = std::exp(-1.0 * i / p->weight_changing_rate);
  GH_HIP(hipMemcpyAsync(wfd, wtab.data(), wtab.size() * sizeof(double), hipMemcpyHostToDevice, s));
  LoopState h;
  memset(&h, 0, sizeof(h));
  h.RMS = 99999; h.para1 = p->para1; h.para2 = p->para2;  // ghicp_reg.h:98, 33-34
  for (int d = 0; d < 4; d++) h.Rt_till[d * 5] = 1.0;
  GH_HIP(hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, s));
  GH_HIP(hipMemcpyAsync(kpS, kpS_in, (size_t)ks * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));

  // FD transposed once so that the row sweep reads it coalesced; the original layout serves the column sweep
  const void* FDt = nullptr;
  if (FT != GHICP_FEATURE_NONE) {
    const size_t esz = (FT == GHICP_FEATURE_BSC) ? 2 : 4;
    char* t;
    GH_TRY(ctx->reserve(B_LOOP_FDT, (size_t)ks * kt * esz + 16, &t));
    dim3 g(cdiv(kt, 32), cdiv(ks, 32)), b(32, 8);
    if (ks > 0 && kt > 0) {
      if (FT == GHICP_FEATURE_BSC) hipLaunchKernelGGL(k_transpose<uint16_t>, g, b, 0, s, (const uint16_t*)FD, ks, kt, (uint16_t*)t);
      else hipLaunchKernelGGL(k_transpose<float>, g, b, 0, s, (const float*)FD, ks, kt, (float*)t);
    }
    FDt = t;
  }
  double* kmw = nullptr; int* kmmatch = nullptr;
  if (p->corr == GHICP_CORR_KM) {
    GH_TRY(ctx->reserve(B_LOOP_KMW, (size_t)C.n * C.n + 1, &kmw));
    GH_TRY(ctx->reserve(B_LOOP_KMMATCH, (size_t)C.n + 1, &kmmatch));
  }

  int* hflag = reinterpret_cast<int*>(ctx->pinned);
  const int poll_every = (p->corr == GHICP_CORR_KM) ? 1 : 4;
  int launched = 0;
  bool done = (ks <= 0 || kt <= 0);
  while (!done && launched < p->max_iter) {
    for (int r = 0; r < poll_every && launched < p->max_iter; r++, launched++) {
      dim3 gA(cdiv(ks, ROWS), C.nchunk_b);
      hipEvent_t kev = ctx->kt_begin(KT_CD_ROWMIN);
      hipLaunchKernelGGL((k_cd_rowmin<FT, true>), gA, dim3(ROWS), 0, s, st, kpS, ks, kpT, kt, FDt, C.chunk_b, C.scale, wfd, pminA, pidxA, psum);
      ctx->kt_end(KT_CD_ROWMIN, kev);
      if (p->corr == GHICP_CORR_NNR) {
        dim3 gB(cdiv(kt, ROWS), C.nchunk_a);
        hipLaunchKernelGGL((k_cd_rowmin<FT, false>), gB, dim3(ROWS), 0, s, st, kpT, kt, kpS, ks, FD, C.chunk_a, C.scale, wfd, pminB, pidxB,
                           (double*)nullptr);
      }
      hipLaunchKernelGGL(k_penalty, dim3(1), dim3(256), 0, s, st, C, psum, nparts, wfd);
      if (p->corr == GHICP_CORR_KM) {
        dim3 gw(cdiv(C.n, 256), C.n);
        hipEvent_t kw = ctx->kt_begin(KT_KM_WEIGHTS);
        hipLaunchKernelGGL(k_km_weights<FT>, gw, dim3(256), 0, s, st, C, kpS, kpT, FD, wfd, kmw);
        ctx->kt_end(KT_KM_WEIGHTS, kw);
        GH_TRY(gh_km_solve_dev(ctx, kmw, C.n, C.km_eps, kmmatch, &st->done));
      }
      hipLaunchKernelGGL(k_solve<FT>, dim3(1), dim3(1024), 0, s, st, C, kpS, kpT, FD, pminA, pidxA, pidxB, pminB, kmw, kmmatch, SP, TP, SVs,
                         TVs, trace, matchlist);
    }
    GH_HIP(hipGetLastError());
    GH_HIP(hipMemcpyAsync(hflag, st, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    GH_HIP(hipStreamSynchronize(s));
    done = hflag[1] != 0;
  }
  LoopState fin;
  if (ks <= 0 || kt <= 0) {
    fin = h;
  } else {
    GH_HIP(hipMemcpyAsync(&fin, st, sizeof(fin), hipMemcpyDeviceToHost, s));
    GH_HIP(hipStreamSynchronize(s));
  }
  for (int d = 0; d < 16; d++) Rt16[d] = fin.Rt_till[d];
  if (n_iter) *n_iter = fin.it;
  if (trace_host && fin.it > 0) {
    GH_HIP(hipMemcpyAsync(trace_host, trace, (size_t)fin.it * sizeof(ghicp_iter), hipMemcpyDeviceToHost, s));
    GH_HIP(hipStreamSynchronize(s));
  }
  return GHICP_OK;
}

}  // namespace

int gh_register_dev(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS, int ks, const double* kpT, int kt, const void* FD,
                    double* Rt16, ghicp_iter* trace, int32_t* n_iter, int32_t* matchlist) {
  GH_ARG(p != nullptr && Rt16 != nullptr);
  GH_ARG(ks >= 0 && kt >= 0 && p->max_iter > 0 && p->max_iter <= 100000);
  GH_ARG(p->corr == GHICP_CORR_NN || p->corr == GHICP_CORR_NNR || p->corr == GHICP_CORR_KM);
  switch (p->feature) {
    case GHICP_FEATURE_BSC:
      GH_ARG(FD != nullptr || ks == 0 || kt == 0);
      return run_loop<GHICP_FEATURE_BSC>(ctx, p, kpS, ks, kpT, kt, FD, Rt16, trace, n_iter, matchlist);
    case GHICP_FEATURE_FPFH:
      GH_ARG(FD != nullptr || ks == 0 || kt == 0);
      return run_loop<GHICP_FEATURE_FPFH>(ctx, p, kpS, ks, kpT, kt, FD, Rt16, trace, n_iter, matchlist);
    case GHICP_FEATURE_NONE:
    case GHICP_FEATURE_ROPS:  // test/ghicp_main.cpp:130-134 falls through with no feature
      return run_loop<GHICP_FEATURE_NONE>(ctx, p, kpS, ks, kpT, kt, nullptr, Rt16, trace, n_iter, matchlist);
    default:
      return ctx->fail(GHICP_ERR_ARG, "unknown feature type %d", p->feature);
  }
}

extern "C" int ghicp_register(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS, int64_t ks, const double* kpT, int64_t kt,
                              const void* FD, double* Rt16, ghicp_iter* trace, int32_t* n_iter, int32_t* matchlist) {
  if (!ctx) return GHICP_ERR_ARG;
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
