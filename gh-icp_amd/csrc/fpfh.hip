// FPFH on gfx950: FPFHfeature::compute_fpfh_feature (reference include/fpfh.hpp:36-58) = pcl::NormalEstimation
// (k = 20, viewpoint at the origin) + pcl::FPFHEstimationOMP (k = 20) over the WHOLE down-sampled cloud, and
// keyfpfh (fpfh.hpp:93-115) = gather of the keypoint rows.  PCL semantics as restated in SURVEY.md §8c:
//   k_knn       exact k nearest neighbours (query included, float L2, ties -> lower index) by ring expansion over
//               the uniform grid; the k best live in registers, the result is sorted by (d2, index)
//   k_normals   covariance of the k neighbours (f64, N2), Jacobi, eigenvector of the smallest eigenvalue, flipped
//               towards the viewpoint
//   k_spfh      Darboux-frame pair features -> 3 x 11 bins, increment 100/(k-1) (computePointSPFHSignature)
//   k_fpfh      sum_{d2 != 0} SPFH(neighbour) / d2, each 11-bin block rescaled to 100 (weightPointSPFHSignature)
// HBM-bound by contract (16 B in + 132 B out per point); the neighbour lists (160 B/point) stay L2-resident.
#include "grid.h"
#include "devmath.h"

namespace {

constexpr int KNN = 20;

__device__ inline bool pair_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

// `kk` <= KNN neighbours are kept (rows of nn/nd stay KNN wide; slots >= kk are never filled)
__device__ inline void knn_point(const GridArgs& G, float cell, int kk, const float4 P, int* __restrict__ nn, float* __restrict__ nd, int* __restrict__ nk) {
  const int self = (int)__float_as_uint(P.w);
  const int cx = gh_cell_coord(P.x, G.d.mn[0], G.d.inv, G.d.dim[0]);
  const int cy = gh_cell_coord(P.y, G.d.mn[1], G.d.inv, G.d.dim[1]);
  const int cz = gh_cell_coord(P.z, G.d.mn[2], G.d.inv, G.d.dim[2]);
  float bd[KNN];
  int bi[KNN];
#pragma unroll
  for (int t = 0; t < KNN; t++) { bd[t] = 3.0e38f; bi[t] = 0x7fffffff; }
  int cnt = 0, wpos = 0;  // wpos = position of the current worst entry
  const int rmax = max(G.d.dim[0], max(G.d.dim[1], G.d.dim[2]));
  for (int r = 0; r <= rmax; r++) {
    const int x0 = max(cx - r, 0), x1 = min(cx + r, G.d.dim[0] - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, G.d.dim[1] - 1);
    const int z0 = max(cz - r, 0), z1 = min(cz + r, G.d.dim[2] - 1);
    auto scan_cell = [&](int x, int y, int z) {
      const unsigned key = ((unsigned)x * G.d.dim[1] + y) * G.d.dim[2] + z;
      const unsigned e = G.start[key + 1];
      for (unsigned q = G.start[key]; q < e; q++) {
        const float4 Q = G.pts[q];
        const int qi = (int)__float_as_uint(Q.w);
        const float dx = P.x - Q.x, dy = P.y - Q.y, dz = P.z - Q.z;
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        if (cnt < kk) {
#pragma unroll
          for (int t = 0; t < KNN; t++)
            if (t == cnt) { bd[t] = d2; bi[t] = qi; }
          cnt++;
          if (cnt == kk) {  // locate the worst
            wpos = 0;
            float wd = bd[0]; int wi = bi[0];
#pragma unroll
            for (int t = 1; t < KNN; t++)
              if (t < kk && pair_less(wd, wi, bd[t], bi[t])) { wpos = t; wd = bd[t]; wi = bi[t]; }
          }
        } else {
          float wd = 0; int wi = 0;
#pragma unroll
          for (int u = 0; u < KNN; u++) if (u == wpos) { wd = bd[u]; wi = bi[u]; }
          if (pair_less(d2, qi, wd, wi)) {
#pragma unroll
            for (int t = 0; t < KNN; t++)
              if (t == wpos) { bd[t] = d2; bi[t] = qi; }
            int np_ = 0; float nwd = bd[0]; int nwi = bi[0];
#pragma unroll
            for (int t = 1; t < KNN; t++)
              if (t < kk && pair_less(nwd, nwi, bd[t], bi[t])) { np_ = t; nwd = bd[t]; nwi = bi[t]; }
            wpos = np_;
          }
        }
      }
    };
    for (int x = x0; x <= x1; x++)
      for (int y = y0; y <= y1; y++) {
        if ((abs(x - cx) == r) || (abs(y - cy) == r)) {  // rim column of shell r: every z of the block
          for (int z = z0; z <= z1; z++) scan_cell(x, y, z);
        } else {  // interior column: only the two caps
          if (cz - r >= 0) scan_cell(x, y, cz - r);
          if (cz + r <= G.d.dim[2] - 1) scan_cell(x, y, cz + r);
        }
      }
    if (cnt == kk) {
      float wd = 0;
#pragma unroll
      for (int u = 0; u < KNN; u++) if (u == wpos) wd = bd[u];
      const float reach = (float)r * cell;  // every unscanned point is at least this far away
      if (wd < reach * reach) break;
    }
  }
  // sort by (d2, index): 20-element insertion network on registers
#pragma unroll
  for (int a = 1; a < KNN; a++) {
#pragma unroll
    for (int b = a; b > 0; b--) {
      if (pair_less(bd[b], bi[b], bd[b - 1], bi[b - 1])) {
        const float td = bd[b]; bd[b] = bd[b - 1]; bd[b - 1] = td;
        const int ti = bi[b]; bi[b] = bi[b - 1]; bi[b - 1] = ti;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KNN; t++) { nn[(size_t)self * KNN + t] = bi[t]; nd[(size_t)self * KNN + t] = bd[t]; }
  nk[self] = cnt;
}

__global__ __launch_bounds__(128) void k_knn(GridArgs G, float cell, int kk, int* __restrict__ nn, float* __restrict__ nd, int* __restrict__ nk) {
  const int p = blockIdx.x * 128 + threadIdx.x;
  if (p >= G.d.n) return;
  knn_point(G, cell, kk, G.pts[p], nn, nd, nk);
}

// the same over the concatenated clouds of a batch (batch.hip): globally numbered cells, one GridDesc per cloud; a point's neighbours are
// indices into the concatenated array and never leave its cloud
__global__ __launch_bounds__(128) void k_knn_batch(const float4* __restrict__ pts, const unsigned* __restrict__ start, const GridDesc* __restrict__ gd,
                                                   const unsigned* __restrict__ cell_base, const int* __restrict__ moff, int nb, int M, int kk,
                                                   int* __restrict__ nn, float* __restrict__ nd, int* __restrict__ nk) {
  const int p = blockIdx.x * 128 + threadIdx.x;
  if (p >= M) return;
  const float4 P = pts[p];
  const int self = (int)__float_as_uint(P.w);
  int lo = 0, hi = nb - 1;  // cloud of the point: largest b with moff[b] <= self
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (moff[mid] <= self) lo = mid;
    else hi = mid - 1;
  }
  GridArgs G;
  G.d = gd[lo]; G.pts = pts; G.start = start + cell_base[lo];
  knn_point(G, 1.0f / G.d.inv, kk, P, nn, nd, nk);
}

__global__ __launch_bounds__(256) void k_normals(const float* __restrict__ xyz, int stride, long long m, const int* __restrict__ nn,
                                                 const int* __restrict__ nk, float* __restrict__ normals, int check_normals) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  const int kk = nk[i];
  if (check_normals && kk < 3) {  // PrincipleComponentAnalysis::CheckNormals (pca.h:258-271): non-finite normal -> 0.577
    normals[i * 3] = 0.577f; normals[i * 3 + 1] = 0.577f; normals[i * 3 + 2] = 0.577f;
    return;
  }
  double c[3] = {0, 0, 0};
  for (int t = 0; t < kk; t++) {
    const long long j = nn[i * KNN + t];
    c[0] += (double)xyz[j * stride]; c[1] += (double)xyz[j * stride + 1]; c[2] += (double)xyz[j * stride + 2];
  }
  c[0] /= kk; c[1] /= kk; c[2] /= kk;
  double S[6] = {0, 0, 0, 0, 0, 0};
  for (int t = 0; t < kk; t++) {
    const long long j = nn[i * KNN + t];
    const double dx = xyz[j * stride] - c[0], dy = xyz[j * stride + 1] - c[1], dz = xyz[j * stride + 2] - c[2];
    S[0] += dx * dx; S[1] += dx * dy; S[2] += dx * dz; S[3] += dy * dy; S[4] += dy * dz; S[5] += dz * dz;
  }
  for (int t = 0; t < 6; t++) S[t] /= kk;
  gh_quant_grid(S, 6);
  double V[9];
  gh_jacobi3(S[0], S[1], S[2], S[3], S[4], S[5], V);
  int im = 0;
  double ev = S[0];
  if (S[3] < ev) { im = 1; ev = S[3]; }
  if (S[5] < ev) { im = 2; }
  float nx = (float)V[0 * 3 + im], ny = (float)V[1 * 3 + im], nz = (float)V[2 * 3 + im];
  const float vx = 0.f - xyz[i * stride], vy = 0.f - xyz[i * stride + 1], vz = 0.f - xyz[i * stride + 2];
  const float cs = (vx * nx + vy * ny) + vz * nz;  // flipNormalTowardsViewpoint, viewpoint (0,0,0)
  if (cs < 0) { nx = -nx; ny = -ny; nz = -nz; }
  normals[i * 3] = nx; normals[i * 3 + 1] = ny; normals[i * 3 + 2] = nz;
}

__device__ inline float dot3f(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

__global__ __launch_bounds__(128) void k_spfh(const float* __restrict__ xyz, int stride, long long m, const int* __restrict__ nn,
                                              const int* __restrict__ nk, const float* __restrict__ normals, float* __restrict__ spfh) {
  const long long i = blockIdx.x * 128ll + threadIdx.x;
  if (i >= m) return;
  float H[33];
#pragma unroll
  for (int b = 0; b < 33; b++) H[b] = 0.f;
  const int kk = nk[i];
  const float incr = 100.0f / (float)(kk - 1);
  const float p1[3] = {xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2]};
  const float n1[3] = {normals[i * 3], normals[i * 3 + 1], normals[i * 3 + 2]};
  for (int t = 0; t < kk; t++) {
    const long long j = nn[i * KNN + t];
    if (j == i) continue;
    const float n2[3] = {normals[j * 3], normals[j * 3 + 1], normals[j * 3 + 2]};
    float dp[3] = {xyz[j * stride] - p1[0], xyz[j * stride + 1] - p1[1], xyz[j * stride + 2] - p1[2]};
    const float f4 = sqrtf(dot3f(dp, dp));
    if (f4 == 0.0f) continue;
    float a1[3] = {n1[0], n1[1], n1[2]}, a2[3] = {n2[0], n2[1], n2[2]};
    const float angle1 = dot3f(a1, dp) / f4, angle2 = dot3f(a2, dp) / f4;
    float f3;
    if (fabsf(angle1) < fabsf(angle2)) {
      for (int d = 0; d < 3; d++) { a1[d] = n2[d]; a2[d] = n1[d]; dp[d] = -dp[d]; }
      f3 = -angle2;
    } else {
      f3 = angle1;
    }
    float vv[3] = {dp[1] * a1[2] - dp[2] * a1[1], dp[2] * a1[0] - dp[0] * a1[2], dp[0] * a1[1] - dp[1] * a1[0]};
    const float vn = sqrtf(dot3f(vv, vv));
    if (vn == 0.0f) continue;
    const float iv = 1.0f / vn;
    for (int d = 0; d < 3; d++) vv[d] *= iv;
    const float ww[3] = {a1[1] * vv[2] - a1[2] * vv[1], a1[2] * vv[0] - a1[0] * vv[2], a1[0] * vv[1] - a1[1] * vv[0]};
    const float f2 = dot3f(vv, a2);
    const float f1 = gh_atan2f(dot3f(ww, a2), dot3f(a1, a2));  // N7: the contract's atan2f, not the device library's
    int h1 = (int)floor(11 * (((double)f1 + M_PI) * (1.0 / (2.0 * M_PI))));
    int h2 = (int)floor(11 * (((double)f2 + 1.0) * 0.5));
    int h3 = (int)floor(11 * (((double)f3 + 1.0) * 0.5));
    h1 = min(max(h1, 0), 10); h2 = min(max(h2, 0), 10); h3 = min(max(h3, 0), 10);
#pragma unroll
    for (int b = 0; b < 11; b++) {
      if (b == h1) H[b] += incr;
      if (b == h2) H[11 + b] += incr;
      if (b == h3) H[22 + b] += incr;
    }
  }
#pragma unroll
  for (int b = 0; b < 33; b++) spfh[i * 33 + b] = H[b];
}

__global__ __launch_bounds__(128) void k_fpfh(long long m, const int* __restrict__ nn, const float* __restrict__ nd, const int* __restrict__ nk,
                                              const float* __restrict__ spfh, float* __restrict__ hist) {
  const long long i = blockIdx.x * 128ll + threadIdx.x;
  if (i >= m) return;
  float F[33];
#pragma unroll
  for (int b = 0; b < 33; b++) F[b] = 0.f;
  float s0 = 0, s1 = 0, s2 = 0;
  const int kk = nk[i];
  for (int t = 0; t < kk; t++) {
    const float d = nd[i * KNN + t];
    if (d == 0) continue;  // the query itself (and exact duplicates)
    const float w = 1.0f / d;
    const float* H = &spfh[(long long)nn[i * KNN + t] * 33];
#pragma unroll
    for (int b = 0; b < 11; b++) {
      const float v0 = H[b] * w, v1 = H[11 + b] * w, v2 = H[22 + b] * w;
      s0 += v0; F[b] += v0;
      s1 += v1; F[11 + b] += v1;
      s2 += v2; F[22 + b] += v2;
    }
  }
  if (s0 != 0) s0 = 100.0f / s0;
  if (s1 != 0) s1 = 100.0f / s1;
  if (s2 != 0) s2 = 100.0f / s2;
#pragma unroll
  for (int b = 0; b < 11; b++) { hist[i * 33 + b] = F[b] * s0; hist[i * 33 + 11 + b] = F[11 + b] * s1; hist[i * 33 + 22 + b] = F[22 + b] * s2; }
}

__global__ __launch_bounds__(256) void k_gather_rows33(const float* __restrict__ hist, const int* __restrict__ idx, long long k, float* __restrict__ out) {
  const long long t = blockIdx.x * 256ll + threadIdx.x;
  if (t >= k * 33) return;
  out[t] = hist[(long long)idx[t / 33] * 33 + t % 33];
}

}  // namespace

float gh_fpfh_cell(const float* mm, long long m);

int gh_fpfh_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float* normals_opt, float* hist) {
  if (m <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  float mm[6];
  GH_TRY(gh_bbox_dev(ctx, xyz, m, stride, mm));
  const float cell = gh_fpfh_cell(mm, m);
  DeviceGrid G;
  const GridSlots sl = {B_GRID2_KEYS, B_GRID2_KEYS2, B_GRID2_VALS, B_GRID2_VALS2, B_GRID2_START, B_GRID2_PTS};
  GH_TRY(gh_grid_build(ctx, xyz, m, stride, cell, sl, &G));
  const float cell_eff = 1.0f / G.d.inv;  // the grid may have been coarsened
  int *nn, *nk;
  float *nd, *spfh, *nrm;
  GH_TRY(ctx->reserve(B_FE_SORTK, (size_t)m * KNN + 1, &nn));
  GH_TRY(ctx->reserve(B_FE_SORTK2, (size_t)m * KNN + 1, &nd));
  GH_TRY(ctx->reserve(B_FE_COUNT, (size_t)m + 1, &nk));
  GH_TRY(ctx->reserve(B_FE_CPTS, (size_t)m * 33 + 1, &spfh));
  GH_TRY(ctx->reserve(B_FE_LAMBDA, (size_t)m * 3 + 3, &nrm));
  if (normals_opt) nrm = normals_opt;
  GridArgs A = {G.d, G.pts, G.start};
  hipLaunchKernelGGL(k_knn, dim3(cdiv(m, 128)), dim3(128), 0, s, A, cell_eff, KNN, nn, nd, nk);
  hipLaunchKernelGGL(k_normals, dim3(cdiv(m, 256)), dim3(256), 0, s, xyz, stride, m, nn, nk, nrm, 0);
  hipLaunchKernelGGL(k_spfh, dim3(cdiv(m, 128)), dim3(128), 0, s, xyz, stride, m, nn, nk, nrm, spfh);
  hipLaunchKernelGGL(k_fpfh, dim3(cdiv(m, 128)), dim3(128), 0, s, m, nn, nd, nk, spfh, hist);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

// FPFH of every point of a batch's concatenated down-sampled clouds (float4 rows, M points) over the batch's kNN grid (batch.hip builds it:
// per-cloud cell size as in gh_fpfh_dev, globally numbered cells).  The normal / SPFH / FPFH kernels are the single-cloud ones: neighbour
// indices are indices into the concatenated array.
float gh_fpfh_cell(const float* mm, long long m) {  // cell size of the kNN grid of one cloud (gh_fpfh_dev)
  const double vol = fmax(1e-9, (double)(mm[3] - mm[0] + 1e-3) * (mm[4] - mm[1] + 1e-3) * (mm[5] - mm[2] + 1e-3));
  float cell = (float)cbrt(vol / (double)(m > 0 ? m : 1) * 8.0);
  return cell < 0.05f ? 0.05f : cell;
}
int gh_fpfh_batch_dev(ghicp_ctx* ctx, const float4* dsg, int M, const float4* pts, const unsigned* start, const GridDesc* gd_dev, const unsigned* cell_base_dev,
                      const int* moff_dev, int nb, float* hist) {
  if (M <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  int *nn, *nk;
  float *nd, *spfh, *nrm;
  GH_TRY(ctx->reserve(B_FE_SORTK, (size_t)M * KNN + 1, &nn));
  GH_TRY(ctx->reserve(B_FE_SORTK2, (size_t)M * KNN + 1, &nd));
  GH_TRY(ctx->reserve(B_FE_COUNT, (size_t)M + 1, &nk));
  GH_TRY(ctx->reserve(B_FE_CPTS, (size_t)M * 33 + 1, &spfh));
  GH_TRY(ctx->reserve(B_FE_LAMBDA, (size_t)M * 3 + 3, &nrm));
  const float* xyz = reinterpret_cast<const float*>(dsg);
  hipLaunchKernelGGL(k_knn_batch, dim3(cdiv(M, 128)), dim3(128), 0, s, pts, start, gd_dev, cell_base_dev, moff_dev, nb, M, KNN, nn, nd, nk);
  hipLaunchKernelGGL(k_normals, dim3(cdiv(M, 256)), dim3(256), 0, s, xyz, 4, (long long)M, nn, nk, nrm, 0);
  hipLaunchKernelGGL(k_spfh, dim3(cdiv(M, 128)), dim3(128), 0, s, xyz, 4, (long long)M, nn, nk, nrm, spfh);
  hipLaunchKernelGGL(k_fpfh, dim3(cdiv(M, 128)), dim3(128), 0, s, (long long)M, nn, nd, nk, spfh, hist);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

// PrincipleComponentAnalysis::CalculateNormalVector_KNN (include/pca.h:92-109): pcl::NormalEstimation with setKSearch(k)
// followed by CheckNormals.  Shares the exact-kNN and covariance kernels with the FPFH path; k <= 20.
int gh_knn_normals_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, int k, float* normals) {
  if (m <= 0) return GHICP_OK;
  if (k < 1 || k > KNN) return ctx->fail(GHICP_ERR_ARG, "knn normals: k must be in [1, %d]", KNN);
  hipStream_t s = ctx->stream;
  float mm[6];
  GH_TRY(gh_bbox_dev(ctx, xyz, m, stride, mm));
  const double vol = fmax(1e-9, (double)(mm[3] - mm[0] + 1e-3) * (mm[4] - mm[1] + 1e-3) * (mm[5] - mm[2] + 1e-3));
  float cell = (float)cbrt(vol / (double)m * 8.0);
  if (cell < 0.05f) cell = 0.05f;
  DeviceGrid G;
  const GridSlots sl = {B_GRID2_KEYS, B_GRID2_KEYS2, B_GRID2_VALS, B_GRID2_VALS2, B_GRID2_START, B_GRID2_PTS};
  GH_TRY(gh_grid_build(ctx, xyz, m, stride, cell, sl, &G));
  int *nn, *nk;
  float* nd;
  GH_TRY(ctx->reserve(B_FE_SORTK, (size_t)m * KNN + 1, &nn));
  GH_TRY(ctx->reserve(B_FE_SORTK2, (size_t)m * KNN + 1, &nd));
  GH_TRY(ctx->reserve(B_FE_COUNT, (size_t)m + 1, &nk));
  GridArgs A = {G.d, G.pts, G.start};
  hipLaunchKernelGGL(k_knn, dim3(cdiv(m, 128)), dim3(128), 0, s, A, 1.0f / G.d.inv, k, nn, nd, nk);
  hipLaunchKernelGGL(k_normals, dim3(cdiv(m, 256)), dim3(256), 0, s, xyz, stride, m, nn, nk, normals, 1);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

int gh_gather_rows33_dev(ghicp_ctx* ctx, const float* hist, const int32_t* idx, long long k, float* out) {
  if (k <= 0) return GHICP_OK;
  hipLaunchKernelGGL(k_gather_rows33, dim3(cdiv(k * 33, 256)), dim3(256), 0, ctx->stream, hist, idx, k, out);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

extern "C" int ghicp_fpfh(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, int k_normal, int k_feature, float* normals, float* hist) {
  GH_ENTER(ctx);
  GH_ARG(m >= 0 && m < (1ll << 31) - 2 && stride >= 3 && hist != nullptr);
  if (k_normal != KNN || k_feature != KNN) return ctx->fail(GHICP_ERR_ARG, "ghicp_fpfh: k must be 20/20 as in fpfh.hpp:43,52");
  Stager sg(ctx);
  const float* d;
  float *dn, *dh;
  GH_TRY(sg.in_cloud(xyz, (size_t)m * stride, &d));
  GH_TRY(sg.out(normals, (size_t)m * 3, &dn));
  GH_TRY(sg.out(hist, (size_t)m * 33, &dh));
  GH_TRY(gh_fpfh_dev(ctx, d, m, stride, dn, dh));
  return sg.finish();
}

extern "C" int ghicp_fpfh_keypoints(ghicp_ctx* ctx, const float* hist, const int32_t* kp_idx, int64_t k, float* out) {
  GH_ENTER(ctx);
  GH_ARG(k >= 0);
  if (ctx->host_ptrs) return ctx->fail(GHICP_ERR_ARG, "ghicp_fpfh_keypoints: device-pointer mode only");
  return gh_gather_rows33_dev(ctx, hist, kp_idx, k, out);
}
