// Keypoint front end, part 1 (gfx950):
//   k_pca_cells : PrincipleComponentAnalysis::CalculatePcaFeaturesOfPointCloud + CalculatePcaFeature
//                 (reference include/pca.h:133-165, 202-250).  One wave per occupied grid cell: the
//                 lanes are the cell's query points, the 27-cell neighbourhood (9 contiguous runs of
//                 the cell-sorted float4 array) is staged through LDS in coalesced 16-byte loads and
//                 every lane sweeps the tile with broadcast ds_read_b128.  Two sweeps (centroid, then
//                 de-meaned scatter) in f64, scatter rounded once to f32 (pcl::PCA is a Matrix3f),
//                 eigenvalues by cyclic Jacobi (numerics contract N1-N3).
//   prune       : CKeypointDetect::pruneUnstablePoints (include/keypoint_detect.hpp:132-147).
// Roofline: HBM-bound by contract (16 B in + 24 B out per point); the 27-cell gather re-reads are
// served by LDS/L2 (streamed traffic 16*M*27-cell amplification).
#include "grid.h"
#include "devmath.h"
#include "pca_dev.h"

#include "prims.h"

namespace {

template <int CHUNK>
__global__ __launch_bounds__(64) void k_pca_cells(GridArgs G, const unsigned* __restrict__ cells, const int* __restrict__ ncells,
                                                   int* __restrict__ counter, float r2, double* __restrict__ scat, int* __restrict__ count) {
  __shared__ float4 sC[CHUNK];
  const int lane = threadIdx.x;
  const int nc = *ncells;
  // cells are dealt out statically (block b takes cells b, b + grid, ...): the round-2 version popped ONE cell per atomicAdd on a single
  // global counter, and ~450 k pops per batch of 32 clouds, serialised in L2, were the kernel's whole run time (5.9 ms whatever the
  // arithmetic inside cost: profiles/r03_kernel_stats_fe_one_stream*.txt)
  gh_pca_for_my_runs(nc, counter, [&](int c0, int cnt) {  // (round 5: runs of consecutive cells per workgroup, neighbourhoods per XCD -- pca_dev.h)
    const unsigned key_l = lane < cnt ? cells[c0 + lane] : 0u;
    PcaMeta mn = gh_pca_meta(G, (unsigned)__builtin_amdgcn_readlane((int)key_l, 0), lane);
    for (int j = 0; j < cnt; j++) {
      const PcaMeta m = mn;
      const unsigned key = (unsigned)__builtin_amdgcn_readlane((int)key_l, j);
      if (j + 1 < cnt) mn = gh_pca_meta(G, (unsigned)__builtin_amdgcn_readlane((int)key_l, j + 1), lane);  // the next cell's lookups fly during this cell's pass
      __syncthreads();
      gh_pca_cell_body<CHUNK>(G, key, m, r2, scat, count, sC, lane);
    }
  });
}

__global__ __launch_bounds__(256) void k_pca_eigen(const double* __restrict__ scat, const int* __restrict__ count, long long m, float* __restrict__ lambda,
                                                   double* __restrict__ curvature) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i < m) gh_pca_eigen_point(scat, count, i, lambda, curvature);
}

// keypoint_detect.hpp:132-147: float ratios of the (f32-valued) double eigenvalues; NaN fails
__global__ __launch_bounds__(256) void k_prune_flags(const float* __restrict__ lambda, const int* __restrict__ count, long long m, float ratio_max,
                                                     int min_n, unsigned char* __restrict__ flags) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  const double l1 = (double)lambda[i * 3], l2 = (double)lambda[i * 3 + 1], l3 = (double)lambda[i * 3 + 2];
  const float r1 = (float)(l2 / l1), r2 = (float)(l3 / l2);
  flags[i] = (r1 < ratio_max && r2 < ratio_max && count[i] > min_n) ? 1 : 0;
}

}  // namespace

int gh_pca_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float radius, float* lambda, double* curvature, int32_t* count) {
  if (m <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  DeviceGrid G;
  const GridSlots sl = {B_GRID_KEYS, B_GRID_KEYS2, B_GRID_VALS, B_GRID_VALS2, B_GRID_START, B_GRID_PTS};
  GH_TRY(gh_grid_build(ctx, xyz, m, stride, radius * 1.0001f, sl, &G));
  // occupied cells = unique sorted keys
  unsigned* cells;
  int* misc;
  GH_TRY(ctx->reserve(B_FE_SORTK, (size_t)m + 1, &cells));
  GH_TRY(ctx->reserve(B_FE_SCAN, 16, &misc));
  GH_TRY(gh_unique_sorted_u32(ctx, G.keys, m, cells, misc));  // prims.hip
  GH_HIP(hipMemsetAsync(misc + 4, 0, 8 * sizeof(int), s));
  GridArgs A = {G.d, G.pts, G.start};
  const float r2 = (float)((double)radius * (double)radius);  // pcl radiusSearch: static_cast<float>(radius*radius)
  const int blocks = ctx->num_cu * 20;
  hipEvent_t kt = ctx->kt_begin(KT_PCA);
  double* scat;
  GH_TRY(ctx->reserve(B_FE_SCATTER, (size_t)m * 6 + 6, &scat));
  hipLaunchKernelGGL(k_pca_cells<PCA_CHUNK>, dim3(blocks), dim3(64), 0, s, A, cells, misc, misc + 4, r2, scat, count);
  hipLaunchKernelGGL(k_pca_eigen, dim3(cdiv(m, 256)), dim3(256), 0, s, (const double*)scat, (const int*)count, m, lambda, curvature);
  ctx->kt_end(KT_PCA, kt);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

int gh_prune_dev(ghicp_ctx* ctx, const float* lambda, const int32_t* count, long long m, float ratio_max, int min_n, int32_t* cand, long long* c_out) {
  *c_out = 0;
  if (m <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  unsigned char* flags;
  int* dcount;
  GH_TRY(ctx->reserve(B_FE_FLAGS, (size_t)m + 16, &flags));
  GH_TRY(ctx->reserve(B_FE_SCAN, 16, &dcount));
  hipLaunchKernelGGL(k_prune_flags, dim3(cdiv(m, 256)), dim3(256), 0, s, lambda, count, m, ratio_max, min_n, flags);
  GH_TRY(gh_select_flagged_iota(ctx, flags, m, cand, dcount));  // prims.hip
  int* hc = reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->pinned) + 320);  // pinned: see gh_bbox_dev
  GH_HIP(hipMemcpyAsync(hc, dcount, sizeof(int), hipMemcpyDeviceToHost, s));
  GH_HIP(hipStreamSynchronize(s));
  *c_out = *hc;
  return GHICP_OK;
}

extern "C" int ghicp_pca_curvature(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, float radius, float* lambda, double* curvature,
                                   int32_t* count) {
  GH_ENTER(ctx);
  GH_ARG(m >= 0 && m < (1ll << 31) - 2 && stride >= 3 && radius > 0.f);
  Stager sg(ctx);
  const float* d;
  float* dl;
  double* dc;
  int32_t* dn;
  GH_TRY(sg.in_cloud(xyz, (size_t)m * stride, &d));
  GH_TRY(sg.out(lambda, (size_t)m * 3, &dl));
  GH_TRY(sg.out(curvature, (size_t)m, &dc));
  GH_TRY(sg.out(count, (size_t)m, &dn));
  GH_TRY(gh_pca_dev(ctx, d, m, stride, radius, dl, dc, dn));
  return sg.finish();
}

extern "C" int ghicp_prune(ghicp_ctx* ctx, const float* lambda, const int32_t* count, int64_t m, float ratio_max, int min_n, int32_t* cand,
                           int64_t* c) {
  GH_ENTER(ctx);
  GH_ARG(m >= 0 && m < (1ll << 31) - 2 && c != nullptr);
  Stager sg(ctx);
  const float* dl;
  const int32_t* dn;
  int32_t* dc;
  GH_TRY(sg.in(lambda, (size_t)m * 3, &dl));
  GH_TRY(sg.in(count, (size_t)m, &dn));
  GH_TRY(sg.out(cand, (size_t)m, &dc));
  long long cc = 0;
  GH_TRY(gh_prune_dev(ctx, dl, dn, m, ratio_max, min_n, dc, &cc));
  *c = cc;
  return sg.finish();
}
