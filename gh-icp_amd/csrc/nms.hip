// Keypoint front end, part 2 (gfx950): CKeypointDetect::nonMaximaSuppression
// (reference include/keypoint_detect.hpp:149-191) and the PCA -> prune -> NMS driver
// keypointDetectionBasedOnCurvature (keypoint_detect.hpp:27-51).
//
// The reference is a sequential greedy sweep: sort candidates by curvature (descending), repeatedly
// emit the best unsuppressed one and erase everything within R of it.  Greedy NMS over a strict total
// order is the unique fixed point of
//     selected(i)  <=>  no selected j with rank(j) < rank(i) and d2(i,j) < R^2          
// and is computed by ONE workgroup per cloud that sweeps the rank-ordered candidates in chunks (nms_dev.h).
// Rank = (curvature desc, candidate order asc): stable radix sort, so ties resolve to the lower point index.
#include "grid.h"

#include <algorithm>

#include "prims.h"

#include <cmath>
#include <cstdlib>
#include "devmath.h"
#include "nms_dev.h"

namespace {

// order-preserving map f64 -> u64 (ascending); NaN never reaches here (prune rejects it)
__device__ inline unsigned long long f64_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void k_nms_keys(const double* __restrict__ curvature, const int* __restrict__ cand, int c,
                                                  unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c) return;
  keys[i] = ~f64_key(curvature[cand[i]]);
  vals[i] = i;
}

// cpts[r] = xyz of the rank-r candidate (packed float3 rows for gh_grid_build)
__global__ __launch_bounds__(256) void k_nms_points(const float* __restrict__ xyz, int stride, const int* __restrict__ cand,
                                                    const int* __restrict__ ord, int c, float* __restrict__ cpts) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= c) return;
  const long long s = cand[ord[r]];
  cpts[r * 3] = xyz[s * stride];
  cpts[r * 3 + 1] = xyz[s * stride + 1];
  cpts[r * 3 + 2] = xyz[s * stride + 2];
}

__global__ __launch_bounds__(NMS_T) void k_nms_greedy(const float* __restrict__ cpts, int c, GridDesc g, float r2, int* __restrict__ head,
                                                      int* __restrict__ next, const int* __restrict__ cand, const int* __restrict__ ord,
                                                      int* __restrict__ kp, int* __restrict__ kcount) {
  gh_nms_greedy_cloud(cpts, c, g, r2, head, next, cand, ord, kp, kcount, 0);
}

}  // namespace

int gh_pca_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float radius, float* lambda, double* curvature, int32_t* count);
int gh_prune_dev(ghicp_ctx* ctx, const float* lambda, const int32_t* count, long long m, float ratio_max, int min_n, int32_t* cand, long long* c_out);

int gh_nms_dev(ghicp_ctx* ctx, const float* xyz, int stride, const double* curvature, const int32_t* cand, long long c, float radius, int32_t* kp,
               long long* k_out) {
  *k_out = 0;
  if (c <= 0) return GHICP_OK;  // NB the reference's do-while dereferences an empty set here (keypoint_detect.hpp:177)
  hipStream_t s = ctx->stream;
  unsigned long long *keys, *keys2;
  int *vals, *ord, *misc;
  float* cpts;
  GH_TRY(ctx->reserve(B_FE_SORTK, (size_t)c + 1, &keys));
  GH_TRY(ctx->reserve(B_FE_SORTK2, (size_t)c + 1, &keys2));
  GH_TRY(ctx->reserve(B_FE_SORTV, (size_t)c + 1, &vals));
  GH_TRY(ctx->reserve(B_FE_SORTV2, (size_t)c + 1, &ord));
  GH_TRY(ctx->reserve(B_FE_CPTS, (size_t)c * 3 + 3, &cpts));
  GH_TRY(ctx->reserve(B_FE_SCAN, 16, &misc));
  hipLaunchKernelGGL(k_nms_keys, dim3(cdiv(c, 256)), dim3(256), 0, s, curvature, cand, (int)c, keys, vals);
  // descending curvature, ties in candidate order: a stable ascending sort of the complemented keys (prims.hip)
  GH_TRY(gh_radix_sort_u64(ctx, keys, keys2, reinterpret_cast<const unsigned*>(vals), reinterpret_cast<unsigned*>(ord), c, 0, 64));
  hipLaunchKernelGGL(k_nms_points, dim3(cdiv(c, 256)), dim3(256), 0, s, xyz, stride, cand, ord, (int)c, cpts);
  const float r2 = (float)((double)radius * (double)radius);
  int* hflag = reinterpret_cast<int*>(ctx->pinned);
  // ---- single-launch exact greedy NMS over a grid of SELECTED keypoints
  float mm[6];
  GH_TRY(gh_bbox_dev(ctx, cpts, c, 3, mm));
  const GridDesc g = gh_grid_desc(mm, c, radius * 1.0001f);
  int *head, *next;
  GH_TRY(ctx->reserve(B_GRID_START, (size_t)g.ncell + 2, &head));
  GH_TRY(ctx->reserve(B_FE_STATE, (size_t)c + 1, &next));
  GH_HIP(hipMemsetAsync(head, 0xff, (size_t)g.ncell * sizeof(int), s));
  hipEvent_t kev = ctx->kt_begin(KT_NMS_ROUND);
  hipLaunchKernelGGL(k_nms_greedy, dim3(1), dim3(NMS_T), 0, s, cpts, (int)c, g, r2, head, next, cand, ord, kp, misc);
  ctx->kt_end(KT_NMS_ROUND, kev);
  GH_HIP(hipGetLastError());
  GH_HIP(hipMemcpyAsync(hflag, misc, sizeof(int), hipMemcpyDeviceToHost, s));
  GH_HIP(hipStreamSynchronize(s));
  *k_out = hflag[0];
  return GHICP_OK;
}

int gh_keypoints_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float radius, float ratio_max, int min_n, float nms_radius,
                     int32_t* kp, long long* k_out) {
  *k_out = 0;
  if (m <= 0) return GHICP_OK;
  float* lambda;
  double* curv;
  int *count, *cand;
  GH_TRY(ctx->reserve(B_FE_LAMBDA, (size_t)m * 3 + 3, &lambda));
  GH_TRY(ctx->reserve(B_FE_CURV, (size_t)m + 1, &curv));
  GH_TRY(ctx->reserve(B_FE_COUNT, (size_t)m + 1, &count));
  GH_TRY(ctx->reserve(B_FE_CAND, (size_t)m + 1, &cand));
  GH_TRY(gh_pca_dev(ctx, xyz, m, stride, radius, lambda, curv, count));
  long long c = 0;
  GH_TRY(gh_prune_dev(ctx, lambda, count, m, ratio_max, min_n, cand, &c));
  return gh_nms_dev(ctx, xyz, stride, curv, cand, c, nms_radius, kp, k_out);
}

// CKeypointDetect::keypointDetectionBasedOnCurvature_adaptive (include/keypoint_detect.hpp:53-111): the PCA features are
// computed once; prune + NMS are repeated with ratioMax -= 0.05 while more than `upper` keypoints come out (one step of
// +0.025 and stop once fewer than `lower` do), never below 0.65.  float/double mixing as in the reference.
int gh_keypoints_adaptive_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float radius, float ratio_max, int min_n, float nms_radius,
                              long long upper, long long lower, int32_t* kp, long long* k_out, float* ratio_used, int* rounds) {
  *k_out = 0;
  if (ratio_used) *ratio_used = ratio_max;
  if (rounds) *rounds = 0;
  if (m <= 0) return GHICP_OK;
  float* lambda;
  double* curv;
  int *count, *cand;
  GH_TRY(ctx->reserve(B_FE_LAMBDA, (size_t)m * 3 + 3, &lambda));
  GH_TRY(ctx->reserve(B_FE_CURV, (size_t)m + 1, &curv));
  GH_TRY(ctx->reserve(B_FE_COUNT, (size_t)m + 1, &count));
  GH_TRY(ctx->reserve(B_FE_CAND, (size_t)m + 1, &cand));
  GH_TRY(gh_pca_dev(ctx, xyz, m, stride, radius, lambda, curv, count));
  long long c = 0, K = 0;
  GH_TRY(gh_prune_dev(ctx, lambda, count, m, ratio_max, min_n, cand, &c));
  GH_TRY(gh_nms_dev(ctx, xyz, stride, curv, cand, c, nms_radius, kp, &K));
  bool finish = false;
  float ratioMax = ratio_max;
  int nr = 0;
  if (K > upper) {
    do {
      if (K < lower) { ratioMax += 0.025; finish = true; }
      else ratioMax -= 0.05;
      GH_TRY(gh_prune_dev(ctx, lambda, count, m, ratioMax, min_n, cand, &c));
      GH_TRY(gh_nms_dev(ctx, xyz, stride, curv, cand, c, nms_radius, kp, &K));
      nr++;
    } while ((K < lower || K > upper) && !finish && ratioMax >= 0.65);
  }
  *k_out = K;
  if (ratio_used) *ratio_used = ratioMax;
  if (rounds) *rounds = nr;
  return GHICP_OK;
}

extern "C" int ghicp_keypoints_adaptive(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, float radius, float ratio_max, int min_n,
                                        float nms_radius, int64_t upper, int64_t lower, int32_t* kp_idx, int64_t* k, float* ratio_used,
                                        int32_t* rounds) {
  GH_ENTER(ctx);
  GH_ARG(m >= 0 && m < (1ll << 31) - 2 && stride >= 3 && radius > 0.f && nms_radius > 0.f && k != nullptr && upper >= 0 && lower >= 0);
  Stager sg(ctx);
  const float* d;
  int32_t* dk;
  GH_TRY(sg.in_cloud(xyz, (size_t)m * stride, &d));
  GH_TRY(sg.out(kp_idx, (size_t)m, &dk));
  long long kk = 0;
  int nr = 0;
  GH_TRY(gh_keypoints_adaptive_dev(ctx, d, m, stride, radius, ratio_max, min_n, nms_radius, upper, lower, dk, &kk, ratio_used, &nr));
  *k = kk;
  if (rounds) *rounds = nr;
  return sg.finish();
}

extern "C" int ghicp_nms(ghicp_ctx* ctx, const float* xyz, int stride, const double* curvature, const int32_t* cand, int64_t c, float radius,
                         int32_t* kp, int64_t* k) {
  GH_ENTER(ctx);
  GH_ARG(c >= 0 && c < (1ll << 31) - 2 && stride >= 3 && radius > 0.f && k != nullptr);
  if (ctx->host_ptrs) return ctx->fail(GHICP_ERR_ARG, "ghicp_nms: device-pointer mode only (use ghicp_keypoints from host memory)");
  long long kk = 0;
  GH_TRY(gh_nms_dev(ctx, xyz, stride, curvature, cand, c, radius, kp, &kk));
  *k = kk;
  return GHICP_OK;
}

extern "C" int ghicp_keypoints(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, float radius, float ratio_max, int min_n,
                               float nms_radius, int32_t* kp_idx, int64_t* k) {
  GH_ENTER(ctx);
  GH_ARG(m >= 0 && m < (1ll << 31) - 2 && stride >= 3 && radius > 0.f && nms_radius > 0.f && k != nullptr);
  Stager sg(ctx);
  const float* d;
  int32_t* dk;
  GH_TRY(sg.in_cloud(xyz, (size_t)m * stride, &d));
  GH_TRY(sg.out(kp_idx, (size_t)m, &dk));
  long long kk = 0;
  GH_TRY(gh_keypoints_dev(ctx, d, m, stride, radius, ratio_max, min_n, nms_radius, dk, &kk));
  *k = kk;
  return sg.finish();
}
