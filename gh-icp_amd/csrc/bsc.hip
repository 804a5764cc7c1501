// Binary Shape Context encoder on gfx950: BSCEncoder::extractBinaryFeatures and everything below it
// (reference include/binary_feature_extraction.hpp:603-676, 762-837, 121-193, 196-373, 464-565,
// 678-758, 940-1035).  One 256-thread workgroup per keypoint:
//   sweep A  neighbours within sqrt(3)*R (9 contiguous runs of the cell-sorted cloud, coalesced
//            float4 loads): count, centroid, sum of (sqrt2*R - d)                          (bfe:956-968)
//   sweep B  weighted scatter about the centroid -> f32 covariance -> Jacobi -> LCS axes    (bfe:970-1030, 121-155)
//   sweep C  rotate into the LCS and splat Gaussian weights into the 3 x 7 x 7 projection cells with
//            LDS f64 atomics (cell sums of f32 terms are exact in f64, hence order independent)  (bfe:196-331)
//   tail     cell depth / normalised weight, per-plane pair statistics, 441-bit string + the flip
//            variants with their "294-cell" quirk                                            (bfe:333-372, 464-565, 678-837)
// Wavefront shuffle reductions carry the 4 + 6 f64 sums of sweeps A/B.
#include "grid.h"
#include "devmath.h"
#include "bsc_dev.h"

namespace {

__global__ __launch_bounds__(BT) void k_bsc(GridArgs G, const int* __restrict__ kp, BscConst C, uint8_t* __restrict__ feat, float* __restrict__ lcs) {
  gh_bsc_keypoint(G, C, (int)blockIdx.x, C.K, feat, lcs);
}

// lcs[k][9..11] = keypoint coordinates (the LCS origin, bfe:146-148)
__global__ __launch_bounds__(256) void k_bsc_origins(const float* __restrict__ xyz, int stride, const int* __restrict__ kp, int K, float* __restrict__ lcs) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const long long s = kp[k];
  lcs[(size_t)k * 12 + 9] = xyz[s * stride];
  lcs[(size_t)k * 12 + 10] = xyz[s * stride + 1];
  lcs[(size_t)k * 12 + 11] = xyz[s * stride + 2];
}

}  // namespace

// the constants of BSCEncoder(R, 7) (bfe:63-73) and of one extractBinaryFeatures call; r_search = sqrt(3) R as float (bfe:641)
int gh_bsc_make_const(ghicp_ctx* ctx, float R, int dof, const int32_t* pattern_host, BscConst* out, float* r_search) {
  BscConst C;
  memset(&C, 0, sizeof(C));
  C.R = R;
  const double rs = std::sqrt(3.0) * (double)R;  // bfe:641
  C.r2s = (float)(rs * rs);
  C.u = 2 * R / 7;                                      // bfe:71
  const float delta = (float)(C.u * 0.5);               // bfe:204
  C.den = 2 * delta * delta;                            // bfe:239
  C.r2c = (float)((1.5 * (double)C.u) * (1.5 * (double)C.u));
  C.area = (float)(M_PI * (double)R * (double)R);       // bfe:337
  C.radius_w = std::sqrt(2.0) * (double)R;              // bfe:951
  int ex = 0;
  (void)std::frexp((double)R * 0.5, &ex);               // R / 2 = m 2^ex, m in [0.5, 1): ulp_f32(R / 2) = 2^(ex - 24)
  C.dunit = std::ldexp(1.0, 24 - ex);                   // depth is a multiple of 2^(ex - 24) (bsc_dev.h: exact depth sums)
  C.dinv = std::ldexp(1.0, ex - 54);                    // x weight multiples of 2^-30
  for (int i = 0; i < 7; i++) C.centre[i] = (float)((i + 0.5) * (double)C.u - (double)R);  // bfe:226-227
  for (int i = 0; i < 98; i++) {
    C.pattern[i] = pattern_host[i];
    if (C.pattern[i] < 0 || C.pattern[i] > 48) return ctx->fail(GHICP_ERR_ARG, "BSC sample pattern entry %d out of [0,48]", i);
  }
  C.K = 0;
  C.nvar = (dof > 4) ? 4 : (dof > 0 ? 2 : 1);  // bfe:648-660
  *out = C;
  *r_search = (float)rs;
  return GHICP_OK;
}

int gh_bsc_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, const int32_t* kp, long long K, float R, int dof, const int32_t* pattern_host,
               uint8_t* feat, float* lcs) {
  if (K <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  BscConst C;
  float r_search_f = 0.f;
  GH_TRY(gh_bsc_make_const(ctx, R, dof, pattern_host, &C, &r_search_f));
  C.K = (int)K;
  DeviceGrid G;
  const GridSlots sl = {B_GRID2_KEYS, B_GRID2_KEYS2, B_GRID2_VALS, B_GRID2_VALS2, B_GRID2_START, B_GRID2_PTS};
  GH_TRY(gh_grid_build(ctx, xyz, m, stride, r_search_f * 1.0001f, sl, &G));
  GH_HIP(hipMemsetAsync(feat, 0, (size_t)4 * K * 56, s));
  hipLaunchKernelGGL(k_bsc_origins, dim3(cdiv(K, 256)), dim3(256), 0, s, xyz, stride, kp, (int)K, lcs);
  GridArgs A = {G.d, G.pts, G.start};
  hipEvent_t kt = ctx->kt_begin(KT_BSC);
  hipLaunchKernelGGL(k_bsc, dim3((unsigned)K), dim3(BT), 0, s, A, kp, C, feat, lcs);
  ctx->kt_end(KT_BSC, kt);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

extern "C" int ghicp_bsc_encode(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, const int32_t* kp_idx, int64_t k, float radius, int dof,
                                const int32_t* pattern, uint8_t* feat, float* lcs) {
  GH_ENTER(ctx);
  GH_ARG(m >= 0 && m < (1ll << 31) - 2 && k >= 0 && k <= m && stride >= 3 && radius > 0.f && pattern != nullptr);
  Stager sg(ctx);
  const float* d;
  const int32_t* dk;
  uint8_t* df;
  float* dl;
  GH_TRY(sg.in_cloud(xyz, (size_t)m * stride, &d));
  GH_TRY(sg.in(kp_idx, (size_t)k, &dk));
  GH_TRY(sg.out(feat, (size_t)4 * k * 56, &df));
  GH_TRY(sg.out(lcs, (size_t)k * 12, &dl));
  GH_TRY(gh_bsc_dev(ctx, d, m, stride, dk, k, radius, dof, pattern, df, dl));
  return sg.finish();
}
