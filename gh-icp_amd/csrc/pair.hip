// Whole-pair pipeline on device: the call order of the reference's only caller, main()
// (test/ghicp_main.cpp:86-153): voxel filter x2 -> bbx of the down-sampled source -> curvature keypoints x2
// -> keypoint coordinates as f64 (DataIo::savecoordinates, dataio.hpp:609-627) -> BSC (target dof 0,
// source reg_dof) -> feature distance -> GHRegistration::ghicp_reg.  Every stage is the public ABI
// function of the same name; nothing leaves HBM between stages except element counts.
#include "ctx.h"

int gh_voxel_filter_dev(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float voxel, int32_t* keep, long long* m_out);
int gh_keypoints_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float radius, float ratio_max, int min_n, float nms_radius,
                     int32_t* kp, long long* k_out);
int gh_bsc_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, const int32_t* kp, long long K, float R, int dof, const int32_t* pattern_host,
               uint8_t* feat, float* lcs);
int gh_bbox_dev(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float* mm_host6);

namespace {

__global__ __launch_bounds__(256) void k_gather4p(const float* __restrict__ xyz, int stride, const int* __restrict__ idx, long long m,
                                                  float4* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  const long long s = idx[i];
  out[i] = make_float4(xyz[s * stride], xyz[s * stride + 1], xyz[s * stride + 2], 0.f);
}

__global__ __launch_bounds__(256) void k_iota_f4(const float* __restrict__ xyz, int stride, long long m, float4* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  out[i] = make_float4(xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2], 0.f);
}

// dataio.hpp:609-627: keypoint xyz (f32) copied into an Eigen::MatrixX3d
__global__ __launch_bounds__(256) void k_kp_xyz64(const float4* __restrict__ pts, const int* __restrict__ kp, long long k, double* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= k) return;
  const float4 p = pts[kp[i]];
  out[i * 3] = (double)p.x;
  out[i * 3 + 1] = (double)p.y;
  out[i * 3 + 2] = (double)p.z;
}

struct Ev {
  hipEvent_t e = nullptr;
  ~Ev() { if (e) (void)hipEventDestroy(e); }
};

}  // namespace

extern "C" int ghicp_register_pair(ghicp_ctx* ctx, const ghicp_pair_config* cfg, const float* xyzS, int64_t nS, const float* xyzT, int64_t nT,
                                   int stride, ghicp_pair_stats* stats, ghicp_iter* trace) {
  if (!ctx) return GHICP_ERR_ARG;
  GH_ARG(cfg != nullptr && stats != nullptr && stride >= 3 && nS >= 0 && nT >= 0 && nS < (1ll << 31) - 2 && nT < (1ll << 31) - 2);
  GH_ARG(cfg->reg.feature == GHICP_FEATURE_BSC || cfg->reg.feature == GHICP_FEATURE_NONE || cfg->reg.feature == GHICP_FEATURE_ROPS);
  hipStream_t s = ctx->stream;
  Stager sg(ctx);
  const float *dS, *dT;
  GH_TRY(sg.in(xyzS, (size_t)nS * stride, &dS));
  GH_TRY(sg.in(xyzT, (size_t)nT * stride, &dT));
  memset(stats, 0, sizeof(*stats));
  stats->n_s = nS; stats->n_t = nT;
  Ev ev[7];
  for (auto& e : ev) GH_HIP(hipEventCreate(&e.e));
  GH_HIP(hipEventRecord(ev[0].e, s));

  // ---- down-sampling (main:89-90)
  const float* cloud[2] = {dS, dT};
  const long long nraw[2] = {nS, nT};
  float4* ds[2];
  long long m[2];
  const BufSlot keepslot[2] = {B_P_KEEP_S, B_P_KEEP_T}, dsslot[2] = {B_P_DS_S, B_P_DS_T};
  for (int c = 0; c < 2; c++) {
    if (cfg->voxel > 0.f) {
      int* keep;
      GH_TRY(ctx->reserve(keepslot[c], (size_t)nraw[c] + 2, &keep));
      GH_TRY(gh_voxel_filter_dev(ctx, cloud[c], nraw[c], stride, cfg->voxel, keep, &m[c]));
      GH_TRY(ctx->reserve(dsslot[c], (size_t)m[c] + 1, &ds[c]));
      if (m[c] > 0) hipLaunchKernelGGL(k_gather4p, dim3(cdiv(m[c], 256)), dim3(256), 0, s, cloud[c], stride, keep, m[c], ds[c]);
    } else {
      m[c] = nraw[c];
      GH_TRY(ctx->reserve(dsslot[c], (size_t)m[c] + 1, &ds[c]));
      if (m[c] > 0) hipLaunchKernelGGL(k_iota_f4, dim3(cdiv(m[c], 256)), dim3(256), 0, s, cloud[c], stride, m[c], ds[c]);
    }
  }
  stats->m_s = m[0]; stats->m_t = m[1];
  // ---- bbx_magnitude of the down-sampled source (main:91-93)
  ghicp_params reg = cfg->reg;
  {
    float mm[6] = {0, 0, 0, 0, 0, 0};
    if (m[0] > 0) GH_TRY(gh_bbox_dev(ctx, reinterpret_cast<const float*>(ds[0]), m[0], 4, mm));
    reg.bbx_magnitude = (float)((double)mm[3] - (double)mm[0] + (double)mm[4] - (double)mm[1] + (double)mm[5] - (double)mm[2]);
    stats->bbx_magnitude = reg.bbx_magnitude;
  }
  GH_HIP(hipEventRecord(ev[1].e, s));

  // ---- keypoints (main:96-100); T first, then S, as the reference does
  int* kp[2];
  long long k[2] = {0, 0};
  const BufSlot kpslot[2] = {B_P_KP_S, B_P_KP_T}, kpxslot[2] = {B_P_KPXYZ_S, B_P_KPXYZ_T};
  double* kpx[2];
  for (int c = 1; c >= 0; c--) {
    GH_TRY(ctx->reserve(kpslot[c], (size_t)m[c] + 1, &kp[c]));
    GH_TRY(gh_keypoints_dev(ctx, reinterpret_cast<const float*>(ds[c]), m[c], 4, cfg->neighborhood_radius, cfg->ratio_max, cfg->min_neighbors,
                            reg.radius_nonmax, kp[c], &k[c]));
    GH_TRY(ctx->reserve(kpxslot[c], (size_t)k[c] * 3 + 3, &kpx[c]));
    if (k[c] > 0) hipLaunchKernelGGL(k_kp_xyz64, dim3(cdiv(k[c], 256)), dim3(256), 0, s, ds[c], kp[c], k[c], kpx[c]);
  }
  stats->k_s = k[0]; stats->k_t = k[1];
  GH_HIP(hipEventRecord(ev[2].e, s));

  // ---- features + feature distance (main:109-140, ghicp_reg.cpp:34-44)
  const void* FD = nullptr;
  if (reg.feature == GHICP_FEATURE_BSC) {
    uint8_t *fS, *fT;
    float* lcs;
    uint16_t* fd;
    GH_TRY(ctx->reserve(B_P_FEAT_S, (size_t)4 * k[0] * 56 + 64, &fS));
    GH_TRY(ctx->reserve(B_P_FEAT_T, (size_t)4 * k[1] * 56 + 64, &fT));
    GH_TRY(ctx->reserve(B_P_LCS, (size_t)(k[0] > k[1] ? k[0] : k[1]) * 12 + 12, &lcs));
    // BSCEncoder(curvature_non_max_radius, 7): the BSC radius is the NMS radius (main:113)
    GH_TRY(gh_bsc_dev(ctx, reinterpret_cast<const float*>(ds[1]), m[1], 4, kp[1], k[1], reg.radius_nonmax, 0, cfg->pattern, fT, lcs));
    GH_TRY(gh_bsc_dev(ctx, reinterpret_cast<const float*>(ds[0]), m[0], 4, kp[0], k[0], reg.radius_nonmax, reg.dof, cfg->pattern, fS, lcs));
    GH_HIP(hipEventRecord(ev[3].e, s));
    GH_TRY(ctx->reserve(B_P_FD, (size_t)k[0] * k[1] + 8, &fd));
    const int V = reg.dof == 6 ? 4 : 2;  // use_6dof_case_ (ghicp_reg.h:109-112, ghicp_reg.cpp:178-182)
    GH_TRY(gh_fd_bsc_dev(ctx, fS, (int)k[0], V, fT, (int)k[1], fd));
    FD = fd;
  } else {
    GH_HIP(hipEventRecord(ev[3].e, s));
  }
  GH_HIP(hipEventRecord(ev[4].e, s));

  // ---- the loop
  int32_t n_iter = 0;
  GH_TRY(gh_register_dev(ctx, &reg, kpx[0], (int)k[0], kpx[1], (int)k[1], FD, stats->Rt, trace, &n_iter, nullptr));
  GH_HIP(hipEventRecord(ev[5].e, s));
  GH_HIP(hipEventSynchronize(ev[5].e));
  stats->iterations = n_iter;
  float t;
  GH_HIP(hipEventElapsedTime(&t, ev[0].e, ev[1].e)); stats->ms_voxel = t;
  GH_HIP(hipEventElapsedTime(&t, ev[1].e, ev[2].e)); stats->ms_keypoints = t;
  GH_HIP(hipEventElapsedTime(&t, ev[2].e, ev[3].e)); stats->ms_feature = t;
  GH_HIP(hipEventElapsedTime(&t, ev[3].e, ev[4].e)); stats->ms_fd = t;
  GH_HIP(hipEventElapsedTime(&t, ev[4].e, ev[5].e)); stats->ms_loop = t;
  GH_HIP(hipEventElapsedTime(&t, ev[0].e, ev[5].e)); stats->ms_total = t;
  return GHICP_OK;
}
