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
int gh_fpfh_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float* normals_opt, float* hist);
int gh_gather_rows33_dev(ghicp_ctx* ctx, const float* hist, const int32_t* idx, long long k, float* out);
int gh_fd_fpfh_dev(ghicp_ctx* ctx, const float* histS, int ks, const float* histT, int kt, float* FD);
int gh_register_pairs_batched(ghicp_ctx* ctx, const ghicp_pair_config* cfg, int32_t n_pairs, const float* const* xyzS, const int64_t* nS,
                              const float* const* xyzT, const int64_t* nT, int stride, ghicp_pair_stats* stats, int* handled);  // cloud.hip

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

// Front end of one pair (main:86-140): down-sample, keypoints, BSC, feature distance.  The three outputs the
// loop needs (keypoint coordinates of both clouds and the FD matrix) go into caller-provided grow-only buffers
// so that a whole batch of pairs can be resident when the batched loop starts.
struct PairFront {
  ghicp_params reg;
  long long m[2], k[2];
  double* kpx[2];
  const void* FD;
};

static int front_end(ghicp_ctx* ctx, const ghicp_pair_config* cfg, const float* dS, long long nS, const float* dT, long long nT, int stride,
                     DevBuf* out_kps, DevBuf* out_kpt, DevBuf* out_fd, PairFront* F, hipEvent_t* ev /*5 or null*/) {
  hipStream_t s = ctx->stream;
  if (ev) GH_HIP(hipEventRecord(ev[0], s));
  // ---- down-sampling (main:89-90)
  const float* cloud[2] = {dS, dT};
  const long long nraw[2] = {nS, nT};
  float4* ds[2];
  long long* m = F->m;
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
  // ---- bbx_magnitude of the down-sampled source (main:91-93)
  F->reg = cfg->reg;
  {
    float mm[6] = {0, 0, 0, 0, 0, 0};
    if (m[0] > 0) GH_TRY(gh_bbox_dev(ctx, reinterpret_cast<const float*>(ds[0]), m[0], 4, mm));
    F->reg.bbx_magnitude = (float)((double)mm[3] - (double)mm[0] + (double)mm[4] - (double)mm[1] + (double)mm[5] - (double)mm[2]);
  }
  if (ev) GH_HIP(hipEventRecord(ev[1], s));

  // ---- keypoints (main:96-100); T first, then S, as the reference does
  int* kp[2];
  long long* k = F->k;
  k[0] = k[1] = 0;
  const BufSlot kpslot[2] = {B_P_KP_S, B_P_KP_T};
  DevBuf* kpxbuf[2] = {out_kps, out_kpt};
  for (int c = 1; c >= 0; c--) {
    GH_TRY(ctx->reserve(kpslot[c], (size_t)m[c] + 1, &kp[c]));
    GH_TRY(gh_keypoints_dev(ctx, reinterpret_cast<const float*>(ds[c]), m[c], 4, cfg->neighborhood_radius, cfg->ratio_max, cfg->min_neighbors,
                            F->reg.radius_nonmax, kp[c], &k[c]));
    GH_HIP(kpxbuf[c]->reserve(((size_t)k[c] * 3 + 3) * sizeof(double)));
    F->kpx[c] = kpxbuf[c]->as<double>();
    if (k[c] > 0) hipLaunchKernelGGL(k_kp_xyz64, dim3(cdiv(k[c], 256)), dim3(256), 0, s, ds[c], kp[c], k[c], F->kpx[c]);
  }
  if (ev) GH_HIP(hipEventRecord(ev[2], s));

  // ---- features + feature distance (main:109-140, ghicp_reg.cpp:34-44)
  F->FD = nullptr;
  if (F->reg.feature == GHICP_FEATURE_BSC) {
    uint8_t *fS, *fT;
    float* lcs;
    GH_TRY(ctx->reserve(B_P_FEAT_S, (size_t)4 * k[0] * 56 + 64, &fS));
    GH_TRY(ctx->reserve(B_P_FEAT_T, (size_t)4 * k[1] * 56 + 64, &fT));
    GH_TRY(ctx->reserve(B_P_LCS, (size_t)(k[0] > k[1] ? k[0] : k[1]) * 12 + 12, &lcs));
    // BSCEncoder(curvature_non_max_radius, 7): the BSC radius is the NMS radius (main:113)
    GH_TRY(gh_bsc_dev(ctx, reinterpret_cast<const float*>(ds[1]), m[1], 4, kp[1], k[1], F->reg.radius_nonmax, 0, cfg->pattern, fT, lcs));
    GH_TRY(gh_bsc_dev(ctx, reinterpret_cast<const float*>(ds[0]), m[0], 4, kp[0], k[0], F->reg.radius_nonmax, F->reg.dof, cfg->pattern, fS, lcs));
    if (ev) GH_HIP(hipEventRecord(ev[3], s));
    GH_HIP(out_fd->reserve(((size_t)k[0] * k[1] + 8) * sizeof(uint16_t)));
    const int V = F->reg.dof == 6 ? 4 : 2;  // use_6dof_case_ (ghicp_reg.h:109-112, ghicp_reg.cpp:178-182)
    GH_TRY(gh_fd_bsc_dev(ctx, fS, (int)k[0], V, fT, (int)k[1], out_fd->as<uint16_t>()));
    F->FD = out_fd->p;
  } else if (F->reg.feature == GHICP_FEATURE_FPFH) {
    // compute_fpfh_feature over the WHOLE down-sampled clouds, then keyfpfh (main:122-127, fpfh.hpp:36-58, 93-115)
    float *hist, *kS, *kT;
    const long long mmax = m[0] > m[1] ? m[0] : m[1];
    GH_TRY(ctx->reserve(B_P_FEAT_S, (size_t)mmax * 33 * sizeof(float) + 64, (char**)&hist));
    GH_TRY(ctx->reserve(B_P_FEAT_T, (size_t)(k[0] + k[1]) * 33 * sizeof(float) + 64, (char**)&kS));
    kT = kS + (size_t)k[0] * 33;
    GH_TRY(gh_fpfh_dev(ctx, reinterpret_cast<const float*>(ds[1]), m[1], 4, nullptr, hist));
    GH_TRY(gh_gather_rows33_dev(ctx, hist, kp[1], k[1], kT));
    GH_TRY(gh_fpfh_dev(ctx, reinterpret_cast<const float*>(ds[0]), m[0], 4, nullptr, hist));
    GH_TRY(gh_gather_rows33_dev(ctx, hist, kp[0], k[0], kS));
    if (ev) GH_HIP(hipEventRecord(ev[3], s));
    GH_HIP(out_fd->reserve(((size_t)k[0] * k[1] + 8) * sizeof(float)));
    GH_TRY(gh_fd_fpfh_dev(ctx, kS, (int)k[0], kT, (int)k[1], out_fd->as<float>()));
    F->FD = out_fd->p;
  } else if (ev) {
    GH_HIP(hipEventRecord(ev[3], s));
  }
  if (ev) GH_HIP(hipEventRecord(ev[4], s));
  return GHICP_OK;
}

extern "C" int ghicp_register_pair(ghicp_ctx* ctx, const ghicp_pair_config* cfg, const float* xyzS, int64_t nS, const float* xyzT, int64_t nT,
                                   int stride, ghicp_pair_stats* stats, ghicp_iter* trace) {
  GH_ENTER(ctx);
  GH_ARG(cfg != nullptr && stats != nullptr && stride >= 3 && nS >= 0 && nT >= 0 && nS < (1ll << 31) - 2 && nT < (1ll << 31) - 2);
  GH_ARG(cfg->reg.feature >= GHICP_FEATURE_BSC && cfg->reg.feature <= GHICP_FEATURE_NONE);
  hipStream_t s = ctx->stream;
  Stager sg(ctx);
  const float *dS, *dT;
  GH_TRY(sg.in_cloud(xyzS, (size_t)nS * stride, &dS));
  GH_TRY(sg.in_cloud(xyzT, (size_t)nT * stride, &dT));
  memset(stats, 0, sizeof(*stats));
  stats->n_s = nS; stats->n_t = nT;
  Ev ev[7];
  hipEvent_t raw[6];
  for (int i = 0; i < 6; i++) { GH_HIP(hipEventCreate(&ev[i].e)); raw[i] = ev[i].e; }
  PairFront F;
  GH_TRY(front_end(ctx, cfg, dS, nS, dT, nT, stride, &ctx->buf[B_P_KPXYZ_S], &ctx->buf[B_P_KPXYZ_T], &ctx->buf[B_P_FD], &F, raw));
  stats->m_s = F.m[0]; stats->m_t = F.m[1]; stats->k_s = F.k[0]; stats->k_t = F.k[1];
  stats->bbx_magnitude = F.reg.bbx_magnitude;
  int32_t n_iter = 0, conv = 0;
  gh_loop_job J;
  memset(&J, 0, sizeof(J));
  J.p = &F.reg; J.kpS = F.kpx[0]; J.ks = (int)F.k[0]; J.kpT = F.kpx[1]; J.kt = (int)F.k[1]; J.FD = F.FD; J.Rt16 = stats->Rt; J.trace = trace;
  J.n_iter = &n_iter; J.converged = &conv; J.rmse_after = &stats->rmse_after;
  GH_TRY(gh_register_batch_dev(ctx, 1, &J));
  GH_HIP(hipEventRecord(raw[5], s));
  GH_HIP(hipEventSynchronize(raw[5]));
  stats->iterations = n_iter;
  stats->converged = conv;
  stats->registered_ok = gh_registered_ok(conv, stats->rmse_after, F.reg.radius_nonmax);
  float t;
  GH_HIP(hipEventElapsedTime(&t, raw[0], raw[1])); stats->ms_voxel = t;
  GH_HIP(hipEventElapsedTime(&t, raw[1], raw[2])); stats->ms_keypoints = t;
  GH_HIP(hipEventElapsedTime(&t, raw[2], raw[3])); stats->ms_feature = t;
  GH_HIP(hipEventElapsedTime(&t, raw[3], raw[4])); stats->ms_fd = t;
  GH_HIP(hipEventElapsedTime(&t, raw[4], raw[5])); stats->ms_loop = t;
  GH_HIP(hipEventElapsedTime(&t, raw[0], raw[5])); stats->ms_total = t;
  return GHICP_OK;
}

// A batch of independent pairs (BASELINE configs[3]): front ends one after the other on the
// context's stream, then ONE batched loop in which every pair's sweep / KM solve / rigid solve runs concurrently.
// stats[i].ms_* hold the batch-level timings divided by the number of pairs (ms_total = batch wall time / n).
extern "C" int ghicp_register_pairs(ghicp_ctx* ctx, const ghicp_pair_config* cfg, int32_t n_pairs, const float* const* xyzS, const int64_t* nS,
                                    const float* const* xyzT, const int64_t* nT, int stride, ghicp_pair_stats* stats) {
  GH_ENTER(ctx);
  GH_ARG(cfg != nullptr && stats != nullptr && n_pairs >= 0 && n_pairs <= 65535 && stride >= 3 && xyzS && xyzT && nS && nT);
  GH_ARG(cfg->reg.feature >= GHICP_FEATURE_BSC && cfg->reg.feature <= GHICP_FEATURE_NONE);
  if (ctx->host_ptrs) return ctx->fail(GHICP_ERR_ARG, "ghicp_register_pairs: device-pointer mode only");
  if (n_pairs == 0) return GHICP_OK;
  for (int i = 0; i < n_pairs; i++) GH_ARG(nS[i] >= 0 && nT[i] >= 0 && nS[i] < (1ll << 31) - 2 && nT[i] < (1ll << 31) - 2);
  if (n_pairs > 1) {  // the batched front end (batch.hip) when the configuration allows it
    int handled = 0;
    GH_TRY(gh_register_pairs_batched(ctx, cfg, n_pairs, xyzS, nS, xyzT, nT, stride, stats, &handled));
    if (handled) return GHICP_OK;
  }
  hipStream_t s = ctx->stream;
  if (ctx->pairbuf.size() < (size_t)n_pairs * 3) ctx->pairbuf.resize((size_t)n_pairs * 3);
  std::vector<PairFront> F(n_pairs);
  std::vector<gh_loop_job> jobs(n_pairs);
  std::vector<int32_t> iters(n_pairs, 0), conv(n_pairs, 0);
  Ev e0, e1, e2;
  GH_HIP(hipEventCreate(&e0.e)); GH_HIP(hipEventCreate(&e1.e)); GH_HIP(hipEventCreate(&e2.e));
  GH_HIP(hipEventRecord(e0.e, s));
  for (int i = 0; i < n_pairs; i++) {
    GH_ARG(nS[i] >= 0 && nT[i] >= 0 && nS[i] < (1ll << 31) - 2 && nT[i] < (1ll << 31) - 2);
    memset(&stats[i], 0, sizeof(stats[i]));
    GH_TRY(front_end(ctx, cfg, xyzS[i], nS[i], xyzT[i], nT[i], stride, &ctx->pairbuf[(size_t)i * 3], &ctx->pairbuf[(size_t)i * 3 + 1],
                     &ctx->pairbuf[(size_t)i * 3 + 2], &F[i], nullptr));
    stats[i].n_s = nS[i]; stats[i].n_t = nT[i]; stats[i].m_s = F[i].m[0]; stats[i].m_t = F[i].m[1]; stats[i].k_s = F[i].k[0]; stats[i].k_t = F[i].k[1];
    stats[i].bbx_magnitude = F[i].reg.bbx_magnitude;
    gh_loop_job& J = jobs[i];
    memset(&J, 0, sizeof(J));
    J.p = &F[i].reg; J.kpS = F[i].kpx[0]; J.ks = (int)F[i].k[0]; J.kpT = F[i].kpx[1]; J.kt = (int)F[i].k[1]; J.FD = F[i].FD; J.Rt16 = stats[i].Rt;
    J.n_iter = &iters[i]; J.converged = &conv[i]; J.rmse_after = &stats[i].rmse_after;
  }
  GH_HIP(hipEventRecord(e1.e, s));
  GH_TRY(gh_register_batch_dev(ctx, n_pairs, jobs.data()));
  GH_HIP(hipEventRecord(e2.e, s));
  GH_HIP(hipEventSynchronize(e2.e));
  float tf = 0, tl = 0;
  GH_HIP(hipEventElapsedTime(&tf, e0.e, e1.e));
  GH_HIP(hipEventElapsedTime(&tl, e1.e, e2.e));
  for (int i = 0; i < n_pairs; i++) {
    stats[i].iterations = iters[i];
    stats[i].converged = conv[i];
    stats[i].registered_ok = gh_registered_ok(conv[i], stats[i].rmse_after, cfg->reg.radius_nonmax);
    stats[i].ms_keypoints = tf / n_pairs;  // whole front end (voxel + keypoints + feature + FD), batch average
    stats[i].ms_loop = tl / n_pairs;
    stats[i].ms_total = (tf + tl) / n_pairs;
  }
  return GHICP_OK;
}
