// Per-cloud front-end cache.  In multi-view / all-pairs registration every cloud takes part in many
// pairs, but its down-sampling, keypoints and descriptors (test/ghicp_main.cpp:86-127) depend on the cloud alone.
// A ghicp_cloud holds those results in HBM once; ghicp_register_clouds() then needs only the feature distance and the
// GH-ICP loop per pair.  The BSC strings are computed with the registration's dof (V source variants, bfe:648-660);
// variant 0 is what extractBinaryFeatures(..., 0, ...) gives, so the same handle serves as source or as target.
// ghicp_sbf_write / ghicp_sbf_read speak the reference's dump format (StereoBinaryFeature::writeFeatures /
// readFeatures, src/stereo_binary_feature.cpp:107-148) so a cache can live on disk and come back through
// ghicp_cloud_from_features().
#include "cloud.h"

#include <cstdio>

int gh_voxel_filter_dev(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float voxel, int32_t* keep, long long* m_out);
int gh_keypoints_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float radius, float ratio_max, int min_n, float nms_radius,
                     int32_t* kp, long long* k_out);
int gh_bsc_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, const int32_t* kp, long long K, float R, int dof, const int32_t* pattern_host,
               uint8_t* feat, float* lcs);
int gh_bbox_dev(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float* mm_host6);
int gh_fpfh_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, float* normals_opt, float* hist);
int gh_gather_rows33_dev(ghicp_ctx* ctx, const float* hist, const int32_t* idx, long long k, float* out);
int gh_fd_fpfh_dev(ghicp_ctx* ctx, const float* histS, int ks, const float* histT, int kt, float* FD);

namespace {

__global__ __launch_bounds__(256) void k_cl_gather4(const float* __restrict__ xyz, int stride, const int* __restrict__ idx, long long m,
                                                    float4* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  const long long s = idx ? (long long)idx[i] : i;
  out[i] = make_float4(xyz[s * stride], xyz[s * stride + 1], xyz[s * stride + 2], 0.f);
}

__global__ __launch_bounds__(256) void k_cl_kpxyz(const float4* __restrict__ pts, const int* __restrict__ kp, long long k, double* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= k) return;
  const float4 p = pts[kp[i]];
  out[i * 3] = (double)p.x; out[i * 3 + 1] = (double)p.y; out[i * 3 + 2] = (double)p.z;
}

}  // namespace

// (re)computes the cached results of `c` for a raw cloud; buffers are grow-only, so a handle that is recomputed every
// step of a steady-state pipeline performs no allocation.  Ends with a stream synchronisation: afterwards the handle may
// be used from any context of the same device.
static int cloud_fill(ghicp_ctx* ctx, ghicp_cloud* c, const float* d, long long n, int stride) {
  hipStream_t s = ctx->stream;
  const ghicp_pair_config* cfg = &c->cfg;
  c->n = n; c->m = 0; c->k = 0; c->cand = 0;
  c->V = cfg->reg.dof > 4 ? 4 : (cfg->reg.dof > 0 ? 2 : 1);
  // down-sampling (main:89-90)
  if (cfg->voxel > 0.f) {
    int* keep;
    GH_TRY(ctx->reserve(B_P_KEEP_S, (size_t)n + 2, &keep));
    GH_TRY(gh_voxel_filter_dev(ctx, d, n, stride, cfg->voxel, keep, &c->m));
    GH_HIP(c->ds.reserve(((size_t)c->m + 1) * sizeof(float4)));
    if (c->m > 0) hipLaunchKernelGGL(k_cl_gather4, dim3(cdiv(c->m, 256)), dim3(256), 0, s, d, stride, keep, c->m, c->ds.as<float4>());
  } else {
    c->m = n;
    GH_HIP(c->ds.reserve(((size_t)c->m + 1) * sizeof(float4)));
    if (c->m > 0) hipLaunchKernelGGL(k_cl_gather4, dim3(cdiv(c->m, 256)), dim3(256), 0, s, d, stride, (const int*)nullptr, c->m, c->ds.as<float4>());
  }
  const float* ds = reinterpret_cast<const float*>(c->ds.p);
  {  // bounding-box magnitude (main:91-93), used as Energyfunction scale when the cloud is the source
    float mm[6] = {0, 0, 0, 0, 0, 0};
    if (c->m > 0) GH_TRY(gh_bbox_dev(ctx, ds, c->m, 4, mm));
    c->bbx = (float)((double)mm[3] - (double)mm[0] + (double)mm[4] - (double)mm[1] + (double)mm[5] - (double)mm[2]);
    if (c->m > 0) {  // the grids of the keypoint detector and of the BSC encoder are built over the same cloud
      ctx->bbox_valid = true; ctx->bbox_ptr = ds; ctx->bbox_n = c->m; ctx->bbox_stride = 4;
      memcpy(ctx->bbox_mm, mm, sizeof(mm));
    }
  }
  struct BoxScope { ghicp_ctx* c; ~BoxScope() { c->bbox_valid = false; } } box_scope{ctx};
  // keypoints (main:96-100) and their coordinates as f64 (dataio.hpp:609-627)
  GH_HIP(c->kp.reserve(((size_t)c->m + 1) * sizeof(int)));
  GH_TRY(gh_keypoints_dev(ctx, ds, c->m, 4, cfg->neighborhood_radius, cfg->ratio_max, cfg->min_neighbors, cfg->reg.radius_nonmax, c->kp.as<int>(), &c->k));
  GH_HIP(c->kpx.reserve(((size_t)c->k * 3 + 3) * sizeof(double)));
  if (c->k > 0) hipLaunchKernelGGL(k_cl_kpxyz, dim3(cdiv(c->k, 256)), dim3(256), 0, s, c->ds.as<float4>(), c->kp.as<int>(), c->k, c->kpx.as<double>());
  // descriptors (main:109-127)
  if (cfg->reg.feature == GHICP_FEATURE_BSC) {
    float* lcs;
    GH_TRY(ctx->reserve(B_P_LCS, (size_t)c->k * 12 + 12, &lcs));
    GH_HIP(c->feat.reserve((size_t)4 * c->k * 56 + 64));
    GH_TRY(gh_bsc_dev(ctx, ds, c->m, 4, c->kp.as<int>(), c->k, cfg->reg.radius_nonmax, cfg->reg.dof, cfg->pattern, c->feat.as<uint8_t>(), lcs));
  } else if (cfg->reg.feature == GHICP_FEATURE_FPFH) {
    float* hist;
    GH_TRY(ctx->reserve(B_P_FEAT_S, (size_t)c->m * 33 * sizeof(float) + 64, (char**)&hist));
    GH_HIP(c->feat.reserve(((size_t)c->k * 33 + 33) * sizeof(float)));
    GH_TRY(gh_fpfh_dev(ctx, ds, c->m, 4, nullptr, hist));
    GH_TRY(gh_gather_rows33_dev(ctx, hist, c->kp.as<int>(), c->k, c->feat.as<float>()));
  }
  GH_HIP(hipStreamSynchronize(s));
  return GHICP_OK;
}

extern "C" int ghicp_cloud_create(ghicp_ctx* ctx, const ghicp_pair_config* cfg, const float* xyz, int64_t n, int stride, ghicp_cloud** out) {
  GH_ENTER(ctx);
  GH_ARG(cfg != nullptr && out != nullptr && stride >= 3 && n >= 0 && n < (1ll << 31) - 2);
  GH_ARG(cfg->reg.feature >= GHICP_FEATURE_BSC && cfg->reg.feature <= GHICP_FEATURE_NONE);
  Stager sg(ctx);
  const float* d;
  GH_TRY(sg.in_cloud(xyz, (size_t)n * stride, &d));
  ghicp_cloud* c = new ghicp_cloud();
  c->ctx = ctx; c->cfg = *cfg;
  const int rc = cloud_fill(ctx, c, d, n, stride);
  if (rc != GHICP_OK) { ghicp_cloud_destroy(c); return rc; }
  *out = c;
  return GHICP_OK;
}

// Recomputes an existing handle for another raw cloud (same front-end configuration), reusing its buffers.
extern "C" int ghicp_cloud_recompute(ghicp_cloud* c, const float* xyz, int64_t n, int stride) {
  if (!c || !c->ctx) return GHICP_ERR_ARG;
  ghicp_ctx* ctx = c->ctx;
  GH_ENTER(ctx);
  GH_ARG(stride >= 3 && n >= 0 && n < (1ll << 31) - 2);
  Stager sg(ctx);
  const float* d;
  GH_TRY(sg.in_cloud(xyz, (size_t)n * stride, &d));
  return cloud_fill(ctx, c, d, n, stride);
}

extern "C" int ghicp_cloud_from_features(ghicp_ctx* ctx, const ghicp_pair_config* cfg, const double* kp_xyz, int64_t k, const void* feat,
                                         float bbx_magnitude, ghicp_cloud** out) {
  GH_ENTER(ctx);
  GH_ARG(cfg != nullptr && out != nullptr && k >= 0 && k < (1 << 24) && (k == 0 || kp_xyz != nullptr));
  GH_ARG(cfg->reg.feature >= GHICP_FEATURE_BSC && cfg->reg.feature <= GHICP_FEATURE_NONE);
  GH_ARG(cfg->reg.feature == GHICP_FEATURE_NONE || k == 0 || feat != nullptr);
  hipStream_t s = ctx->stream;
  const hipMemcpyKind kind = ctx->host_ptrs ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  ghicp_cloud* c = new ghicp_cloud();
  struct Guard { ghicp_cloud* c; ~Guard() { if (c) { c->kpx.release(); c->feat.release(); delete c; } } } guard{c};
  c->ctx = ctx; c->cfg = *cfg; c->k = k; c->bbx = bbx_magnitude;
  c->V = cfg->reg.dof > 4 ? 4 : (cfg->reg.dof > 0 ? 2 : 1);
  GH_HIP(c->kpx.reserve(((size_t)k * 3 + 3) * sizeof(double)));
  if (k > 0) GH_HIP(hipMemcpyAsync(c->kpx.p, kp_xyz, (size_t)k * 3 * sizeof(double), kind, s));
  if (cfg->reg.feature == GHICP_FEATURE_BSC) {
    GH_HIP(c->feat.reserve((size_t)4 * k * 56 + 64));
    GH_HIP(hipMemsetAsync(c->feat.p, 0, (size_t)4 * k * 56, s));
    if (k > 0) GH_HIP(hipMemcpyAsync(c->feat.p, feat, (size_t)c->V * k * 56, kind, s));
  } else if (cfg->reg.feature == GHICP_FEATURE_FPFH) {
    GH_HIP(c->feat.reserve(((size_t)k * 33 + 33) * sizeof(float)));
    if (k > 0) GH_HIP(hipMemcpyAsync(c->feat.p, feat, (size_t)k * 33 * sizeof(float), kind, s));
  }
  GH_HIP(hipStreamSynchronize(s));
  guard.c = nullptr;
  *out = c;
  return GHICP_OK;
}

extern "C" int ghicp_cloud_destroy(ghicp_cloud* c) {
  if (!c) return GHICP_OK;
  if (c->ctx) (void)hipSetDevice(c->ctx->device);
  c->ds.release(); c->kp.release(); c->kpx.release(); c->feat.release();
  delete c;
  return GHICP_OK;
}

extern "C" int ghicp_cloud_get_info(const ghicp_cloud* c, ghicp_cloud_info* info) {
  if (!c || !info) return GHICP_ERR_ARG;
  info->n = c->n; info->m = c->m; info->k = c->k;
  info->variants = c->V;
  info->feature = c->cfg.reg.feature;
  info->bbx_magnitude = c->bbx;
  info->candidates = (int32_t)c->cand;
  info->feature_bytes = c->cfg.reg.feature == GHICP_FEATURE_BSC ? (int64_t)c->V * c->k * 56
                        : (c->cfg.reg.feature == GHICP_FEATURE_FPFH ? (int64_t)c->k * 33 * 4 : 0);
  return GHICP_OK;
}

// Copies the cached results out (any pointer may be NULL): down-sampled points m x 3 f32, keypoint ids k, keypoint
// coordinates k x 3 f64, descriptors (info.feature_bytes).  Destination memory follows the context's pointer mode.
extern "C" int ghicp_cloud_download(const ghicp_cloud* c, float* ds_xyz, int32_t* kp_idx, double* kp_xyz, void* feat) {
  if (!c) return GHICP_ERR_ARG;
  ghicp_ctx* ctx = c->ctx;
  GH_ENTER(ctx);
  hipStream_t s = ctx->stream;
  const hipMemcpyKind kind = ctx->host_ptrs ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (ds_xyz && c->m > 0) {
    if (!c->ds.p) return ctx->fail(GHICP_ERR_ARG, "ghicp_cloud_download: this handle was rebuilt from stored features and holds no points");
    GH_HIP(hipMemcpy2DAsync(ds_xyz, 12, c->ds.p, 16, 12, (size_t)c->m, kind, s));
  }
  if (kp_idx && c->k > 0) {
    if (!c->kp.p) return ctx->fail(GHICP_ERR_ARG, "ghicp_cloud_download: this handle holds no keypoint ids");
    GH_HIP(hipMemcpyAsync(kp_idx, c->kp.p, (size_t)c->k * sizeof(int32_t), kind, s));
  }
  if (kp_xyz && c->k > 0) GH_HIP(hipMemcpyAsync(kp_xyz, c->kpx.p, (size_t)c->k * 3 * sizeof(double), kind, s));
  ghicp_cloud_info info;
  ghicp_cloud_get_info(c, &info);
  if (feat && info.feature_bytes > 0) GH_HIP(hipMemcpyAsync(feat, c->feat.p, (size_t)info.feature_bytes, kind, s));
  GH_HIP(hipStreamSynchronize(s));
  return GHICP_OK;
}

// Registers n_pairs (S[i] -> T[i]) from cached front ends: feature distance per pair, then one batched GH-ICP loop.
extern "C" int ghicp_register_clouds(ghicp_ctx* ctx, const ghicp_pair_config* cfg, int32_t n_pairs, const ghicp_cloud* const* S,
                                     const ghicp_cloud* const* T, ghicp_pair_stats* stats) {
  GH_ENTER(ctx);
  if (n_pairs > 0) { ctx->loop_total.store(n_pairs, std::memory_order_relaxed); ctx->loop_active.store(n_pairs, std::memory_order_relaxed); }  // progress: nothing done yet
  GH_ARG(cfg != nullptr && stats != nullptr && n_pairs >= 0 && n_pairs <= 65535 && (n_pairs == 0 || (S && T)));
  if (n_pairs == 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  if (ctx->pairbuf.size() < (size_t)n_pairs * 3) ctx->pairbuf.resize((size_t)n_pairs * 3);
  std::vector<ghicp_params> reg(n_pairs);
  std::vector<gh_loop_job> jobs(n_pairs);
  std::vector<gh_fd_bsc_job> fdj(n_pairs);
  std::vector<int32_t> iters(n_pairs, 0), conv(n_pairs, 0);
  hipEvent_t e0, e1, e2;
  GH_HIP(hipEventCreate(&e0)); GH_HIP(hipEventCreate(&e1)); GH_HIP(hipEventCreate(&e2));
  struct EvGuard { hipEvent_t a, b, c; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); } } eg{e0, e1, e2};
  GH_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < n_pairs; i++) {
    const ghicp_cloud *a = S[i], *b = T[i];
    GH_ARG(a != nullptr && b != nullptr && a->ctx && b->ctx && a->ctx->device == ctx->device && b->ctx->device == ctx->device);
    if (!same_front_end(a->cfg, *cfg) || !same_front_end(b->cfg, *cfg))
      return ctx->fail(GHICP_ERR_ARG, "ghicp_register_clouds: pair %d was cached with a different front-end configuration", i);
    memset(&stats[i], 0, sizeof(stats[i]));
    stats[i].n_s = a->n; stats[i].n_t = b->n; stats[i].m_s = a->m; stats[i].m_t = b->m; stats[i].k_s = a->k; stats[i].k_t = b->k;
    stats[i].bbx_magnitude = a->bbx;
    reg[i] = cfg->reg;
    reg[i].bbx_magnitude = a->bbx;
    DevBuf& fd = ctx->pairbuf[(size_t)i * 3 + 2];
    const void *FD = nullptr, *FDt = nullptr;
    if (cfg->reg.feature == GHICP_FEATURE_BSC) {
      // matrix and transposed copy side by side; every pair of the call goes through ONE launch of the MFMA Hamming kernel below
      const size_t cells = ((size_t)a->k * b->k + 15) & ~(size_t)7;
      GH_HIP(fd.reserve(2 * cells * sizeof(uint16_t)));
      FD = fd.p;
      FDt = fd.as<uint16_t>() + cells;
      fdj[i] = {a->feat.as<uint8_t>(), b->feat.as<uint8_t>(), fd.as<uint16_t>(), fd.as<uint16_t>() + cells, (int)a->k, (int)b->k, a->V};
    } else if (cfg->reg.feature == GHICP_FEATURE_FPFH) {
      GH_HIP(fd.reserve(((size_t)a->k * b->k + 8) * sizeof(float)));
      GH_TRY(gh_fd_fpfh_dev(ctx, a->feat.as<float>(), (int)a->k, b->feat.as<float>(), (int)b->k, fd.as<float>()));
      FD = fd.p;
    }
    gh_loop_job& J = jobs[i];
    memset(&J, 0, sizeof(J));
    J.p = &reg[i]; J.kpS = a->kpx.as<double>(); J.ks = (int)a->k; J.kpT = b->kpx.as<double>(); J.kt = (int)b->k; J.FD = FD; J.FDt = FDt; J.Rt16 = stats[i].Rt;
    J.n_iter = &iters[i]; J.converged = &conv[i]; J.rmse_after = &stats[i].rmse_after;
  }
  if (cfg->reg.feature == GHICP_FEATURE_BSC) GH_TRY(gh_fd_bsc_batch_dev(ctx, n_pairs, fdj.data()));
  GH_HIP(hipEventRecord(e1, s));
  GH_TRY(gh_register_batch_dev(ctx, n_pairs, jobs.data()));
  GH_HIP(hipEventRecord(e2, s));
  GH_HIP(hipEventSynchronize(e2));
  float tf = 0, tl = 0;
  GH_HIP(hipEventElapsedTime(&tf, e0, e1));
  GH_HIP(hipEventElapsedTime(&tl, e1, e2));
  for (int i = 0; i < n_pairs; i++) {
    stats[i].iterations = iters[i];
    stats[i].converged = conv[i];
    stats[i].registered_ok = gh_registered_ok(conv[i], stats[i].rmse_after, cfg->reg.radius_nonmax);
    stats[i].ms_fd = tf / n_pairs;
    stats[i].ms_loop = tl / n_pairs;
    stats[i].ms_total = (tf + tl) / n_pairs;
  }
  return GHICP_OK;
}

// ghicp_register_pairs through the batched front end: the 2 n raw clouds become (context-owned, reused) cloud handles filled by
// ghicp_clouds_recompute -- one launch sequence per 64 clouds instead of ~100 operations per cloud --, then the pairs are registered
// from the handles.  Same results as the pair-by-pair front end (the target's strings are variant 0 of its handle, bfe:648-660).
// Returns GHICP_OK with *handled = 0 when the configuration is not covered by the batch (no down-sampling, host pointers).
int gh_register_pairs_batched(ghicp_ctx* ctx, const ghicp_pair_config* cfg, int32_t n_pairs, const float* const* xyzS, const int64_t* nS,
                              const float* const* xyzT, const int64_t* nT, int stride, ghicp_pair_stats* stats, int* handled) {
  *handled = 0;
  // (FPFH batches go cloud by cloud here until the FPFH branch of the batched front end has been through the GPU suite once)
  if (cfg->reg.feature == GHICP_FEATURE_FPFH || !(cfg->voxel > 0.f) || ctx->host_ptrs || n_pairs <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  std::vector<ghicp_cloud*>& pool = ctx->pair_clouds;
  while (pool.size() < (size_t)n_pairs * 2) {
    ghicp_cloud* c = new ghicp_cloud();
    c->ctx = ctx;
    pool.push_back(c);
  }
  std::vector<ghicp_cloud*> h((size_t)n_pairs * 2);
  std::vector<const float*> xyz((size_t)n_pairs * 2);
  std::vector<int64_t> n((size_t)n_pairs * 2);
  for (int i = 0; i < n_pairs; i++) {
    h[(size_t)i * 2] = pool[(size_t)i * 2]; h[(size_t)i * 2 + 1] = pool[(size_t)i * 2 + 1];
    xyz[(size_t)i * 2] = xyzS[i]; xyz[(size_t)i * 2 + 1] = xyzT[i];
    n[(size_t)i * 2] = nS[i]; n[(size_t)i * 2 + 1] = nT[i];
  }
  for (ghicp_cloud* c : h) c->cfg = *cfg;
  hipEvent_t e0, e1;
  GH_HIP(hipEventCreate(&e0)); GH_HIP(hipEventCreate(&e1));
  struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } eg{e0, e1};
  GH_HIP(hipEventRecord(e0, s));
  GH_TRY(ghicp_clouds_recompute(ctx, n_pairs * 2, h.data(), xyz.data(), n.data(), stride));
  GH_HIP(hipEventRecord(e1, s));
  std::vector<const ghicp_cloud*> S(n_pairs), T(n_pairs);
  for (int i = 0; i < n_pairs; i++) { S[i] = h[(size_t)i * 2]; T[i] = h[(size_t)i * 2 + 1]; }
  GH_TRY(ghicp_register_clouds(ctx, cfg, n_pairs, S.data(), T.data(), stats));
  float tf = 0;
  GH_HIP(hipEventElapsedTime(&tf, e0, e1));
  for (int i = 0; i < n_pairs; i++) {
    stats[i].ms_keypoints = tf / n_pairs + stats[i].ms_fd;  // whole front end (voxel + keypoints + feature + FD), batch average
    stats[i].ms_fd = 0.f;
    stats[i].ms_total = stats[i].ms_keypoints + stats[i].ms_loop;
  }
  *handled = 1;
  return GHICP_OK;
}

// ---- StereoBinaryFeature::writeFeatures / readFeatures (src/stereo_binary_feature.cpp:107-148): u32 bit count, u32 byte
// count, i32 number of features, then byte_ bytes per feature.  Host memory, no context.
extern "C" int ghicp_sbf_write(const char* path, const uint8_t* feat, int64_t k) {
  if (!path || k < 0 || k > 0x7fffffff || (k > 0 && !feat)) return GHICP_ERR_ARG;
  FILE* f = fopen(path, "wb");
  if (!f) return GHICP_ERR_ARG;
  const unsigned bits = 441, bytes = 56;
  const int cnt = (int)k;
  bool ok = fwrite(&bits, 4, 1, f) == 1 && fwrite(&bytes, 4, 1, f) == 1 && fwrite(&cnt, 4, 1, f) == 1;
  if (ok && k > 0) ok = fwrite(feat, 56, (size_t)k, f) == (size_t)k;
  ok = (fclose(f) == 0) && ok;
  return ok ? GHICP_OK : GHICP_ERR_ARG;
}

extern "C" int ghicp_sbf_read(const char* path, uint8_t* feat, int64_t capacity, int64_t* k) {
  if (!path || !k) return GHICP_ERR_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return GHICP_ERR_ARG;
  unsigned bits = 0, bytes = 0;
  int cnt = 0;
  bool ok = fread(&bits, 4, 1, f) == 1 && fread(&bytes, 4, 1, f) == 1 && fread(&cnt, 4, 1, f) == 1 && bits == 441 && bytes == 56 && cnt >= 0;
  if (ok) {
    *k = cnt;
    if (feat) ok = cnt <= capacity && (cnt == 0 || fread(feat, 56, (size_t)cnt, f) == (size_t)cnt);
  }
  fclose(f);
  return ok ? GHICP_OK : GHICP_ERR_ARG;
}
