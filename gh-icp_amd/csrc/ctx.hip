// Context management + small stand-alone entry points (rigid solve, cloud transform, gather, bbox).
#include "ctx.h"
#include "devmath.h"

extern "C" const char* ghicp_version(void) { return "ghicp-hip 0.1 (gfx950)"; }

extern "C" void ghicp_params_default(ghicp_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->feature = GHICP_FEATURE_BSC;
  p->corr = GHICP_CORR_KM;
  p->dof = 6;
  p->max_iter = 200;  // = the column count of the reference's matchlist (ghicp_reg.h:100)
  p->radius_nonmax = 1.5f;
  p->adjust_ratio = 1.1f;  // README.md:79-85
  p->adjust_step = 0.1f;
  p->est_iou = 0.6f;
  p->converge_t = 0.02f;  // ghicp_reg.h:80
  p->converge_r = 0.02f;
  p->bbx_magnitude = 0.f;
  p->penalty_initial = 2.0;  // ghicp_reg.h:32-38
  p->para1 = 1.0;
  p->para2 = 1.0;
  p->km_eps = 0.01;
  p->min_cor = 10;
  p->weight_changing_rate = 6;
}

extern "C" int ghicp_ctx_create(int device, ghicp_ctx** out) {
  if (!out) return GHICP_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return GHICP_ERR_NO_GPU;
  if (hipSetDevice(device) != hipSuccess) return GHICP_ERR_NO_GPU;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GHICP_ERR_NO_GPU;
  ghicp_ctx* c = new ghicp_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->km_stats = getenv("GHICP_KM_STATS") != nullptr;
  c->km_force_hazard = getenv("GHICP_KM_FORCE_HAZARD") != nullptr;
  if (const char* e = getenv("GHICP_LOOP_SLOTS")) c->loop_slots_cap = atoi(e) > 0 ? atoi(e) : 0;
  if (const char* e = getenv("GHICP_LOOP_CONFINE")) c->loop_confine = atoi(e) != 0;
  if (const char* e = getenv("GHICP_LOOP_MIN_LDS")) c->loop_min_lds = atoi(e) > 0 ? atoi(e) : 0;
  if (const char* e = getenv("GHICP_LOOP_CONFINE_MARGIN")) { const double m = atof(e); if (m >= 0.25 && m <= 4.0) c->loop_confine_margin = m; }  // experiment hook: margin on the confined class's share
  if (hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault) != hipSuccess) { delete c; return GHICP_ERR_HIP; }
  c->pinned_cap = 4096;
  *out = c;
  return GHICP_OK;
}

extern "C" int ghicp_ctx_destroy(ghicp_ctx* ctx) {
  if (!ctx) return GHICP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  ctx->kt_collect();
  for (hipEvent_t e : ctx->kt_pool) (void)hipEventDestroy(e);
  for (ghicp_cloud* c : ctx->pair_clouds) (void)ghicp_cloud_destroy(c);
  ctx->pair_clouds.clear();
  for (int i = 0; i < B_NUM; i++) ctx->buf[i].release();
  for (auto& b : ctx->pairbuf) b.release();
  ctx->stage_clear();
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->job_pinned) (void)hipHostFree(ctx->job_pinned);
  if (ctx->job_event) (void)hipEventDestroy(ctx->job_event);
  if (ctx->fb_pinned) (void)hipHostFree(ctx->fb_pinned);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  for (auto& e : ctx->confine_cache) { (void)hipStreamDestroy(e.confined); (void)hipStreamDestroy(e.rest); }
  for (hipStream_t a : ctx->aux_streams) (void)hipStreamDestroy(a);
  for (hipEvent_t e : ctx->aux_events) (void)hipEventDestroy(e);
  if (ctx->progress_host) (void)hipHostFree(ctx->progress_host);
  delete ctx;
  return GHICP_OK;
}

extern "C" int ghicp_ctx_stage_stats(const ghicp_ctx* ctx, int64_t* hits, int64_t* misses, int64_t* bytes_kept) {
  if (!ctx) return GHICP_ERR_ARG;
  if (hits) *hits = ctx->staged_hits;
  if (misses) *misses = ctx->staged_misses;
  if (bytes_kept) *bytes_kept = (int64_t)ctx->staged_bytes;
  return GHICP_OK;
}

extern "C" int ghicp_ctx_set_stream(ghicp_ctx* ctx, void* s) {
  GH_ENTER(ctx);
  ctx->stream = reinterpret_cast<hipStream_t>(s);
  return GHICP_OK;
}
// Replaces the context's stream by one that may only use the compute units whose bit is set in `mask` (32 CUs per word,
// hipExtStreamCreateWithCUMask).  Used to keep a few CUs free of Kuhn-Munkres waves -- whose LDS footprint otherwise
// fills every CU -- so that the small front-end kernels of the next batch run next to a solve launch.
extern "C" int ghicp_ctx_set_cu_mask(ghicp_ctx* ctx, const uint32_t* mask, int32_t n_words) {
  GH_ENTER(ctx);
  GH_ARG(mask != nullptr && n_words >= 1 && n_words <= 64);
  bool any = false;
  for (int i = 0; i < n_words; i++) any = any || mask[i] != 0u;
  GH_ARG(any);
  hipStream_t s = nullptr;
  GH_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask));
  if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); (void)hipStreamDestroy(ctx->own_stream); }
  ctx->own_stream = s;
  ctx->stream = s;
  ctx->cu_mask.assign(mask, mask + n_words);
  for (hipStream_t a : ctx->aux_streams) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }  // they carry the previous mask
  ctx->aux_streams.clear();
  return GHICP_OK;
}

extern "C" int ghicp_ctx_set_host_pointers(ghicp_ctx* ctx, int on) {
  GH_ENTER(ctx);
  ctx->host_ptrs = on != 0;
  if (!ctx->host_ptrs) ctx->stage_clear();  // device-pointer mode stages nothing: the kept copies go with the mode
  return GHICP_OK;
}
extern "C" int ghicp_ctx_stage_clear(ghicp_ctx* ctx) {
  GH_ENTER(ctx);
  ctx->stage_clear();
  return GHICP_OK;
}
extern "C" int ghicp_ctx_synchronize(ghicp_ctx* ctx) {
  GH_ENTER(ctx);
  GH_HIP(hipStreamSynchronize(ctx->stream));
  return GHICP_OK;
}
extern "C" int ghicp_ctx_kernel_timing(ghicp_ctx* ctx, int on) {
  GH_ENTER(ctx);
  GH_HIP(hipStreamSynchronize(ctx->stream));
  ctx->kt_collect();
  ctx->kt_on = on != 0;
  for (int i = 0; i < KT_NUM; i++) { ctx->kt_ms[i] = 0; ctx->kt_count[i] = 0; }
  if (on) {  // Kuhn-Munkres launch records start over (they stay readable after timing is switched off)
    unsigned long long* lstat;
    GH_TRY(ctx->reserve(B_KM_LSTAT, (size_t)ghicp_ctx::KM_LSTAT_MAX * ghicp_ctx::KM_LSTAT_W, &lstat));
    GH_HIP(hipMemsetAsync(lstat, 0, (size_t)ghicp_ctx::KM_LSTAT_MAX * ghicp_ctx::KM_LSTAT_W * sizeof(unsigned long long), ctx->stream));
    ctx->km_launches = 0;
    ctx->km_slots.clear();
  }
  return GHICP_OK;
}
extern "C" int ghicp_ctx_loop_progress(const ghicp_ctx* ctx, int64_t* active, int64_t* total) {  // no device work: callable from any thread
  if (!ctx || !active || !total) return GHICP_ERR_ARG;
  *total = ctx->loop_total.load(std::memory_order_relaxed);
  if (ctx->progress_live.load(std::memory_order_acquire) && ctx->progress_host) {  // persistent pair loop: the device counts completed pairs
    const long long done = *(volatile int*)ctx->progress_host;
    *active = *total > done ? *total - done : 0;
  } else {
    *active = ctx->loop_active.load(std::memory_order_relaxed);
  }
  return GHICP_OK;
}
extern "C" int ghicp_ctx_loop_progress_reset(ghicp_ctx* ctx, int64_t total) {  // no device work
  if (!ctx || total < 0) return GHICP_ERR_ARG;
  ctx->progress_live.store(false, std::memory_order_release);
  ctx->loop_total.store(total, std::memory_order_relaxed);
  ctx->loop_active.store(total, std::memory_order_relaxed);
  return GHICP_OK;
}
extern "C" int ghicp_ctx_kernel_time(ghicp_ctx* ctx, const char* name, double* total_ms, int64_t* launches) {
  GH_ENTER(ctx);
  GH_ARG(name != nullptr);
  GH_HIP(hipStreamSynchronize(ctx->stream));
  ctx->kt_collect();
  for (int i = 0; i < KT_NUM; i++)
    if (strcmp(name, kKtNames[i]) == 0) {
      if (total_ms) *total_ms = ctx->kt_ms[i];
      if (launches) *launches = ctx->kt_count[i];
      return GHICP_OK;
    }
  return ctx->fail(GHICP_ERR_ARG, "unknown kernel name '%s'", name);
}
extern "C" int ghicp_ctx_set_loop_cost_hints(ghicp_ctx* ctx, int32_t n_pairs, const float* cost) {
  GH_ENTER(ctx);
  GH_ARG(n_pairs >= 0 && (n_pairs == 0 || cost != nullptr));
  ctx->loop_cost_hints.assign(cost, cost + n_pairs);
  for (float& c : ctx->loop_cost_hints)
    if (!(c == c)) c = 0.f;  // NaN would break the strict weak order of the sort
  return GHICP_OK;
}
extern "C" int ghicp_ctx_loop_timeline(ghicp_ctx* ctx, int64_t* out3, int64_t cap_pairs, int64_t* n_pairs) {
  GH_ENTER(ctx);
  GH_ARG(n_pairs != nullptr && cap_pairs >= 0);
  const int64_t n = (int64_t)(ctx->loop_timeline.size() / 3);
  *n_pairs = n;
  if (out3)
    for (int64_t i = 0; i < std::min(n, cap_pairs) * 3; i++) out3[i] = ctx->loop_timeline[(size_t)i];
  return GHICP_OK;
}
extern "C" const char* ghicp_last_error(const ghicp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

namespace {

// single-workgroup float-Umeyama (same arithmetic as the k_solve stage of loop.hip)
__global__ __launch_bounds__(1024) void k_rigid_svd(const double* __restrict__ src, const double* __restrict__ tgt, int c,
                                                    double* __restrict__ out16) {
  __shared__ double red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  double m[6] = {0, 0, 0, 0, 0, 0};
  for (int i = tid; i < c; i += nt)
    for (int d = 0; d < 3; d++) { m[d] += (double)(float)src[(size_t)i * 3 + d]; m[3 + d] += (double)(float)tgt[(size_t)i * 3 + d]; }
  for (int d = 0; d < 6; d++) m[d] = gh_block_sum(m[d], red);
  float msf[3], mtf[3];
  for (int d = 0; d < 3; d++) { msf[d] = (float)(m[d] / (double)c); mtf[d] = (float)(m[3 + d] / (double)c); }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = tid; i < c; i += nt) {
    double a[3], b[3];
    for (int d = 0; d < 3; d++) {
      a[d] = (double)(float)tgt[(size_t)i * 3 + d] - (double)mtf[d];
      b[d] = (double)(float)src[(size_t)i * 3 + d] - (double)msf[d];
    }
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) H[r * 3 + q] += a[r] * b[q];
  }
  for (int d = 0; d < 9; d++) H[d] = gh_block_sum(H[d], red);
  if (tid == 0) {
    double A[9], R[9];
    for (int d = 0; d < 9; d++) A[d] = H[d] / (double)c;
    gh_quant_grid(A, 9);  // N2: umeyama's sigma is a Matrix3f
    gh_kabsch(A, R);
    float Rf[9];
    for (int d = 0; d < 9; d++) Rf[d] = (float)R[d];
    for (int d = 0; d < 16; d++) out16[d] = 0;
    for (int r = 0; r < 3; r++) {
      const float t = (float)((double)mtf[r] - (((double)Rf[r * 3] * (double)msf[0] + (double)Rf[r * 3 + 1] * (double)msf[1]) +
                                                 (double)Rf[r * 3 + 2] * (double)msf[2]));
      for (int q = 0; q < 3; q++) out16[r * 4 + q] = (double)Rf[r * 3 + q];
      out16[r * 4 + 3] = (double)t;
    }
    out16[15] = 1;
  }
}

// pcl::transformPointCloud with a float 4x4 (test/ghicp_main.cpp:153): ((m0 x + m1 y) + m2 z) + m3
struct M34 { float m[12]; };
__global__ __launch_bounds__(256) void k_transform(const float* __restrict__ xyz, long long n, int stride, M34 M, float* __restrict__ out) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float x = xyz[i * stride], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
    out[i * 3 + 0] = ((M.m[0] * x + M.m[1] * y) + M.m[2] * z) + M.m[3];
    out[i * 3 + 1] = ((M.m[4] * x + M.m[5] * y) + M.m[6] * z) + M.m[7];
    out[i * 3 + 2] = ((M.m[8] * x + M.m[9] * y) + M.m[10] * z) + M.m[11];
  }
}

// S7 of a whole batch of pairs in one launch: job (blockIdx.y + job0) = one raw cloud under its pair's final transform.  Packed clouds
// (stride 3) move as float4s -- four points are three 16-byte loads and three 16-byte stores per lane --, the arithmetic per point is
// k_transform's, so the result is bit-identical to ghicp_transform_cloud.
struct TransformJob { const float* xyz; float* out; long long n; M34 M; };
__device__ __forceinline__ void transform_point(const M34& M, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = ((M.m[0] * x + M.m[1] * y) + M.m[2] * z) + M.m[3];
  oy = ((M.m[4] * x + M.m[5] * y) + M.m[6] * z) + M.m[7];
  oz = ((M.m[8] * x + M.m[9] * y) + M.m[10] * z) + M.m[11];
}
__global__ __launch_bounds__(256) void k_transform_batch(const TransformJob* __restrict__ jobs, int job0, int stride) {
  const TransformJob J = jobs[job0 + blockIdx.y];
  const long long n = J.n;
  if (stride == 3 && ((reinterpret_cast<uintptr_t>(J.xyz) | reinterpret_cast<uintptr_t>(J.out)) & 15) == 0) {
    const long long n4 = n >> 2;
    const float4* __restrict__ in4 = reinterpret_cast<const float4*>(J.xyz);
    float4* __restrict__ out4 = reinterpret_cast<float4*>(J.out);
    for (long long g = blockIdx.x * 256ll + threadIdx.x; g < n4; g += (long long)gridDim.x * 256) {
      const float4 a = in4[g * 3], b = in4[g * 3 + 1], c = in4[g * 3 + 2];
      float4 oa, ob, oc;
      transform_point(J.M, a.x, a.y, a.z, oa.x, oa.y, oa.z);
      transform_point(J.M, a.w, b.x, b.y, oa.w, ob.x, ob.y);
      transform_point(J.M, b.z, b.w, c.x, ob.z, ob.w, oc.x);
      transform_point(J.M, c.y, c.z, c.w, oc.y, oc.z, oc.w);
      out4[g * 3] = oa; out4[g * 3 + 1] = ob; out4[g * 3 + 2] = oc;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
      const long long i = (n4 << 2) + threadIdx.x;
      transform_point(J.M, J.xyz[i * 3], J.xyz[i * 3 + 1], J.xyz[i * 3 + 2], J.out[i * 3], J.out[i * 3 + 1], J.out[i * 3 + 2]);
    }
    return;
  }
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    transform_point(J.M, J.xyz[i * stride], J.xyz[i * stride + 1], J.xyz[i * stride + 2], J.out[i * 3], J.out[i * 3 + 1], J.out[i * 3 + 2]);
}

__global__ __launch_bounds__(256) void k_gather4(const float* __restrict__ xyz, int stride, const int* __restrict__ idx, long long m,
                                                 float4* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  const long long s = idx[i];
  out[i] = make_float4(xyz[s * stride], xyz[s * stride + 1], xyz[s * stride + 2], 0.f);
}

__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ xyz, long long n, int stride, float* __restrict__ mm /*6: min xyz, max xyz as ordered ints*/) {
  __shared__ float smin[3][4], smax[3][4];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    for (int d = 0; d < 3; d++) { const float v = xyz[i * stride + d]; mn[d] = fminf(mn[d], v); mx[d] = fmaxf(mx[d], v); }
  for (int d = 0; d < 3; d++) {
    for (int o = 32; o > 0; o >>= 1) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[d][threadIdx.x >> 6] = mn[d]; smax[d][threadIdx.x >> 6] = mx[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    float a = smin[d][0], b = smax[d][0];
    for (int w = 1; w < 4; w++) { a = fminf(a, smin[d][w]); b = fmaxf(b, smax[d][w]); }
    // float atomic min/max through the order-preserving int mapping
    auto enc = [](float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; };
    atomicMin(reinterpret_cast<int*>(mm) + d, enc(a));
    atomicMax(reinterpret_cast<int*>(mm) + 3 + d, enc(b));
  }
}

}  // namespace

__global__ void k_bbox_init(int* mm) {  // enc(+FLT_MAX) x 3, enc(-FLT_MAX) x 3
  if (threadIdx.x < 6) mm[threadIdx.x] = threadIdx.x < 3 ? 0x7f7fffff : (int)0x80800000;
}

// Bounding box of a device cloud on the host.  Small transfers go through the context's PINNED scratch: a pageable hipMemcpyAsync is
// staged and serialised inside the runtime (the front end did 12 of them per cloud from 16 threads).  The box of the cloud a front end
// is working on is computed once and remembered (ctx->bbox_*): the PCA grid, the BSC grid and the bbx magnitude all ask for it.
int gh_bbox_dev(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float* mm_host6) {
  if (ctx->bbox_valid && ctx->bbox_ptr == xyz && ctx->bbox_n == n && ctx->bbox_stride == stride) {
    memcpy(mm_host6, ctx->bbox_mm, 6 * sizeof(float));
    return GHICP_OK;
  }
  int* d;
  GH_TRY(ctx->reserve(B_GRID_MISC, 64, &d));
  hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(64), 0, ctx->stream, d);
  if (n > 0) hipLaunchKernelGGL(k_bbox, dim3(min(cdiv(n, 256), 2048)), dim3(256), 0, ctx->stream, xyz, n, stride, reinterpret_cast<float*>(d));
  int* h = reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->pinned) + 256);
  GH_HIP(hipMemcpyAsync(h, d, 6 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 6; k++) {
    int i = h[k] >= 0 ? h[k] : h[k] ^ 0x7fffffff;
    memcpy(&mm_host6[k], &i, 4);
  }
  return GHICP_OK;
}

extern "C" int ghicp_cloud_bounds(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, double* out6) {
  GH_ENTER(ctx);
  GH_ARG(out6 != nullptr && n > 0 && stride >= 3);
  Stager sg(ctx);
  const float* d;
  GH_TRY(sg.in_cloud(xyz, (size_t)n * stride, &d));
  float mm[6];
  GH_TRY(gh_bbox_dev(ctx, d, n, stride, mm));
  for (int k = 0; k < 6; k++) out6[k] = (double)mm[k];
  return GHICP_OK;
}

extern "C" int ghicp_bbx_magnitude(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, float* bbx) {
  GH_ENTER(ctx);
  GH_ARG(bbx != nullptr && n >= 0 && stride >= 3);
  Stager sg(ctx);
  const float* d;
  GH_TRY(sg.in_cloud(xyz, (size_t)n * stride, &d));
  if (n == 0) { *bbx = 0.f; return GHICP_OK; }
  float mm[6];
  GH_TRY(gh_bbox_dev(ctx, d, n, stride, mm));
  // Bounds are doubles holding float values; the sum is formed in double and stored to float (main:91-93)
  *bbx = (float)((double)mm[3] - (double)mm[0] + (double)mm[4] - (double)mm[1] + (double)mm[5] - (double)mm[2]);
  return GHICP_OK;
}

extern "C" int ghicp_rigid_svd(ghicp_ctx* ctx, const double* src, const double* tgt, int64_t c, double* Rt16) {
  GH_ENTER(ctx);
  GH_ARG(c > 0 && src && tgt && Rt16);
  Stager sg(ctx);
  const double *ds, *dt;
  GH_TRY(sg.in(src, (size_t)c * 3, &ds));
  GH_TRY(sg.in(tgt, (size_t)c * 3, &dt));
  double* out;
  GH_TRY(ctx->reserve(B_P_MISC, 16, &out));
  hipLaunchKernelGGL(k_rigid_svd, dim3(1), dim3(1024), 0, ctx->stream, ds, dt, (int)c, out);
  GH_HIP(hipGetLastError());
  GH_HIP(hipMemcpyAsync(Rt16, out, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(hipStreamSynchronize(ctx->stream));
  return GHICP_OK;
}

// Host-side float Umeyama with the arithmetic of k_rigid_svd (N1-N5: float means, f64 covariance rounded once onto the f32
// grid, Kabsch via Jacobi, R and t rounded to f32) for the few-point closed-form solvers of the reference's public API
// (CRegistration::SVD_6DOF, src/common_reg.cpp:774-888), which are not worth a kernel launch.  No context, no GPU.
extern "C" int ghicp_rigid_svd_host(const double* src, const double* tgt, int64_t c, double* out16) {
  if (!src || !tgt || !out16 || c < 1) return GHICP_ERR_ARG;
  double m[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t i = 0; i < c; i++)
    for (int d = 0; d < 3; d++) { m[d] += (double)(float)src[(size_t)i * 3 + d]; m[3 + d] += (double)(float)tgt[(size_t)i * 3 + d]; }
  float msf[3], mtf[3];
  for (int d = 0; d < 3; d++) { msf[d] = (float)(m[d] / (double)c); mtf[d] = (float)(m[3 + d] / (double)c); }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = 0; i < c; i++) {
    double a[3], b[3];
    for (int d = 0; d < 3; d++) {
      a[d] = (double)(float)tgt[(size_t)i * 3 + d] - (double)mtf[d];
      b[d] = (double)(float)src[(size_t)i * 3 + d] - (double)msf[d];
    }
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) H[r * 3 + q] += a[r] * b[q];
  }
  double A[9], R[9];
  for (int d = 0; d < 9; d++) A[d] = H[d] / (double)c;
  gh_quant_grid(A, 9);
  gh_kabsch(A, R);
  float Rf[9];
  for (int d = 0; d < 9; d++) Rf[d] = (float)R[d];
  for (int d = 0; d < 16; d++) out16[d] = 0;
  for (int r = 0; r < 3; r++) {
    const float t = (float)((double)mtf[r] - (((double)Rf[r * 3] * (double)msf[0] + (double)Rf[r * 3 + 1] * (double)msf[1]) +
                                               (double)Rf[r * 3 + 2] * (double)msf[2]));
    for (int q = 0; q < 3; q++) out16[r * 4 + q] = (double)Rf[r * 3 + q];
    out16[r * 4 + 3] = (double)t;
  }
  out16[15] = 1;
  return GHICP_OK;
}

extern "C" int ghicp_transform_cloud(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, const double* Rt, float* out) {
  GH_ENTER(ctx);
  GH_ARG(n >= 0 && stride >= 3 && Rt != nullptr);
  Stager sg(ctx);
  const float* d;
  float* o;
  GH_TRY(sg.in_cloud(xyz, (size_t)n * stride, &d));
  GH_TRY(sg.out(out, (size_t)n * 3, &o));
  M34 M;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) M.m[r * 4 + c] = (float)Rt[r * 4 + c];
  if (n > 0) {
    hipLaunchKernelGGL(k_transform, dim3(min(cdiv(n, 256), 4096)), dim3(256), 0, ctx->stream, d, (long long)n, stride, M, o);
    GH_HIP(hipGetLastError());
  }
  return sg.finish();
}

extern "C" int ghicp_transform_clouds(ghicp_ctx* ctx, int32_t n_clouds, const float* const* xyz, const int64_t* n, int stride, const double* Rt16,
                                      float* const* out) {
  GH_ENTER(ctx);
  GH_ARG(n_clouds >= 0 && stride >= 3 && (n_clouds == 0 || (xyz && n && Rt16 && out)));
  if (ctx->host_ptrs) return ctx->fail(GHICP_ERR_ARG, "ghicp_transform_clouds: device-pointer mode only (use ghicp_transform_cloud per cloud)");
  if (n_clouds == 0) return GHICP_OK;
  std::vector<TransformJob> h((size_t)n_clouds);
  int64_t nmax = 0;
  for (int i = 0; i < n_clouds; i++) {
    GH_ARG(n[i] >= 0 && (n[i] == 0 || (xyz[i] && out[i])));
    h[i].xyz = xyz[i]; h[i].out = out[i]; h[i].n = n[i];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) h[i].M.m[r * 4 + c] = (float)Rt16[(size_t)i * 16 + r * 4 + c];  // Rt_final.cast<float>() (main:153)
    nmax = std::max<int64_t>(nmax, n[i]);
  }
  TransformJob* d;
  GH_TRY(ctx->reserve(B_TRANSFORM_JOBS, h.size(), &d));
  GH_TRY(ctx->upload_table(h.data(), h.size() * sizeof(TransformJob), d));  // through the pinned job buffer: no stream synchronisation
  if (nmax > 0) {
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(nmax, 256 * 8), 2048));
    hipEvent_t kev = ctx->kt_begin(KT_TRANSFORM);
    for (int j0 = 0; j0 < n_clouds; j0 += 65535)
      hipLaunchKernelGGL(k_transform_batch, dim3(bx, std::min(65535, n_clouds - j0)), dim3(256), 0, ctx->stream, (const TransformJob*)d, j0, stride);
    ctx->kt_end(KT_TRANSFORM, kev);
    GH_HIP(hipGetLastError());
  }
  return GHICP_OK;
}

extern "C" int ghicp_gather_points(ghicp_ctx* ctx, const float* xyz, int stride, const int32_t* idx, int64_t m, float* out) {
  GH_ENTER(ctx);
  GH_ARG(m >= 0 && stride >= 3);
  if (ctx->host_ptrs) return ctx->fail(GHICP_ERR_ARG, "ghicp_gather_points: device-pointer mode only");
  if (m > 0) {
    hipLaunchKernelGGL(k_gather4, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, xyz, stride, idx, (long long)m, reinterpret_cast<float4*>(out));
    GH_HIP(hipGetLastError());
  }
  return GHICP_OK;
}
