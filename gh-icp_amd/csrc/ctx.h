// Internal: context, device buffers, host/device staging.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ghicp_c.h"

#define GH_HIP(call)                                                                                          \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) return ctx->fail(GHICP_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)
#define GH_TRY(call)               \
  do {                             \
    int r_ = (call);               \
    if (r_ != GHICP_OK) return r_; \
  } while (0)
#define GH_ARG(cond)                                                                \
  do {                                                                              \
    if (!(cond)) return ctx->fail(GHICP_ERR_ARG, "%s: bad argument (%s)", __func__, #cond); \
  } while (0)

// First statement of every ABI entry that takes a context: HIP's current device is per THREAD and starts at 0, so a context
// driven from a worker thread (bench.py, any thread pool of a caller) must select its own device before it allocates or launches.
#define GH_ENTER(ctx)                                                                                             \
  do {                                                                                                            \
    if (!(ctx)) return GHICP_ERR_ARG;                                                                             \
    if (hipSetDevice((ctx)->device) != hipSuccess) return (ctx)->fail(GHICP_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->device); \
  } while (0)

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      if (e != hipSuccess) return e;
      p = nullptr; cap = 0;
    }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

enum BufSlot {
  // staging (host-pointer mode)
  B_STAGE0 = 0, B_STAGE1, B_STAGE2, B_STAGE3, B_STAGE4, B_STAGE5, B_STAGE6, B_STAGE7,
  // loop
  B_LOOP_STATE, B_LOOP_KPS, B_LOOP_PARTMIN, B_LOOP_PARTIDX, B_LOOP_PARTSUM, B_LOOP_PARTMIN2, B_LOOP_PARTIDX2, B_LOOP_FDT,
  B_LOOP_SP, B_LOOP_TP, B_LOOP_TRACE, B_LOOP_WFD, B_LOOP_KMW, B_LOOP_KMMATCH, B_LOOP_KMSCR, B_LOOP_ACC,
  // spatial grid / sort
  B_GRID_KEYS, B_GRID_KEYS2, B_GRID_VALS, B_GRID_VALS2, B_GRID_TMP, B_GRID_START, B_GRID_PTS, B_GRID_MISC,
  B_GRID2_KEYS, B_GRID2_KEYS2, B_GRID2_VALS, B_GRID2_VALS2, B_GRID2_START, B_GRID2_PTS,
  // front end
  B_FE_LAMBDA, B_FE_CURV, B_FE_COUNT, B_FE_CAND, B_FE_STATE, B_FE_KP, B_FE_SORTK, B_FE_SORTK2, B_FE_SORTV, B_FE_SORTV2, B_FE_CPTS,
  B_FE_FLAGS, B_FE_SCAN, B_FE_SCATTER,
  // pair pipeline
  B_P_KEEP_S, B_P_KEEP_T, B_P_DS_S, B_P_DS_T, B_P_KP_S, B_P_KP_T, B_P_KPXYZ_S, B_P_KPXYZ_T, B_P_FEAT_S, B_P_FEAT_T, B_P_LCS,
  B_P_FD, B_P_MISC, B_P_PATTERN, B_FD_JOBS, B_TRANSFORM_JOBS,
  B_KM_LX, B_KM_MISC, B_KM_SLACK, B_KM_LSTAT, B_KM_ORDER,
  // batched front end (batch.hip)
  B_FB_DESC, B_FB_HEADPOS, B_FB_DS, B_FB_ORD,
  // hand-written scan / select primitives (prims.hip): tile totals; round-based NMS of the batched front end (batch.hip)
  B_PRIM_TMP, B_NMSR_KEY, B_NMSR_CELL, B_NMSR_TABLE, B_NMSR_HEAD, B_NMSR_PTS, B_NMSR_SKEY, B_NMSR_STATE, B_NMSR_NEXT, B_NMSR_SEL, B_NMSR_MISC, B_NMSR_LIST, B_NMSR_PTS0, B_NMSR_SKEY0,
  // fine registration (icp.hip): coarse target grid, source grids (reciprocal), per-point state
  B_ICP_TC_KEYS, B_ICP_TC_KEYS2, B_ICP_TC_VALS, B_ICP_TC_VALS2, B_ICP_TC_START, B_ICP_TC_PTS,
  B_ICP_SC_KEYS, B_ICP_SC_KEYS2, B_ICP_SC_VALS, B_ICP_SC_VALS2, B_ICP_SC_START, B_ICP_SC_PTS,
  B_ICP_CUR, B_ICP_Q, B_ICP_NN, B_ICP_ND, B_ICP_NN2, B_ICP_ND2, B_ICP_KEYS, B_ICP_KEYS2, B_ICP_SORTTMP, B_ICP_PART, B_ICP_STATE,
  B_ICP_PEND, B_ICP_TNRM, B_ICP_OUT,
  B_NUM
};

enum KtSlot { KT_PCA = 0, KT_BSC, KT_KM_SOLVE, KT_CD_ROWMIN, KT_KM_WEIGHTS, KT_FD_BSC, KT_NMS_ROUND, KT_VOXEL_SORT,
              KT_FB_VOXEL, KT_FB_GRID, KT_FB_PRUNE, KT_FB_RANK, KT_FB_OUT,  // stages of the batched front end (batch.hip) around the kernels above
              KT_PAIR_LOOP,                                                  // the persistent pair loop (loop.hip): all classes of a batch, fork -> join
              KT_TRANSFORM,                                                  // S7 of a batch (ghicp_transform_clouds)
              KT_PAIR_LOOP_DISPATCH,                                         // ONE k_pair_loop dispatch (a class launch of a batch), timed on the stream it runs on -- what rocprofv3's kernel trace reports per row
              KT_NUM };
static const char* const kKtNames[KT_NUM] = {"pca_cells", "bsc", "km_solve", "cd_rowmin", "km_weights", "fd_bsc", "nms_round", "voxel_sort",
                                             "fb_voxel", "fb_grid", "fb_prune", "fb_rank", "fb_out", "pair_loop", "transform", "pair_loop_dispatch"};

struct ghicp_ctx {
  // optional per-kernel timing
  bool kt_on = false;
  double kt_ms[KT_NUM] = {0};
  long long kt_count[KT_NUM] = {0};
  struct KtPending { int slot; hipEvent_t a, b; };
  std::vector<KtPending> kt_pending;
  std::vector<hipEvent_t> kt_pool;
  hipEvent_t kt_event() {
    if (!kt_pool.empty()) { hipEvent_t e = kt_pool.back(); kt_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  hipEvent_t kt_begin(int slot) { return kt_begin_on(slot, stream); }
  void kt_end(int slot, hipEvent_t a) { kt_end_on(slot, a, stream); }
  // the same bracket on ANOTHER stream (the class streams of the persistent pair loop): an event only sees the stream it is recorded on
  hipEvent_t kt_begin_on(int slot, hipStream_t st) {
    if (!kt_on) return nullptr;
    hipEvent_t a = kt_event();
    (void)hipEventRecord(a, st);
    (void)slot;
    return a;
  }
  void kt_end_on(int slot, hipEvent_t a, hipStream_t st) {
    if (!kt_on || !a) return;
    hipEvent_t b = kt_event();
    (void)hipEventRecord(b, st);
    kt_pending.push_back({slot, a, b});
  }
  void kt_collect() {  // stream must be idle
    for (auto& p : kt_pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { kt_ms[p.slot] += ms; kt_count[p.slot]++; }
      kt_pool.push_back(p.a);
      kt_pool.push_back(p.b);
    }
    kt_pending.clear();
  }
  // per-launch statistics of the Kuhn-Munkres solve launches (collected while kernel timing is on): device records of
  // KM_LSTAT_MAX launches x KM_LSTAT_W words, and the solve slots (resident workgroups) each launch had.  Words: first start, last end,
  // sum and max of the solve times, solves; persistent pair loop only: sum of the slot lifetimes, slots that ran
  static constexpr int KM_LSTAT_MAX = 8192;
  static constexpr int KM_LSTAT_W = 8;
  long long km_launches = 0;
  std::vector<int> km_slots;
  long long loop_hazards = 0;            // Kuhn-Munkres solves of this context's batched loops that took the literal fallback of rule R4 (ghicp_ctx_loop_hazards)
  std::vector<float> loop_cost_hints;    // ghicp_ctx_set_loop_cost_hints: queue order of the next persistent batch of exactly this many pairs
  std::vector<long long> loop_timeline;  // last persistent batch with kernel timing on: per pair (begin, end) in 100 MHz ticks and iterations
  // progress of the batched loop that is running on this context (pairs still iterating / pairs of the batch), readable from
  // other threads while ghicp_register_pairs / ghicp_register_clouds is in flight (ghicp_ctx_loop_progress)
  std::atomic<long long> loop_active{0}, loop_total{0};
  // bounding box of the cloud a front end is working on (set and cleared by cloud_fill; see gh_bbox_dev)
  bool bbox_valid = false;
  const float* bbox_ptr = nullptr;
  long long bbox_n = 0;
  int bbox_stride = 0;
  float bbox_mm[6] = {0, 0, 0, 0, 0, 0};
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;  // created by ghicp_ctx_set_cu_mask
  // persistent pair loop: one launch per LDS-occupancy class, concurrently, on these streams (forked from / joined into `stream`)
  // diagnostics, read from the environment ONCE when the context is created (never on a launch path): GHICP_KM_STATS=1 runs the
  // profiled Kuhn-Munkres kernel and prints its stage counters; GHICP_KM_FORCE_HAZARD=1 is the test hook that sends one phase of
  // every solve through the literal fallback of rule R4; GHICP_LOOP_SLOTS=<n> is the test hook that caps the workgroups (solve slots) of
  // every class launch of the persistent pair loop, so that a few slots take MANY pairs one after the other (state carried over in LDS)
  bool km_stats = false, km_force_hazard = false;
  int loop_slots_cap = 0;
  // The class of graphs that fit only three per CU is CONFINED to a few CUs (a stream with a CU mask, made when first needed): a 50 KB
  // slot between 40 KB slots leaves LDS holes no 40 KB slot fits, and a persistent slot never leaves -- a CU that has held one big slot
  // stays at three slots for the rest of the batch (round 5, call 5: 811 of 1024 slots busy from second 3.5 to the end of an 11 s batch).
  // GHICP_LOOP_CONFINE=0 switches it off (A/B measurements).
  bool loop_confine = true;
  double loop_confine_margin = 1.15;  // on the confined class's share of the cost prior (GHICP_LOOP_CONFINE_MARGIN: experiments)
  hipStream_t confine_stream = nullptr, rest_stream = nullptr;  // masks: the confined class's CUs / all the others (the pair in use, owned by confine_cache)
  int confine_cus = 0;
  struct ConfinePair { int cus; hipStream_t confined, rest; };
  std::vector<ConfinePair> confine_cache;  // one masked stream pair per share B seen (loop.hip run_pair_loop)
  int loop_min_lds = 0;  // GHICP_LOOP_MIN_LDS=<bytes> (experiment hook): every solve slot asks for at least this much LDS, e.g. 46080 = three slots per CU with 25 KB of every CU left to other kernels
  std::vector<uint32_t> cu_mask;       // set by ghicp_ctx_set_cu_mask: the auxiliary streams are restricted to the same compute units
  std::vector<hipStream_t> aux_streams;
  std::vector<hipEvent_t> aux_events;  // [0] fork, [1 + c] join of class c
  int* progress_host = nullptr;        // mapped pinned counter: pairs completed by the running persistent loop
  std::atomic<bool> progress_live{false};
  bool host_ptrs = false;
  // host-pointer mode: staged copies of LARGE input arrays, kept between ABI calls so that the reference's call sequence -- voxelfilter ->
  // keypointDetection -> extractBinaryFeatures on the SAME cloud (test/ghicp_main.cpp:86-116) -- uploads a cloud once.  Keyed on host address +
  // size + a 64-bit fingerprint of the CONTENT (a caller may change a cloud in place between calls), least recently used out first.
  struct Staged { const void* host; size_t bytes; uint64_t fp; void* dev; uint64_t tick; };
  std::vector<Staged> staged;
  size_t staged_bytes = 0;
  uint64_t stage_tick = 0;
  long long staged_hits = 0, staged_misses = 0;
  void stage_clear() {  // ghicp_ctx_stage_clear, ghicp_ctx_set_host_pointers(ctx, 0), ghicp_ctx_destroy
    if (staged.empty()) return;
    (void)hipStreamSynchronize(stream);
    for (auto& e : staged) (void)hipFree(e.dev);
    staged.clear();
    staged_bytes = 0;
  }
  static constexpr size_t STAGED_MIN = 256 * 1024, STAGED_CAP = (size_t)1 << 30;
  std::string err;
  DevBuf buf[B_NUM];
  std::vector<DevBuf> pairbuf;  // per-pair outputs of the front end (batched API): 3 per pair slot
  void* pinned = nullptr;  // small pinned host scratch
  // job tables of the batched launches (feature distances, final transforms): staged through a grow-only pinned buffer whose reuse waits
  // for the event behind the previous copy -- no stream synchronisation on the launch path (round-4 advisor: a pageable std::vector +
  // hipStreamSynchronize stalled the loop stream once per batch)
  void* job_pinned = nullptr;
  size_t job_pinned_cap = 0;
  hipEvent_t job_event = nullptr;
  bool job_pending = false;
  int upload_table(const void* src, size_t bytes, void* dst) {
    if (job_pending) {  // the previous table has left the pinned buffer?
      if (hipEventSynchronize(job_event) != hipSuccess) return fail(GHICP_ERR_HIP, "job table: event wait failed");
      job_pending = false;
    }
    if (bytes > job_pinned_cap) {
      if (job_pinned) (void)hipHostFree(job_pinned);
      job_pinned = nullptr;
      job_pinned_cap = 0;
      const size_t cap = ((bytes * 3 / 2) + 4095) & ~(size_t)4095;
      if (hipHostMalloc(&job_pinned, cap, hipHostMallocDefault) != hipSuccess) return fail(GHICP_ERR_HIP, "job table: pinned allocation of %zu bytes failed", cap);
      job_pinned_cap = cap;
    }
    if (!job_event && hipEventCreateWithFlags(&job_event, hipEventDisableTiming) != hipSuccess) return fail(GHICP_ERR_HIP, "job table: event creation failed");
    memcpy(job_pinned, src, bytes);
    if (hipMemcpyAsync(dst, job_pinned, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return fail(GHICP_ERR_HIP, "job table: upload failed");
    if (hipEventRecord(job_event, stream) != hipSuccess) return fail(GHICP_ERR_HIP, "job table: event record failed");
    job_pending = true;
    return GHICP_OK;
  }
  void* fb_pinned = nullptr;  // descriptor block + report of the batched front end (batch.hip)
  std::vector<struct ghicp_cloud*> pair_clouds;  // cloud handles behind ghicp_register_pairs (2 per pair slot; cloud.hip)
  size_t pinned_cap = 0;
  int num_cu = 256;

  int fail(int code, const char* fmt, ...) {
    char tmp[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    err = tmp;
    return code;
  }
  template <typename T> int reserve(BufSlot s, size_t count, T** out) {
    hipError_t e = buf[s].reserve(count * sizeof(T) + 16);
    if (e != hipSuccess) return fail(GHICP_ERR_HIP, "hipMalloc(%zu bytes, slot %d): %s", count * sizeof(T), (int)s, hipGetErrorString(e));
    *out = buf[s].as<T>();
    return GHICP_OK;
  }
};

// Stages ABI arrays when the context is in host-pointer mode; a no-op in device-pointer mode.
struct Stager {
  ghicp_ctx* ctx;
  struct Out { void* host; void* dev; size_t bytes; };
  std::vector<Out> outs;
  std::vector<void*> temps;
  bool uploaded = false;  // this call put an upload of a kept (cached) copy on the stream
  explicit Stager(ghicp_ctx* c) : ctx(c) { ctx->stage_tick++; }
  // hipFree of a temporary waits for the device; a call that staged only KEPT copies and no outputs has nothing that does (round-5 advisor):
  // the stream is joined here so that the caller may reuse or free its array as soon as the call is back
  ~Stager() {
    if (uploaded && temps.empty()) (void)hipStreamSynchronize(ctx->stream);
    for (void* p : temps) (void)hipFree(p);
  }
  // content fingerprint of a host array: four independent multiply-xor lanes over 8-byte words (memory bound, ~10 GB/s per core); the
  // words are read with memcpy (a cloud's rows are 4-byte aligned at best)
  static uint64_t fingerprint(const void* p, size_t bytes) {
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    const size_t nw = bytes / 8;
    uint64_t h0 = 0x9E3779B97F4A7C15ull, h1 = 0xC2B2AE3D27D4EB4Full, h2 = 0x165667B19E3779F9ull, h3 = 0x27D4EB2F165667C5ull;
    size_t i = 0;
    for (; i + 4 <= nw; i += 4) {
      uint64_t w[4];
      memcpy(w, b + i * 8, 32);
      h0 = (h0 ^ w[0]) * 0x100000001B3ull; h1 = (h1 ^ w[1]) * 0x100000001B3ull;
      h2 = (h2 ^ w[2]) * 0x100000001B3ull; h3 = (h3 ^ w[3]) * 0x100000001B3ull;
      h0 ^= h0 >> 29; h1 ^= h1 >> 31; h2 ^= h2 >> 27; h3 ^= h3 >> 33;
    }
    for (; i < nw; i++) { uint64_t w; memcpy(&w, b + i * 8, 8); h0 = (h0 ^ w) * 0x100000001B3ull; }
    for (size_t k = nw * 8; k < bytes; k++) h1 = (h1 ^ b[k]) * 0x100000001B3ull;
    return (h0 ^ (h1 << 1) ^ (h2 << 2) ^ (h3 << 3)) + bytes;
  }
  // in_cloud: a POINT CLOUD argument (xyz rows) -- the only kind of input the reference's call sequence hands over again and again; kept in
  // the staged-input cache.  in: every other input (feature matrices, Kuhn-Munkres weights, index lists): staged for this call only, never
  // fingerprinted (round-5 advisor: every input above 256 KB used to be hashed and kept).
  template <typename T> int in_cloud(const T* p, size_t count, const T** out) { return stage_in(p, count, out, true); }
  template <typename T> int in(const T* p, size_t count, const T** out) { return stage_in(p, count, out, false); }
  template <typename T> int stage_in(const T* p, size_t count, const T** out, bool cacheable) {
    if (!ctx->host_ptrs || p == nullptr) { *out = p; return GHICP_OK; }
    const size_t bytes = count * sizeof(T);
    if (cacheable && bytes >= ghicp_ctx::STAGED_MIN && bytes <= ghicp_ctx::STAGED_CAP / 4) {  // large cloud: look for the copy an earlier call staged
      const uint64_t fp = fingerprint(p, bytes);
      for (auto& e : ctx->staged)
        if (e.host == p && e.bytes == bytes && e.fp == fp) {
          e.tick = ctx->stage_tick;
          ctx->staged_hits++;
          *out = reinterpret_cast<const T*>(e.dev);
          return GHICP_OK;
        }
      while (!ctx->staged.empty() && (ctx->staged_bytes + bytes > ghicp_ctx::STAGED_CAP || ctx->staged.size() >= 16)) {  // LRU out; never an entry this call uses
        size_t lru = ctx->staged.size();
        for (size_t i = 0; i < ctx->staged.size(); i++)
          if (ctx->staged[i].tick != ctx->stage_tick && (lru == ctx->staged.size() || ctx->staged[i].tick < ctx->staged[lru].tick)) lru = i;
        if (lru == ctx->staged.size()) break;
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->staged[lru].dev);
        ctx->staged_bytes -= ctx->staged[lru].bytes;
        ctx->staged.erase(ctx->staged.begin() + (long)lru);
      }
      void* d = nullptr;
      if (hipMalloc(&d, bytes + 16) != hipSuccess) return ctx->fail(GHICP_ERR_HIP, "stage-in hipMalloc failed");
      if (hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { (void)hipFree(d); return ctx->fail(GHICP_ERR_HIP, "stage-in copy failed"); }
      ctx->staged.push_back({p, bytes, fp, d, ctx->stage_tick});
      ctx->staged_bytes += bytes;
      ctx->staged_misses++;
      uploaded = true;  // the copy out of the caller's buffer must have left it when the ABI call returns (~Stager)
      *out = reinterpret_cast<const T*>(d);
      return GHICP_OK;
    }
    void* d = nullptr;
    if (hipMalloc(&d, count * sizeof(T) + 16) != hipSuccess) return ctx->fail(GHICP_ERR_HIP, "stage-in hipMalloc failed");
    temps.push_back(d);
    if (hipMemcpyAsync(d, p, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return ctx->fail(GHICP_ERR_HIP, "stage-in copy failed");
    *out = reinterpret_cast<const T*>(d);
    return GHICP_OK;
  }
  template <typename T> int out(T* p, size_t count, T** outp) {
    if (!ctx->host_ptrs || p == nullptr) { *outp = p; return GHICP_OK; }
    void* d = nullptr;
    if (hipMalloc(&d, count * sizeof(T) + 16) != hipSuccess) return ctx->fail(GHICP_ERR_HIP, "stage-out hipMalloc failed");
    temps.push_back(d);
    outs.push_back({p, d, count * sizeof(T)});
    *outp = reinterpret_cast<T*>(d);
    return GHICP_OK;
  }
  // copy outputs back (all, or only the first `bytes` of the one registered for host pointer `p`)
  int finish() {
    for (auto& o : outs)
      if (hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        return ctx->fail(GHICP_ERR_HIP, "stage-out copy failed");
    if (!outs.empty() && hipStreamSynchronize(ctx->stream) != hipSuccess) return ctx->fail(GHICP_ERR_HIP, "stage-out sync failed");
    outs.clear();
    return GHICP_OK;
  }
};

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
__host__ __device__ inline int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// one job of a batched GH-ICP loop (all pointers device memory except Rt16 / trace / n_iter / converged: host)
struct gh_loop_job {
  const ghicp_params* p;
  const double* kpS;
  int ks;
  const double* kpT;
  int kt;
  const void* FD;
  const void* FDt;     // optional: the transposed feature-distance matrix [kt][ks], when the caller already holds it (gh_fd_bsc_batch_dev)
  double* Rt16;
  ghicp_iter* trace;
  int32_t* n_iter;
  int32_t* converged;
  int32_t* matchlist;
  double* rmse_after;  // host, optional: RMSEafter of the last iteration (ghicp_reg.cpp:905, the value behind "Registration Succeed.")
  // resumed loops (ghicp_iterate): scalar loop state in / out (host, opaque LoopState of loop.hip), the moved source keypoints out (device),
  // the last iteration's record (host), and the iteration whose matches go to matchlist row 0
  const void* resume_in;
  void* resume_out;
  double* kpS_out;
  ghicp_iter* trace_last;
  int ml_row0;
};
int gh_register_batch_dev(ghicp_ctx* ctx, int nb, const gh_loop_job* jobs);
// the reference's own verdict at convergence (src/ghicp_reg.cpp:918-924): "Registration Succeed." iff RMSEafter < 1.5 * nonmax
// (`nonmax` is the double copy of the float ctor argument, ghicp_reg.h:91,187); a loop stopped by the max_iter guard printed neither line
static inline int gh_registered_ok(int converged, double rmse_after, float radius_nonmax) {
  return (converged && rmse_after < 1.5 * (double)radius_nonmax) ? 1 : 0;
}

// ---- internal (device-pointer) entry points shared between translation units
int gh_fd_bsc_dev(ghicp_ctx* ctx, const uint8_t* featS, int ks, int V, const uint8_t* featT, int kt, uint16_t* FD);
struct gh_fd_bsc_job { const uint8_t* featS; const uint8_t* featT; uint16_t* FD; uint16_t* FDt; int ks, kt, V; };  // FDt may be null
int gh_fd_bsc_batch_dev(ghicp_ctx* ctx, int nb, const gh_fd_bsc_job* jobs);  // one launch for every pair of a batch (fd.hip)
int gh_register_dev(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS, int ks, const double* kpT, int kt, const void* FD,
                    double* Rt16, ghicp_iter* trace, int32_t* n_iter, int32_t* matchlist);
int gh_knn_normals_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, int k, float* normals);
int gh_km_solve_dev(ghicp_ctx* ctx, const double* w, int n, double eps, int32_t* match, const int* done_flag);
