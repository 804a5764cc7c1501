// Kuhn-Munkres, fourth-generation kernel (gfx950): host side and the stand-alone solve kernel.  The solver itself is device code in
// km4_dev.h (rules R1-R5 in its header), shared with the persistent pair loop of loop.hip.
#include "km4_dev.h"

#include <algorithm>
#include <vector>
#include <cstdlib>

namespace {

template <bool PROF>
__global__ __launch_bounds__(K4_T) void k_km4(const Km2Problem* __restrict__ probs, int flags, int lds_bytes, unsigned long long* __restrict__ lstat,
                                            const int* __restrict__ order) {
  const Km2Problem P = probs[order ? order[blockIdx.x] : (int)blockIdx.x];
  if (P.n <= 0 || (P.done && *P.done)) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  k4_solve_block<PROF>(P, flags, smem, lds_bytes, lstat);
}

}  // namespace

// lx, ly, slack (f64) + 16 f64 of reduction scratch + 8 bitsets + SH_NUM ints + match, two stacks, listed columns and CSR offsets (u16)
// + list lengths (u8): 44 B per row
size_t gh_km4_lds_bytes(int n) {
  const size_t nw = (size_t)(n + 31) / 32;
  return (size_t)n * 3 * 8 + 16 * 8 + 8 * nw * 4 + SH_NUM * 4 + ((size_t)n * (1 + 2 * K4_CAP) + 2 * ((size_t)n + 2)) * 2 + (size_t)n + 64;
}

bool gh_km4_fits(int n) { return n <= 65534 && gh_km4_lds_bytes(n) <= 160 * 1024 - 256; }

static int k4_launch(ghicp_ctx* ctx, const Km2Problem* d_probs, int nprob, size_t lds, const int* d_order) {
  const size_t want = 160 * 1024;
  // per device and thread safe: the attribute is cheap to set, so it is simply set before every launch
  GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km4<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
  GH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km4<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
  const int kflags = ctx->km_force_hazard ? 4 : 0;  // test hook: sends one phase through the hazard fallback
  unsigned long long* lstat = nullptr;
  if (ctx->kt_on && ctx->km_launches < ghicp_ctx::KM_LSTAT_MAX) {
    GH_TRY(ctx->reserve(B_KM_LSTAT, (size_t)ghicp_ctx::KM_LSTAT_MAX * ghicp_ctx::KM_LSTAT_W, &lstat));
    lstat += ctx->km_launches * ghicp_ctx::KM_LSTAT_W;
    int per_cu = 0;
    GH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_km4<false>), K4_T, lds));
    ctx->km_slots.push_back(per_cu * ctx->num_cu);
    ctx->km_launches++;
  }
  hipEvent_t kt = ctx->kt_begin(KT_KM_SOLVE);
  if (ctx->km_stats) hipLaunchKernelGGL((k_km4<true>), dim3(nprob), dim3(K4_T), lds, ctx->stream, d_probs, kflags, (int)lds, lstat, d_order);
  else hipLaunchKernelGGL((k_km4<false>), dim3(nprob), dim3(K4_T), lds, ctx->stream, d_probs, kflags, (int)lds, lstat, d_order);
  ctx->kt_end(KT_KM_SOLVE, kt);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

int gh_km4_launch(ghicp_ctx* ctx, const Km2Problem* d_probs, int nprob, int n_max) {
  return k4_launch(ctx, d_probs, nprob, gh_km4_lds_bytes(n_max), nullptr);
}

int gh_km4_plan(ghicp_ctx* ctx, const int* h_n, int nprob, Km4Plan* plan, const float* cost) {
  *plan = Km4Plan();
  std::vector<std::pair<int, int>> key((size_t)nprob);  // (problems per CU, n) per problem
  for (int i = 0; i < nprob; i++) {
    const int n = std::max(1, h_n[i]);
    if (!gh_km4_fits(n)) return ctx->fail(GHICP_ERR_INTERNAL, "gh_km4_plan: n = %d does not fit", n);
    // problems per CU by LDS, capped at FOUR: a solve slot is a 256-thread workgroup at 128 VGPRs, so a CU never holds more than four
    // whatever their LDS.  Rounds 2-4 capped at eight: the graphs below n = 745 then formed four more classes, each with its own launch
    // and queue, the classes took the chip in launch order (a persistent workgroup leaves only when ITS queue is dry), and the longest
    // pairs of the later classes -- the cost order is per class -- started 5-6 s into an 11 s batch: 798 pairs running with every queue
    // dry, a quarter of the slot-time idle (profiles/r04_slot_timeline_call4_hints.json).  One class for every graph that fits four
    // per CU = one queue in cost order = longest-processing-time-first over (almost) the whole batch.
    // (measured and dropped, call 8: leaving 4 KB of the 160 out of the count -- n <= 902 instead of <= 924 in the four-per-CU class --
    // did not bring its CUs to four slots each: 3.3 per CU, 384 pairs/s; the rule of call 6 stays)
    key[i] = {(int)std::min<size_t>(4, (160 * 1024) / gh_km4_lds_bytes(n)), n};
  }
  std::vector<int> order((size_t)nprob);
  for (int i = 0; i < nprob; i++) order[i] = i;
  // fewest per CU (largest problems) first; within a class the largest first -- or, with cost hints, the costliest first: the span of a
  // batch is bounded below by its slowest pair (112 iterations x ~50 ms on the bench scenes), so that pair must not start in the middle
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (key[a].first != key[b].first) return key[a].first < key[b].first;
    if (cost && cost[a] != cost[b]) return cost[a] > cost[b];
    return key[a].second > key[b].second;
  });
  int nc = 0;
  for (int i = 0; i < nprob;) {
    int j = i;
    while (j < nprob && key[order[j]].first == key[order[i]].first) j++;
    if (nc == 8) { plan->count[7] += nprob - i; break; }  // cannot happen: at most 8 occupancy classes
    plan->begin[nc] = i; plan->count[nc] = j - i;
    int nmax = 1;  // (with cost hints the first problem of a class is its costliest, not its largest)
    for (int t = i; t < j; t++) nmax = std::max(nmax, key[order[t]].second);
    plan->lds[nc] = gh_km4_lds_bytes(nmax);
    plan->per_cu[nc] = key[order[i]].first;
    for (int t = i; t < j; t++) plan->weight[nc] += (cost && cost[order[t]] > 0.f) ? (double)cost[order[t]] : (double)key[order[t]].second * (double)key[order[t]].second;
    nc++;
    i = j;
  }
  plan->nclass = nc;
  GH_TRY(ctx->reserve(B_KM_ORDER, (size_t)nprob + 1, &plan->d_order));
  GH_HIP(hipMemcpyAsync(plan->d_order, order.data(), (size_t)nprob * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(hipStreamSynchronize(ctx->stream));  // `order` is a local
  return GHICP_OK;
}

int gh_km4_launch_plan(ghicp_ctx* ctx, const Km2Problem* d_probs, const Km4Plan& plan) {
  for (int c = 0; c < plan.nclass; c++)
    if (plan.count[c] > 0) GH_TRY(k4_launch(ctx, d_probs, plan.count[c], plan.lds[c], plan.d_order + plan.begin[c]));
  return GHICP_OK;
}

// Aggregates the launch records written while kernel timing was on (ghicp_ctx_kernel_timing):
//   out[0] launches, [1] solves, [2] mean solve ms, [3] mean over launches of the LONGEST solve (ms), [4] mean launch span ms
//   (first block start -> last block end), [5] solve slots (resident workgroups on the chip), [6] idle-slot fraction =
//   1 - sum(solve time) / sum(min(slots, solves) x span), [7] max over launches of longest/mean solve.
extern "C" int ghicp_ctx_km_launch_stats(ghicp_ctx* ctx, double* out8) {
  GH_ENTER(ctx);
  GH_ARG(out8 != nullptr);
  for (int i = 0; i < 8; i++) out8[i] = 0.0;
  const long long nl = ctx->km_launches;
  if (nl <= 0 || !ctx->buf[B_KM_LSTAT].p) return GHICP_OK;
  GH_HIP(hipStreamSynchronize(ctx->stream));
  constexpr int W = ghicp_ctx::KM_LSTAT_W;
  std::vector<unsigned long long> h((size_t)nl * W);
  GH_HIP(hipMemcpy(h.data(), ctx->buf[B_KM_LSTAT].p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  const double tick_ms = 1e3 / 100e6;
  double solves = 0, sum_dt = 0, sum_max = 0, sum_span = 0, cap = 0, worst = 0, slots = 0;
  long long used = 0;
  for (long long l = 0; l < nl; l++) {
    const unsigned long long* r = &h[(size_t)l * W];
    if (r[4] == 0 || r[6] != 0) continue;  // r[6] != 0: a record of the persistent pair loop (ghicp_ctx_pair_loop_stats)
    const double start = (double)((1ull << 62) - r[0]), span = (double)r[1] - start;
    used++;
    solves += (double)r[4]; sum_dt += (double)r[2]; sum_max += (double)r[3]; sum_span += span;
    slots = (double)ctx->km_slots[(size_t)l];
    cap += std::min(slots, (double)r[4]) * span;
    worst = std::max(worst, (double)r[3] / ((double)r[2] / (double)r[4]));
  }
  if (used == 0) return GHICP_OK;
  out8[0] = (double)used; out8[1] = solves; out8[2] = sum_dt / solves * tick_ms; out8[3] = sum_max / (double)used * tick_ms;
  out8[4] = sum_span / (double)used * tick_ms; out8[5] = slots; out8[6] = cap > 0 ? 1.0 - sum_dt / cap : 0.0; out8[7] = worst;
  return GHICP_OK;
}

// Records of the persistent pair loop (loop.hip:k_pair_loop), one per BATCH (the launches of its LDS-occupancy classes share it),
// collected while kernel timing is on:  out[0] batches, [1] workgroups that ran, [2] solves, [3] mean solve ms, [4] longest solve ms,
// [5] mean batch span ms (first slot start -> last slot end), [6] idle-slot fraction = 1 - sum(slot lifetimes) / sum(capacity x span),
// capacity = the slots one class can keep resident (the classes compete for the same CUs; a slot is busy from its start to the moment
// its queue is empty, so what is idle is the tail of a batch, when the last pairs iterate alone), [7] share of the slot lifetimes spent
// inside Kuhn-Munkres solves (the rest: sweeps, graph build, rigid solve).
extern "C" int ghicp_ctx_loop_hazards(ghicp_ctx* ctx, int64_t* solves) {
  GH_ENTER(ctx);
  GH_ARG(solves != nullptr);
  *solves = (int64_t)ctx->loop_hazards;
  return GHICP_OK;
}

extern "C" int ghicp_ctx_pair_loop_stats(ghicp_ctx* ctx, double* out8) {
  GH_ENTER(ctx);
  GH_ARG(out8 != nullptr);
  for (int i = 0; i < 8; i++) out8[i] = 0.0;
  const long long nl = ctx->km_launches;
  if (nl <= 0 || !ctx->buf[B_KM_LSTAT].p) return GHICP_OK;
  for (hipStream_t a : ctx->aux_streams) GH_HIP(hipStreamSynchronize(a));
  GH_HIP(hipStreamSynchronize(ctx->stream));
  constexpr int W = ghicp_ctx::KM_LSTAT_W;
  std::vector<unsigned long long> h((size_t)nl * W);
  GH_HIP(hipMemcpy(h.data(), ctx->buf[B_KM_LSTAT].p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  const double tick_ms = 1e3 / 100e6;
  double launches = 0, slots = 0, solves = 0, sum_dt = 0, mx = 0, sum_span = 0, life = 0, cap = 0;
  for (long long l = 0; l < nl; l++) {
    const unsigned long long* r = &h[(size_t)l * W];
    if (r[6] == 0) continue;
    const double start = (double)((1ull << 62) - r[0]), span = (double)r[1] - start;
    launches++; slots += (double)r[6]; solves += (double)r[4]; sum_dt += (double)r[2]; mx = std::max(mx, (double)r[3]);
    sum_span += span; life += (double)r[5]; cap += std::min((double)ctx->km_slots[(size_t)l], (double)r[6]) * span;
  }
  if (launches == 0) return GHICP_OK;
  out8[0] = launches; out8[1] = slots; out8[2] = solves; out8[3] = solves > 0 ? sum_dt / solves * tick_ms : 0.0; out8[4] = mx * tick_ms;
  out8[5] = sum_span / launches * tick_ms; out8[6] = cap > 0 ? 1.0 - life / cap : 0.0; out8[7] = life > 0 ? sum_dt / life : 0.0;
  return GHICP_OK;
}
