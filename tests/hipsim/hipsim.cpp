// TEST INFRASTRUCTURE -- see include/hip/hip_runtime.h.  Block scheduler of the host SIMT interpreter: lanes are fibers (hand-written
// x86-64 context switch), a workgroup runs on one OS thread, the workgroups of a launch are spread over a small pool of OS threads.
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__SANITIZE_THREAD__)
// HIPSIM_TSAN=1 build: every lane is a ThreadSanitizer fiber; __syncthreads, wave collectives and workgroup boundaries are release /
// acquire points, so TSan reports LDS / global accesses of two lanes that no barrier or collective orders (a data race on the GPU
// unless the code relies on wave lock-step without saying so).
#include <sanitizer/tsan_interface.h>
#define HIPSIM_TSAN_ON 1
#else
#define HIPSIM_TSAN_ON 0
#endif

extern "C" void hipsim_swap(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipsim_swap
.type hipsim_swap,@function
hipsim_swap:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipsim_swap,.-hipsim_swap
)");

namespace hipsim {
enum { S_RUN = 0, S_SYNC = 1, S_WAVE = 2, S_DONE = 3 };
static const size_t STACK_BYTES = (HIPSIM_TSAN_ON ? 2048 : 256) * 1024;  // per lane (virtual; instrumented frames are larger)
static const int MAX_LANES = 1024;

struct Worker {
#if HIPSIM_TSAN_ON
  void* lane_fiber[1024] = {nullptr};
  void* sched_fiber = nullptr;
  char sync_block = 0, sync_wave[16] = {0}, sync_launch = 0;  // addresses TSan's release / acquire are keyed on
#endif
  char* stacks = nullptr;
  Lane lanes[MAX_LANES];
  void* sched_sp = nullptr;
  void (*tramp)(void*) = nullptr;
  void* closure = nullptr;
  bool yielded = false;
  ~Worker() { if (stacks) munmap(stacks, STACK_BYTES * MAX_LANES); }
};
thread_local Lane* cur = nullptr;
thread_local BlockInfo binfo;
static thread_local Worker* W = nullptr;
static thread_local hipError_t last_error = hipSuccess;
static std::atomic<long long> n_divergent{0}, n_launches{0}, n_blocks{0};
static int trace_div = -1;
static bool policy_max = false;
static int lane_order = 0;  // HIPSIM_ORDER: 0 ascending, 1 reverse, 2 pseudo-random (waves and lanes): results must not depend on it
static unsigned order_seed = 1;
static std::atomic<long long> api_calls[C_NUM];
void count(int what) { api_calls[what]++; }
int poison_byte() {
  static const int v = [] { const char* e = getenv("HIPSIM_POISON"); return e ? (atoi(e) & 0xff) : -1; }();
  return v;
}

hipError_t take_last_error(bool clear) {
  const hipError_t e = last_error;
  if (clear) last_error = hipSuccess;
  return e;
}

static void to_scheduler() {
#if HIPSIM_TSAN_ON
  __tsan_switch_to_fiber(W->sched_fiber, 0);
#endif
  hipsim_swap(&cur->sp, W->sched_sp);
}

static void fiber_main() {
#if HIPSIM_TSAN_ON
  __tsan_acquire(&W->sync_launch);  // after everything the previous workgroup on this worker did
#endif
  W->tramp(W->closure);
#if HIPSIM_TSAN_ON
  __tsan_release(&W->sync_launch);
#endif
  cur->state = S_DONE;
  to_scheduler();
  fprintf(stderr, "hipsim: a finished lane was resumed\n");
  abort();
}

uint64_t wave_collective(int op, int site, uint64_t payload, int src) {
  Lane* me = cur;
  me->op = op; me->site = site; me->payload = payload; me->src = src;
  me->state = S_WAVE;
#if HIPSIM_TSAN_ON
  __tsan_release(&W->sync_wave[me->wave]);
#endif
  to_scheduler();
#if HIPSIM_TSAN_ON
  __tsan_acquire(&W->sync_wave[me->wave]);
#endif
  return me->result;
}
void sync_threads() {
  cur->state = S_SYNC;
#if HIPSIM_TSAN_ON
  __tsan_release(&W->sync_block);
#endif
  to_scheduler();
#if HIPSIM_TSAN_ON
  __tsan_acquire(&W->sync_block);
#endif
}
void yield() {
  if (!cur) return;  // host code
  W->yielded = true;
  to_scheduler();
}

static Worker* worker() {
  static thread_local Worker holder;
  if (!holder.stacks) {
    void* m = mmap(nullptr, STACK_BYTES * MAX_LANES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { perror("hipsim: mmap of the lane stacks"); abort(); }
    holder.stacks = static_cast<char*>(m);
  }
  return &holder;
}

// resolve the pending collective of wave [l0, l1): the group at the smallest source line goes first (lanes that took a divergent
// branch are behind the lanes already waiting at the reconvergence point)
static void resolve(Lane* L, int l0, int l1, int n_live) {
  int site = policy_max ? -1 : INT32_MAX, op = 0;  // HIPSIM_POLICY=max: the opposite order, to expose results that depend on it
  for (int i = l0; i < l1; i++)
    if (L[i].state == S_WAVE && (policy_max ? L[i].site > site : L[i].site < site)) { site = L[i].site; op = L[i].op; }
  uint64_t mask = 0, ball = 0;
  int first = -1, n = 0;
  for (int i = l0; i < l1; i++)
    if (L[i].state == S_WAVE && L[i].site == site && L[i].op == op) {
      mask |= 1ull << (i - l0);
      if (first < 0) first = i;
      if (L[i].payload & 1) ball |= 1ull << (i - l0);
      n++;
    }
  if (n != n_live) {
    n_divergent++;
    if (trace_div > 0) fprintf(stderr, "hipsim: divergent collective op %d at line %d: %d of %d live lanes\n", op, site, n, n_live);
  }
  for (int i = l0; i < l1; i++) {
    if (!((mask >> (i - l0)) & 1)) continue;
    switch (op) {
      case OP_BALLOT: L[i].result = ball; break;
      case OP_SHFL: {
        const int s = L[i].src;
        L[i].result = (s >= 0 && s < l1 - l0 && ((mask >> s) & 1)) ? L[l0 + s].payload : L[i].payload;
        break;
      }
      case OP_FIRST: L[i].result = L[first].payload; break;
      case OP_MFMA_I32_32X32X32_I8: {  // the lowest lane computes the whole block into every lane's operand record
        if (i != first) break;
        if (n != 64 || l1 - l0 != 64) { fprintf(stderr, "hipsim: MFMA executed by %d lanes of a wave (needs all 64)\n", n); abort(); }
        int d[64][16];
        for (int l = 0; l < 64; l++) {
          const MfmaI8* ml = reinterpret_cast<const MfmaI8*>((uintptr_t)L[l0 + l].payload);
          for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            int acc = ml->c[r];
            for (int h = 0; h < 2; h++) {
              const MfmaI8* ma = reinterpret_cast<const MfmaI8*>((uintptr_t)L[l0 + row + 32 * h].payload);
              const MfmaI8* mb = reinterpret_cast<const MfmaI8*>((uintptr_t)L[l0 + col + 32 * h].payload);
              for (int k = 0; k < 16; k++) acc += (int)ma->a[k] * (int)mb->b[k];
            }
            d[l][r] = acc;
          }
        }
        for (int l = 0; l < 64; l++) memcpy(reinterpret_cast<MfmaI8*>((uintptr_t)L[l0 + l].payload)->c, d[l], sizeof(d[l]));
        break;
      }
      default: L[i].result = 0;
    }
  }
  for (int i = l0; i < l1; i++)
    if ((mask >> (i - l0)) & 1) L[i].state = S_RUN;
}

static void run_block(const dim3& grid, const dim3& block, unsigned long long b) {
  Worker* w = W;
  Lane* L = w->lanes;
  const int nt = (int)(block.x * block.y * block.z);
  binfo.bdim = block; binfo.gdim = grid;
  binfo.bid.x = (unsigned)(b % grid.x); binfo.bid.y = (unsigned)((b / grid.x) % grid.y); binfo.bid.z = (unsigned)(b / ((unsigned long long)grid.x * grid.y));
  for (int t = 0; t < nt; t++) {
    Lane& l = L[t];
    l.tid.x = t % block.x; l.tid.y = (t / block.x) % block.y; l.tid.z = t / (block.x * block.y);
    l.lane = t & 63; l.wave = t >> 6; l.state = S_RUN;
    void** top = reinterpret_cast<void**>(w->stacks + STACK_BYTES * (size_t)(t + 1));
    top[-1] = nullptr;                                   // return address slot of fiber_main (never used)
    top[-2] = reinterpret_cast<void*>(&fiber_main);      // `ret` of the first switch lands here
    for (int k = 3; k <= 8; k++) top[-k] = nullptr;      // rbp rbx r12 r13 r14 r15
    l.sp = top - 8;
  }
#if HIPSIM_TSAN_ON
  w->sched_fiber = __tsan_get_current_fiber();
  __tsan_release(&w->sync_launch);  // the host side of the launch happens before the lanes
#endif
  const int nw = (nt + 63) / 64;
  long long idle_rounds = 0;
  unsigned rng_state = order_seed + (unsigned)b * 747796405u;
  for (;;) {
    bool progressed = false;
    for (int wq = 0; wq < nw; wq++) {
      int wv = wq;
      if (lane_order == 1) wv = nw - 1 - wq;
      else if (lane_order == 2) wv = (int)((wq + (rng_state >> 8)) % (unsigned)nw);
      const int l0 = wv * 64, l1 = std::min(nt, l0 + 64);
      for (;;) {
        int ran = 0;
        w->yielded = false;
        const int nl = l1 - l0;
        unsigned rot = 0, step = 1;
        if (lane_order == 2) { rng_state = rng_state * 1664525u + 1013904223u; rot = rng_state >> 10; step = (rng_state >> 4) | 1u; }  // odd step: a permutation of 64
        for (int q = 0; q < nl; q++) {
          int i = l0 + q;
          if (lane_order == 1) i = l1 - 1 - q;
          else if (lane_order == 2 && nl == 64) i = l0 + (int)((rot + (unsigned)q * step) & 63u);
          if (L[i].state == S_RUN) {
            cur = &L[i];
#if HIPSIM_TSAN_ON
            if (!w->lane_fiber[i]) w->lane_fiber[i] = __tsan_create_fiber(0);
            __tsan_switch_to_fiber(w->lane_fiber[i], 0);
#endif
            hipsim_swap(&w->sched_sp, L[i].sp);
            ran++;
          }
        }
        cur = nullptr;
        int n_run = 0, n_wave = 0, n_live = 0;
        for (int i = l0; i < l1; i++) { n_run += L[i].state == S_RUN; n_wave += L[i].state == S_WAVE; n_live += L[i].state != S_DONE; }
        if (ran && !(w->yielded && n_run == ran && n_wave == 0)) progressed = true;
        if (n_run > 0) break;  // pollers: let the other waves run
        if (n_wave > 0) { resolve(L, l0, l1, n_live); progressed = true; continue; }
        break;
      }
    }
    int n_run = 0, n_sync = 0, n_done = 0;
    for (int i = 0; i < nt; i++) { n_run += L[i].state == S_RUN; n_sync += L[i].state == S_SYNC; n_done += L[i].state == S_DONE; }
    if (n_done == nt) {
#if HIPSIM_TSAN_ON
      __tsan_acquire(&w->sync_launch);
#endif
      return;
    }
    if (n_run == 0) {
      if (n_sync + n_done != nt) { fprintf(stderr, "hipsim: scheduler inconsistency\n"); abort(); }
      for (int i = 0; i < nt; i++)
        if (L[i].state == S_SYNC) L[i].state = S_RUN;  // barrier of the lanes that have not exited
      idle_rounds = 0;
      continue;
    }
    if (!progressed && ++idle_rounds > 50'000'000) {
      fprintf(stderr, "hipsim: workgroup (%u,%u,%u) only polls: %d lanes spin, %d wait at a barrier (a wait on another workgroup cannot be simulated)\n",
              binfo.bid.x, binfo.bid.y, binfo.bid.z, n_run, n_sync);
      abort();
    }
    if (progressed) idle_rounds = 0;
  }
}

// ---------------------------------------------------------------------------------------------------- launches
struct Job {
  dim3 grid, block;
  void (*tramp)(void*);
  void* closure;
  size_t shmem = 0;
  char* (*smem_fn[2])() = {nullptr, nullptr};
  std::atomic<unsigned long long> next{0};
  unsigned long long total = 0;
  std::atomic<int> active{0};
};
// leaked on purpose: the detached pool threads wait on these until the process ends
static std::mutex &pool_mu = *new std::mutex, &job_mu = *new std::mutex;
static std::condition_variable &pool_cv = *new std::condition_variable, &done_cv = *new std::condition_variable;
static Job* pool_job = nullptr;
static unsigned long long pool_gen = 0;
static bool pool_started = false;
static bool pool_stop = false;

static void thread_altstack();
static void work_on(Job* j) {
  thread_altstack();
  W = worker();
  W->tramp = j->tramp; W->closure = j->closure;
  for (;;) {
    const unsigned long long b = j->next.fetch_add(1);
    if (b >= j->total) break;
    // guard behind the dynamic LDS the launch asked for (this thread's instances of the two `smem` candidates)
    constexpr size_t GUARD = 256;
    char* sm[2] = {j->smem_fn[0] ? j->smem_fn[0]() : nullptr, j->smem_fn[1] ? j->smem_fn[1]() : nullptr};
    for (char* p : sm)
      if (p) memset(p + j->shmem, 0xC7, GUARD);
    run_block(j->grid, j->block, b);
    for (char* p : sm)
      if (p)
        for (size_t k = 0; k < GUARD; k++)
          if ((unsigned char)p[j->shmem + k] != 0xC7) {
            fprintf(stderr, "hipsim: a workgroup wrote dynamic LDS byte %zu, the launch asked for %zu bytes (block %llu of the grid)\n", j->shmem + k, j->shmem, b);
            abort();
          }
  }
}
static void pool_main() {
  unsigned long long seen = 0;
  for (;;) {
    Job* j;
    {
      std::unique_lock<std::mutex> lk(pool_mu);
      pool_cv.wait(lk, [&] { return pool_stop || (pool_job && pool_gen != seen); });
      if (pool_stop) return;
      seen = pool_gen;
      j = pool_job;
      j->active++;
    }
    work_on(j);
    {
      std::unique_lock<std::mutex> lk(pool_mu);
      j->active--;
      done_cv.notify_all();
    }
  }
}
static int pool_threads() {
  static int n = [] {
    const char* e = getenv("HIPSIM_THREADS");
    int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(v, 64));
  }();
  return n;
}

static void on_segv(int sig, siginfo_t* si, void*) {
  char msg[256];
  Lane* l = cur;
  int n = snprintf(msg, sizeof msg, "\nhipsim: signal %d at address %p; workgroup (%u,%u,%u) thread (%u,%u,%u)%s\n", sig, si->si_addr, binfo.bid.x, binfo.bid.y,
                   binfo.bid.z, l ? l->tid.x : 0, l ? l->tid.y : 0, l ? l->tid.z : 0, l ? "" : " [host code]");
  (void)!write(2, msg, n);
  void* bt[48];
  backtrace_symbols_fd(bt, backtrace(bt, 48), 2);
  _exit(139);
}
static bool segv_trace = false;
static void thread_altstack() {
  static thread_local char* alt = nullptr;
  if (!segv_trace || alt) return;
  alt = static_cast<char*>(malloc(1 << 16));
  stack_t ss;
  ss.ss_sp = alt; ss.ss_size = 1 << 16; ss.ss_flags = 0;
  sigaltstack(&ss, nullptr);
}
static void install_segv_trace() {
  segv_trace = true;
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, nullptr);
  sigaction(SIGBUS, &sa, nullptr);
}

void launch(dim3 grid, dim3 block, size_t shmem, void (*tramp)(void*), void* closure, char* (*smem_a)(), char* (*smem_b)()) {
  if (trace_div < 0) {
    trace_div = getenv("HIPSIM_TRACE_DIVERGENCE") ? 1 : 0;
    policy_max = getenv("HIPSIM_POLICY") && !strcmp(getenv("HIPSIM_POLICY"), "max");
    if (const char* o = getenv("HIPSIM_ORDER")) {
      if (!strcmp(o, "reverse")) lane_order = 1;
      else if (!strncmp(o, "random", 6)) { lane_order = 2; if (o[6] == ':') order_seed = (unsigned)atoi(o + 7) * 2654435761u + 1u; }
    }
    if (getenv("HIPSIM_SEGV_TRACE")) install_segv_trace();
  }
  const unsigned long long total = (unsigned long long)grid.x * grid.y * grid.z;
  const unsigned long long nt = (unsigned long long)block.x * block.y * block.z;
  if (total == 0 || nt == 0 || nt > MAX_LANES || shmem > 160 * 1024 || grid.y > 65535 || grid.z > 65535) {
    last_error = hipErrorInvalidConfiguration;
    return;
  }
  if (cur) { fprintf(stderr, "hipsim: launch from device code\n"); abort(); }
  n_launches++; n_blocks += (long long)total;
  Job j;
  j.grid = grid; j.block = block; j.tramp = tramp; j.closure = closure; j.total = total;
  j.shmem = shmem; j.smem_fn[0] = smem_a; j.smem_fn[1] = smem_b;
  const int nthreads = pool_threads();
  if (total < 4 || nthreads == 1) {  // small launches stay on the calling thread (host threads may launch concurrently)
    work_on(&j);
    return;
  }
  std::lock_guard<std::mutex> one_job(job_mu);
  {
    std::unique_lock<std::mutex> lk(pool_mu);
    if (!pool_started) {
      pool_started = true;
      for (int i = 0; i < nthreads - 1; i++) std::thread(pool_main).detach();
    }
    pool_job = &j;
    pool_gen++;
  }
  pool_cv.notify_all();
  work_on(&j);
  std::unique_lock<std::mutex> lk(pool_mu);
  pool_job = nullptr;  // late wakers must not pick the job up any more
  done_cv.wait(lk, [&] { return j.active.load() == 0; });
}
}  // namespace hipsim

extern "C" void hipsim_counters(long long* out8) {
  out8[0] = hipsim::n_launches.load(); out8[1] = hipsim::n_blocks.load(); out8[2] = hipsim::n_divergent.load();
  for (int i = 0; i < hipsim::C_NUM; i++) out8[3 + i] = hipsim::api_calls[i].load();
}
