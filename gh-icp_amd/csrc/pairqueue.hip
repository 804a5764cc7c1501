// Pair queue behind the C ABI (include/ghicp_c.h, "Pair queue"): independent scan pairs sharded over the GPUs of one node, one process
// per GPU.  Host code only -- no kernel: pairs share nothing, the ranks exchange the manifest (one ncclBroadcast), per
// step the result records (one ncclAllGather of rows x 19 doubles per rank) and, for the dynamic split, claims on a shared counter.
//   * rendezvous segment: a file every rank maps (rank 0 creates it under a temporary name and renames it into place once it is
//     initialised); it holds the ncclUniqueId, a sense-reversing barrier, the shared counter and -- for GHICP_PQ_HOST -- a data window
//     through which broadcast and gather run without RCCL (two ranks on one GPU, CPU-only test machines);
//   * RCCL is opened with dlopen when the first GHICP_PQ_RCCL queue is created: libghicp_hip.so itself does not depend on librccl, and a
//     process that already holds a copy (PyTorch ships one) gets that one.
#include "ctx.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdlib>

namespace {

// ---- the slice of the RCCL API this file uses (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE, ncclChar = 0, ncclFloat64 = 8)
struct PqNcclId { char internal[128]; };
typedef void* PqNcclComm;
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(PqNcclId*) = nullptr;
  int (*CommInitRank)(PqNcclComm*, int, PqNcclId, int) = nullptr;
  int (*CommDestroy)(PqNcclComm) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, PqNcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, PqNcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string path;
};
Rccl* rccl_open(std::string* why) {
  static Rccl R;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* env = getenv("GHICP_RCCL_LIB");
    const char* cand[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* c : cand) {
      if (!c || !*c) continue;
      R.so = dlopen(c, RTLD_NOW | RTLD_LOCAL);
      if (R.so) { R.path = c; break; }
    }
    if (R.so) {
      R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.so, "ncclGetUniqueId"));
      R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.so, "ncclCommInitRank"));
      R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.so, "ncclCommDestroy"));
      R.Broadcast = reinterpret_cast<decltype(R.Broadcast)>(dlsym(R.so, "ncclBroadcast"));
      R.AllGather = reinterpret_cast<decltype(R.AllGather)>(dlsym(R.so, "ncclAllGather"));
      R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.so, "ncclGetErrorString"));
      if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.Broadcast || !R.AllGather) { dlclose(R.so); R.so = nullptr; }
    }
  }
  if (!R.so) {
    if (why) *why = "librccl could not be opened (tried GHICP_RCCL_LIB, librccl.so, /opt/rocm/lib/librccl.so): no GHICP_PQ_RCCL transport in this process";
    return nullptr;
  }
  return &R;
}

constexpr uint64_t PQ_MAGIC = 0x4748495051303031ull;  // "GHIPQ001"
constexpr size_t PQ_DATA = (size_t)4 << 20;           // data window of the host transport

struct Seg {
  std::atomic<uint64_t> magic;
  uint32_t world, transport;
  std::atomic<int64_t> counter;       // dynamic split: next unclaimed pair id
  std::atomic<uint32_t> bar_count, bar_gen;
  std::atomic<uint32_t> attached, id_ready, failed;
  PqNcclId nccl_id;
  uint64_t data_cap;
  alignas(64) unsigned char data[1];
};
static_assert(std::atomic<uint64_t>::is_always_lock_free && std::atomic<uint32_t>::is_always_lock_free, "process-shared atomics must be lock free");

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
void backoff(int& spins) {
  if (++spins < 64) return;
  if (spins < 4096) { sched_yield(); return; }
  timespec t = {0, 50000};
  nanosleep(&t, nullptr);
}

}  // namespace

struct ghicp_pairqueue {
  ghicp_ctx* ctx = nullptr;
  int rank = 0, world = 1, transport = GHICP_PQ_HOST;
  double timeout_s = 120.0;
  std::string path, err;
  Seg* seg = nullptr;
  size_t seg_bytes = 0;
  uint32_t gen = 0;  // barrier generation this rank has completed
  Rccl* R = nullptr;
  PqNcclComm comm = nullptr;
  void* dev = nullptr;  // device staging of the RCCL transport
  size_t dev_cap = 0;

  int fail(int code, const char* fmt, ...) {
    char tmp[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    err = tmp;
    return code;
  }
  int barrier() {
    if (world == 1) return GHICP_OK;
    const uint32_t g = gen;
    if (seg->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
      seg->bar_count.store(0, std::memory_order_relaxed);
      seg->bar_gen.store(g + 1, std::memory_order_release);
    } else {
      const double t0 = now_s();
      int spins = 0;
      while (seg->bar_gen.load(std::memory_order_acquire) == g) {
        if (seg->failed.load(std::memory_order_relaxed)) return fail(GHICP_ERR_INTERNAL, "pair queue: another rank reported a failure");
        backoff(spins);
        if ((spins & 1023) == 0 && now_s() - t0 > timeout_s) {
          seg->failed.store(1, std::memory_order_relaxed);
          return fail(GHICP_ERR_INTERNAL, "pair queue: rank %d waited %.0f s at a barrier (a rank is missing, or the ranks call in different orders)", rank, timeout_s);
        }
      }
    }
    gen = g + 1;
    return GHICP_OK;
  }
  int dev_reserve(size_t bytes) {
    if (bytes <= dev_cap) return GHICP_OK;
    if (dev) (void)hipFree(dev);
    dev = nullptr; dev_cap = 0;
    if (hipMalloc(&dev, bytes + 256) != hipSuccess) return fail(GHICP_ERR_HIP, "pair queue: staging allocation of %zu bytes failed", bytes);
    dev_cap = bytes;
    return GHICP_OK;
  }
};

#define PQ_ENTER(q)                       \
  do {                                    \
    if (!(q)) return GHICP_ERR_ARG;       \
    if ((q)->ctx && hipSetDevice((q)->ctx->device) != hipSuccess) return (q)->fail(GHICP_ERR_HIP, "hipSetDevice failed"); \
  } while (0)
#define PQ_ARG(q, cond)                                                                              \
  do {                                                                                               \
    if (!(cond)) return (q)->fail(GHICP_ERR_ARG, "%s: bad argument (%s)", __func__, #cond);         \
  } while (0)
#define PQ_TRY(call)               \
  do {                             \
    int r_ = (call);               \
    if (r_ != GHICP_OK) return r_; \
  } while (0)
#define PQ_NCCL(q, call)                                                                                                   \
  do {                                                                                                                     \
    int e_ = (call);                                                                                                       \
    if (e_ != 0) return (q)->fail(GHICP_ERR_INTERNAL, "%s -> RCCL error %d (%s)", #call, e_, (q)->R->GetErrorString ? (q)->R->GetErrorString(e_) : "?"); \
  } while (0)
#define PQ_HIP(q, call)                                                                                      \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) return (q)->fail(GHICP_ERR_HIP, "%s -> %s", #call, hipGetErrorString(e_));       \
  } while (0)

extern "C" const char* ghicp_pairqueue_last_error(const ghicp_pairqueue* q) { return q ? q->err.c_str() : "null pair queue"; }

extern "C" int ghicp_pairqueue_info(const ghicp_pairqueue* q, int32_t* rank, int32_t* world, int32_t* transport) {
  if (!q) return GHICP_ERR_ARG;
  if (rank) *rank = q->rank;
  if (world) *world = q->world;
  if (transport) *transport = q->transport;
  return GHICP_OK;
}

extern "C" int ghicp_pairqueue_destroy(ghicp_pairqueue* q) {
  if (!q) return GHICP_ERR_ARG;
  int rc = GHICP_OK;
  if (q->seg) {
    rc = q->barrier();  // nobody is still inside the data window
    if (q->comm && q->R) (void)q->R->CommDestroy(q->comm);
    munmap(q->seg, q->seg_bytes);
    if (q->rank == 0) unlink(q->path.c_str());
  }
  if (q->dev) { if (q->ctx) (void)hipSetDevice(q->ctx->device); (void)hipFree(q->dev); }
  delete q;
  return rc;
}

extern "C" int ghicp_pairqueue_create(ghicp_ctx* ctx, const char* rendezvous, int32_t rank, int32_t world, int32_t transport, double timeout_s,
                                      ghicp_pairqueue** out) {
  if (!out || !rendezvous || !*rendezvous || world < 1 || world > 4096 || rank < 0 || rank >= world ||
      (transport != GHICP_PQ_HOST && transport != GHICP_PQ_RCCL) || (transport == GHICP_PQ_RCCL && !ctx))
    return ctx ? ctx->fail(GHICP_ERR_ARG, "ghicp_pairqueue_create: bad argument") : GHICP_ERR_ARG;
  *out = nullptr;
  ghicp_pairqueue* q = new ghicp_pairqueue();
  q->ctx = ctx; q->rank = rank; q->world = world; q->transport = transport; q->path = rendezvous;
  q->timeout_s = timeout_s > 0.0 ? timeout_s : 120.0;
  q->seg_bytes = sizeof(Seg) + PQ_DATA;
  auto bail = [&](int code, const char* what) {
    if (ctx) ctx->fail(code, "ghicp_pairqueue_create(rank %d of %d, %s): %s%s%s", rank, world, rendezvous, what, errno ? ": " : "", errno ? strerror(errno) : "");
    if (q->seg) munmap(q->seg, q->seg_bytes);
    delete q;
    return code;
  };
  errno = 0;
  int fd = -1;
  if (rank == 0) {
    const std::string tmp = q->path + ".init";
    unlink(tmp.c_str());
    unlink(q->path.c_str());  // a stale segment of an earlier job under the same name
    fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return bail(GHICP_ERR_INTERNAL, "cannot create the rendezvous segment");
    if (ftruncate(fd, (off_t)q->seg_bytes) != 0) { close(fd); return bail(GHICP_ERR_INTERNAL, "cannot size the rendezvous segment"); }
    void* m = mmap(nullptr, q->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return bail(GHICP_ERR_INTERNAL, "cannot map the rendezvous segment");
    q->seg = new (m) Seg();
    q->seg->world = (uint32_t)world; q->seg->transport = (uint32_t)transport;
    q->seg->counter.store(0); q->seg->bar_count.store(0); q->seg->bar_gen.store(0);
    q->seg->attached.store(1); q->seg->id_ready.store(0); q->seg->failed.store(0);
    q->seg->data_cap = PQ_DATA;
    q->seg->magic.store(PQ_MAGIC, std::memory_order_release);
    if (rename(tmp.c_str(), q->path.c_str()) != 0) return bail(GHICP_ERR_INTERNAL, "cannot publish the rendezvous segment");
  } else {
    const double t0 = now_s();
    int spins = 4096;
    for (;;) {
      fd = open(q->path.c_str(), O_RDWR);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= q->seg_bytes) break;
        close(fd);
        fd = -1;
      }
      if (now_s() - t0 > q->timeout_s) { errno = 0; return bail(GHICP_ERR_INTERNAL, "rank 0's rendezvous segment did not appear in time"); }
      backoff(spins);
    }
    errno = 0;
    void* m = mmap(nullptr, q->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return bail(GHICP_ERR_INTERNAL, "cannot map the rendezvous segment");
    q->seg = reinterpret_cast<Seg*>(m);
    if (q->seg->magic.load(std::memory_order_acquire) != PQ_MAGIC || q->seg->world != (uint32_t)world || q->seg->transport != (uint32_t)transport) {
      errno = 0;
      return bail(GHICP_ERR_ARG, "the rendezvous segment belongs to another job (magic / world size / transport differ)");
    }
    q->seg->attached.fetch_add(1);
  }
  if (transport == GHICP_PQ_RCCL) {
    std::string why;
    q->R = rccl_open(&why);
    errno = 0;
    if (!q->R) { q->seg->failed.store(1); return bail(GHICP_ERR_INTERNAL, why.c_str()); }
    if (hipSetDevice(ctx->device) != hipSuccess) return bail(GHICP_ERR_HIP, "hipSetDevice failed");
    if (rank == 0) {
      if (q->R->GetUniqueId(&q->seg->nccl_id) != 0) { q->seg->failed.store(1); return bail(GHICP_ERR_INTERNAL, "ncclGetUniqueId failed"); }
      q->seg->id_ready.store(1, std::memory_order_release);
    } else {
      const double t0 = now_s();
      int spins = 0;
      while (!q->seg->id_ready.load(std::memory_order_acquire)) {
        if (q->seg->failed.load() || now_s() - t0 > q->timeout_s) return bail(GHICP_ERR_INTERNAL, "rank 0 did not publish the ncclUniqueId");
        backoff(spins);
      }
    }
    const int e = q->R->CommInitRank(&q->comm, world, q->seg->nccl_id, rank);
    if (e != 0) { q->seg->failed.store(1); return bail(GHICP_ERR_INTERNAL, "ncclCommInitRank failed"); }
  }
  const int rc = q->barrier();  // every rank is attached (and, for RCCL, in the communicator) when create returns
  if (rc != GHICP_OK) {
    if (ctx) ctx->fail(rc, "ghicp_pairqueue_create: %s", q->err.c_str());
    if (q->comm) (void)q->R->CommDestroy(q->comm);
    munmap(q->seg, q->seg_bytes);
    delete q;
    return rc;
  }
  *out = q;
  return GHICP_OK;
}

extern "C" int ghicp_pairqueue_barrier(ghicp_pairqueue* q) {
  PQ_ENTER(q);
  return q->barrier();
}

extern "C" int ghicp_pairqueue_broadcast(ghicp_pairqueue* q, void* buf, int64_t bytes, int32_t root) {
  PQ_ENTER(q);
  PQ_ARG(q, bytes >= 0 && (bytes == 0 || buf != nullptr) && root >= 0 && root < q->world);
  if (bytes == 0 || q->world == 1) {
    if (q->transport != GHICP_PQ_RCCL || bytes == 0) return GHICP_OK;  // (a one-rank RCCL queue still runs the collective: the path under test)
  }
  if (q->transport == GHICP_PQ_RCCL) {
    hipStream_t s = q->ctx->stream;
    PQ_TRY(q->dev_reserve((size_t)bytes));
    if (q->rank == root) PQ_HIP(q, hipMemcpyAsync(q->dev, buf, (size_t)bytes, hipMemcpyHostToDevice, s));
    PQ_NCCL(q, q->R->Broadcast(q->dev, q->dev, (size_t)bytes, /*ncclChar*/ 0, root, q->comm, s));
    if (q->rank != root) PQ_HIP(q, hipMemcpyAsync(buf, q->dev, (size_t)bytes, hipMemcpyDeviceToHost, s));
    PQ_HIP(q, hipStreamSynchronize(s));
    return GHICP_OK;
  }
  unsigned char* p = reinterpret_cast<unsigned char*>(buf);
  for (int64_t off = 0; off < bytes; off += (int64_t)q->seg->data_cap) {
    const size_t part = (size_t)std::min<int64_t>(bytes - off, (int64_t)q->seg->data_cap);
    if (q->rank == root) memcpy(q->seg->data, p + off, part);
    PQ_TRY(q->barrier());
    if (q->rank != root) memcpy(p + off, q->seg->data, part);
    PQ_TRY(q->barrier());
  }
  return GHICP_OK;
}

extern "C" int ghicp_pairqueue_static_share(const ghicp_pairqueue* q, int64_t n_pairs, int64_t* ids, int64_t cap, int64_t* n) {
  if (!q) return GHICP_ERR_ARG;
  if (n_pairs < 0 || !n || (cap > 0 && !ids)) return GHICP_ERR_ARG;
  int64_t c = 0;
  for (int64_t p = q->rank; p < n_pairs; p += q->world) {
    if (c < cap) ids[c] = p;
    c++;
  }
  *n = c;
  return c <= cap || cap == 0 ? GHICP_OK : GHICP_ERR_CAPACITY;
}

extern "C" int ghicp_pairqueue_claim(ghicp_pairqueue* q, int64_t count, int64_t limit, int64_t* first, int64_t* n) {
  if (!q) return GHICP_ERR_ARG;
  PQ_ARG(q, count >= 0 && limit >= 0 && first != nullptr && n != nullptr);
  int64_t lo = limit;
  if (count > 0) lo = q->seg->counter.fetch_add(count, std::memory_order_acq_rel);
  lo = std::min(lo, limit);
  *first = lo;
  *n = std::min(lo + count, limit) - lo;
  return GHICP_OK;
}

extern "C" int ghicp_pairqueue_counter_reset(ghicp_pairqueue* q) {
  PQ_ENTER(q);
  PQ_TRY(q->barrier());  // every claim of the previous job has been made
  if (q->rank == 0) q->seg->counter.store(0, std::memory_order_release);
  return q->barrier();
}

extern "C" int ghicp_pairqueue_gather_records(ghicp_pairqueue* q, const double* block, int64_t rows, double* all) {
  PQ_ENTER(q);
  PQ_ARG(q, rows >= 0 && (rows == 0 || (block != nullptr && all != nullptr)));
  if (rows == 0) return GHICP_OK;
  const size_t bb = (size_t)rows * GHICP_PQ_RECORD_WIDTH * sizeof(double);
  if (q->transport == GHICP_PQ_RCCL) {
    hipStream_t s = q->ctx->stream;
    PQ_TRY(q->dev_reserve(bb * ((size_t)q->world + 1)));
    unsigned char* d = reinterpret_cast<unsigned char*>(q->dev);
    PQ_HIP(q, hipMemcpyAsync(d, block, bb, hipMemcpyHostToDevice, s));
    PQ_NCCL(q, q->R->AllGather(d, d + bb, (size_t)rows * GHICP_PQ_RECORD_WIDTH, /*ncclFloat64*/ 8, q->comm, s));
    PQ_HIP(q, hipMemcpyAsync(all, d + bb, bb * (size_t)q->world, hipMemcpyDeviceToHost, s));
    PQ_HIP(q, hipStreamSynchronize(s));
    return GHICP_OK;
  }
  if (q->world == 1) { memcpy(all, block, bb); return GHICP_OK; }
  // host transport: the window holds `per` bytes of every rank's block at a time
  const size_t per = std::max<size_t>(sizeof(double), (q->seg->data_cap / (size_t)q->world) & ~(size_t)7);
  const unsigned char* src = reinterpret_cast<const unsigned char*>(block);
  unsigned char* dst = reinterpret_cast<unsigned char*>(all);
  for (size_t off = 0; off < bb; off += per) {
    const size_t part = std::min(per, bb - off);
    memcpy(q->seg->data + (size_t)q->rank * per, src + off, part);
    PQ_TRY(q->barrier());
    for (int r = 0; r < q->world; r++) memcpy(dst + (size_t)r * bb + off, q->seg->data + (size_t)r * per, part);
    PQ_TRY(q->barrier());
  }
  return GHICP_OK;
}

extern "C" int ghicp_pairqueue_pack_records(const int64_t* ids, const ghicp_pair_stats* stats, int64_t n_mine, int64_t rows, double* block) {
  if (n_mine < 0 || rows < n_mine || (rows > 0 && !block) || (n_mine > 0 && (!ids || !stats))) return GHICP_ERR_ARG;
  for (int64_t i = 0; i < rows; i++) {
    double* r = block + (size_t)i * GHICP_PQ_RECORD_WIDTH;
    if (i < n_mine) {
      r[0] = (double)ids[i]; r[1] = (double)stats[i].iterations; r[2] = (double)stats[i].converged;
      for (int k = 0; k < 16; k++) r[3 + k] = stats[i].Rt[k];
    } else {
      r[0] = -1.0;
      for (int k = 1; k < GHICP_PQ_RECORD_WIDTH; k++) r[k] = 0.0;
    }
  }
  return GHICP_OK;
}

extern "C" int ghicp_pairqueue_register_pairs(ghicp_pairqueue* q, ghicp_ctx* ctx, const ghicp_pair_config* cfg, int64_t n_pairs, const float* const* xyzS,
                                              const int64_t* nS, const float* const* xyzT, const int64_t* nT, int stride, int64_t chunk,
                                              ghicp_pair_stats* stats, double* records) {
  PQ_ENTER(q);
  PQ_ARG(q, ctx != nullptr && cfg != nullptr && n_pairs >= 0 && n_pairs < (1ll << 31) && records != nullptr &&
                (n_pairs == 0 || (xyzS && nS && xyzT && nT)) && stride >= 3);
  std::vector<int64_t> mine;
  std::vector<ghicp_pair_stats> st;
  auto run = [&](int64_t first, int64_t count, int64_t step) -> int {  // pairs first, first + step, ... (count of them) in one batched call
    if (count <= 0) return GHICP_OK;
    std::vector<const float*> s((size_t)count), t((size_t)count);
    std::vector<int64_t> ns((size_t)count), nt((size_t)count);
    for (int64_t i = 0; i < count; i++) {
      const int64_t p = first + i * step;
      s[(size_t)i] = xyzS[p]; t[(size_t)i] = xyzT[p]; ns[(size_t)i] = nS[p]; nt[(size_t)i] = nT[p];
      mine.push_back(p);
    }
    const size_t at = st.size();
    st.resize(at + (size_t)count);
    const int rc = ghicp_register_pairs(ctx, cfg, (int32_t)count, s.data(), ns.data(), t.data(), nt.data(), stride, st.data() + at);
    if (rc != GHICP_OK) { q->seg->failed.store(1); return q->fail(rc, "ghicp_register_pairs: %s", ghicp_last_error(ctx)); }
    return GHICP_OK;
  };
  int64_t rows;
  if (chunk <= 0) {  // static: p mod world
    const int64_t cnt = n_pairs > q->rank ? (n_pairs - q->rank + q->world - 1) / q->world : 0;
    PQ_TRY(run(q->rank, cnt, q->world));
    rows = std::max<int64_t>(1, (n_pairs + q->world - 1) / q->world);
  } else {           // dynamic: chunks of consecutive pair ids from the shared counter
    PQ_TRY(ghicp_pairqueue_counter_reset(q));
    for (;;) {
      int64_t first = 0, n = 0;
      PQ_TRY(ghicp_pairqueue_claim(q, chunk, n_pairs, &first, &n));
      if (n == 0) break;
      PQ_TRY(run(first, n, 1));
    }
    rows = std::max<int64_t>(1, n_pairs);  // a rank may have drawn every pair
  }
  std::vector<double> block((size_t)rows * GHICP_PQ_RECORD_WIDTH), all((size_t)rows * GHICP_PQ_RECORD_WIDTH * (size_t)q->world);
  PQ_TRY(ghicp_pairqueue_pack_records(mine.data(), st.data(), (int64_t)mine.size(), rows, block.data()));
  PQ_TRY(ghicp_pairqueue_gather_records(q, block.data(), rows, all.data()));
  for (int64_t p = 0; p < n_pairs; p++) records[(size_t)p * GHICP_PQ_RECORD_WIDTH] = -1.0;
  for (size_t i = 0; i < (size_t)rows * (size_t)q->world; i++) {
    const double* r = &all[i * GHICP_PQ_RECORD_WIDTH];
    if (r[0] >= 0.0 && r[0] < (double)n_pairs) memcpy(records + (size_t)r[0] * GHICP_PQ_RECORD_WIDTH, r, GHICP_PQ_RECORD_WIDTH * sizeof(double));
  }
  for (int64_t p = 0; p < n_pairs; p++)
    if (records[(size_t)p * GHICP_PQ_RECORD_WIDTH] < 0.0) return q->fail(GHICP_ERR_INTERNAL, "pair queue: no rank reported pair %lld", (long long)p);
  if (stats)
    for (size_t i = 0; i < mine.size(); i++) stats[mine[i]] = st[i];
  return GHICP_OK;
}
