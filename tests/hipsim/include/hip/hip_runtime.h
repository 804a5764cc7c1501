// TEST INFRASTRUCTURE -- not part of the product, never loaded by gh-icp_amd/.
// hipsim: a host-side SIMT interpreter for the kernels of gh-icp_amd/csrc/*.hip.  The SAME kernel sources are compiled with g++ against
// this header (tests/hipsim/build.py -> tests/hipsim/_build/libghicp_sim.so); every lane of a workgroup is a fiber, wave collectives
// (__ballot, __shfl*, readlane, wave_barrier) and __syncthreads are scheduling points of the block scheduler in hipsim.cpp.  Purpose: run
// the GPU parity tests' logic on the build container (which has no GPU) while a kernel is being written, so that the scarce MI355X
// minutes go to measurement.  It says nothing about speed, memory ordering or occupancy, and no parity claim rests on it: parity is what
// `pytest -m gpu` shows on the MI355X through libghicp_hip.so.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <type_traits>

#define HIPSIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ thread_local  // one OS thread runs one workgroup at a time: its thread-local storage is the workgroup's LDS
#define __constant__ static
#define __align__(n) __attribute__((aligned(n)))

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
struct float3 { float x, y, z; };
struct float2 { float x, y; };
struct __attribute__((aligned(8))) int2 { int x, y; };
struct __attribute__((aligned(8))) uint2 { unsigned x, y; };
struct __attribute__((aligned(16))) double2 { double x, y; };
struct __attribute__((aligned(16))) int4 { int x, y, z, w; };
struct __attribute__((aligned(16))) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ------------------------------------------------------------------------------------------------ lane / block state
namespace hipsim {
enum WaveOp { OP_BALLOT = 1, OP_SHFL = 2, OP_BARRIER = 3, OP_FIRST = 4, OP_MFMA_I32_32X32X32_I8 = 5 };
struct MfmaI8 { signed char a[16], b[16]; int c[16]; };  // one lane's operands of v_mfma_i32_32x32x32_i8 (payload of the collective = its address)
struct Lane {
  void* sp;
  uint3 tid;
  int state, lane, wave;
  int op, site, src;       // pending wave collective
  uint64_t payload, result;
};
struct BlockInfo { uint3 bid; dim3 bdim, gdim; };
extern thread_local Lane* cur;
extern thread_local BlockInfo binfo;
uint64_t wave_collective(int op, int site, uint64_t payload, int src);
void sync_threads();
void yield();
// smem_a / smem_b: return the calling OS thread's instances of the two candidates for `extern __shared__` (see `smem` below); the launch puts a
// guard pattern behind the `shmem` bytes the launch asked for and checks it after every workgroup: a kernel that uses more dynamic LDS than
// its launch requested aborts the process with a message (on the GPU that is silent corruption of a neighbour's LDS or a fault)
void launch(dim3 grid, dim3 block, size_t shmem, void (*tramp)(void*), void* closure, char* (*smem_a)() = nullptr, char* (*smem_b)() = nullptr);
enum { C_MEMCPY = 0, C_MEMSET, C_SYNC, C_LIBCALL, C_MALLOC, C_NUM };
void count(int what);  // host API calls, for "operations per cloud" bookkeeping (hipsim_counters)
}  // namespace hipsim

#define threadIdx (hipsim::cur->tid)
#define blockIdx (hipsim::binfo.bid)
#define blockDim (hipsim::binfo.bdim)
#define gridDim (hipsim::binfo.gdim)
static const int warpSize = 64;

// dynamic LDS: `extern __shared__ char smem[]` in a kernel resolves to one of these (global or the TU's unnamed namespace)
inline thread_local __attribute__((aligned(64))) char smem[160 * 1024 + 256];  // (+ 256: room for the guard when a launch asks for all 160 KB)
namespace { thread_local __attribute__((aligned(64), unused)) char smem[160 * 1024 + 256]; }
inline char* hipsim_smem_global() { return ::smem; }
namespace { __attribute__((unused)) char* hipsim_smem_tu() { return smem; } }

// ------------------------------------------------------------------------------------------------ synchronisation, collectives
static inline void __syncthreads() { hipsim::sync_threads(); }
static inline void __threadfence_block() {}
static inline void __threadfence() {}
static inline void __builtin_amdgcn_wave_barrier(int line = __builtin_LINE()) { hipsim::wave_collective(hipsim::OP_BARRIER, line, 0, 0); }
static inline void __builtin_amdgcn_s_barrier() { hipsim::sync_threads(); }
static inline void __builtin_amdgcn_s_sleep(int) { hipsim::yield(); }
static inline void __builtin_amdgcn_s_setprio(int) {}  // issue arbitration between the waves of a SIMD: nothing to model
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }  // HW_ID / XCC_ID (where a wave runs: diagnostics of the pair loop): no hardware here
static inline unsigned long long __ballot(int pred, int line = __builtin_LINE()) {
  return hipsim::wave_collective(hipsim::OP_BALLOT, line, pred ? 1 : 0, 0);
}
static inline int __any(int pred, int line = __builtin_LINE()) { return __ballot(pred, line) != 0; }
static inline unsigned long long __activemask(int line = __builtin_LINE()) { return __ballot(1, line); }
namespace hipsim {
template <class T> static inline T shfl_from(T v, int src, int line) {
  static_assert(sizeof(T) <= 8, "shuffle of a type wider than 64 bits");
  uint64_t p = 0;
  memcpy(&p, &v, sizeof(T));
  const uint64_t r = wave_collective(OP_SHFL, line, p, src);
  T out;
  memcpy(&out, &r, sizeof(T));
  return out;
}
}  // namespace hipsim
template <class T> static inline T __shfl(T v, int src_lane, int width = 64, int line = __builtin_LINE()) {
  const int lane = hipsim::cur->lane;
  return hipsim::shfl_from(v, (lane & ~(width - 1)) + (src_lane & (width - 1)), line);
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64, int line = __builtin_LINE()) {
  const int lane = hipsim::cur->lane, src = lane ^ mask;
  return hipsim::shfl_from(v, (src / width == lane / width) ? src : lane, line);
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64, int line = __builtin_LINE()) {
  const int lane = hipsim::cur->lane, src = lane + (int)delta;
  return hipsim::shfl_from(v, (src / width == lane / width && src < 64) ? src : lane, line);
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64, int line = __builtin_LINE()) {
  const int lane = hipsim::cur->lane, src = lane - (int)delta;
  return hipsim::shfl_from(v, (src >= 0 && src / width == lane / width) ? src : lane, line);
}
static inline int __builtin_amdgcn_readlane(int v, int lane, int line = __builtin_LINE()) { return hipsim::shfl_from(v, lane, line); }
// v_mov_b32_dpp with row_mask = bank_mask = 0xf: the source lane of every lane for the controls the kernels use (quad_perm, row_shl /
// row_shr / row_ror, row_mirror, row_half_mirror); a lane without a valid source keeps `old` (bound_ctrl = false) or reads 0 (true)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int line = __builtin_LINE()) {
  const int lane = hipsim::cur->lane, row = lane & ~15, r = lane & 15;
  int from = -1;
  if (ctrl < 0x100) from = (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl >= 0x101 && ctrl <= 0x10f) from = r + (ctrl & 15) < 16 ? lane + (ctrl & 15) : -1;
  else if (ctrl >= 0x111 && ctrl <= 0x11f) from = r - (ctrl & 15) >= 0 ? lane - (ctrl & 15) : -1;
  else if (ctrl >= 0x121 && ctrl <= 0x12f) from = row + ((r - (ctrl & 15)) & 15);
  else if (ctrl == 0x140) from = row + 15 - r;
  else if (ctrl == 0x141) from = (lane & ~7) + 7 - (lane & 7);
  else { fprintf(stderr, "hipsim: dpp control 0x%x is not modelled\n", ctrl); abort(); }
  if (row_mask != 0xf || bank_mask != 0xf) { fprintf(stderr, "hipsim: dpp row/bank masks are not modelled\n"); abort(); }
  const int got = hipsim::shfl_from(src, from < 0 ? lane : from, line);
  return from < 0 ? (bound_ctrl ? 0 : old) : got;
}
// v_mfma_i32_32x32x32_i8: D = A B + C over a 32 x 32 x 32 block held by the 64 lanes of a wave.  Lane l supplies 16 int8 of A's row (l & 31) and
// of B's column (l & 31) at K-offset (l >> 5) * 16; C/D register r of lane l is element (row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31).
typedef int hipsim_v4i __attribute__((vector_size(16)));
typedef int hipsim_v16i __attribute__((vector_size(64)));
static inline hipsim_v16i __builtin_amdgcn_mfma_i32_32x32x32_i8(hipsim_v4i a, hipsim_v4i b, hipsim_v16i c, int, int, int, int line = __builtin_LINE()) {
  hipsim::MfmaI8 m;
  __builtin_memcpy(m.a, &a, 16); __builtin_memcpy(m.b, &b, 16); __builtin_memcpy(m.c, &c, 64);
  hipsim::wave_collective(hipsim::OP_MFMA_I32_32X32X32_I8, line, (uint64_t)(uintptr_t)&m, 0);
  __builtin_memcpy(&c, m.c, 64);
  return c;
}
static inline int __builtin_amdgcn_readfirstlane(int v, int line = __builtin_LINE()) {
  return (int)(uint32_t)hipsim::wave_collective(hipsim::OP_FIRST, line, (uint32_t)v, 0);
}
static inline unsigned long long __builtin_amdgcn_s_memtime() {
  return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline unsigned long long __builtin_readcyclecounter() { return __builtin_amdgcn_s_memtime(); }
static inline unsigned long long __builtin_amdgcn_s_memrealtime() { return __builtin_amdgcn_s_memtime() / 10; }
static inline unsigned long long wall_clock64() { return __builtin_amdgcn_s_memrealtime(); }
static inline long long clock64() { return (long long)__builtin_amdgcn_s_memtime(); }

// ------------------------------------------------------------------------------------------------ bit / conversion intrinsics
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
#define HIPSIM_BITCAST(name, From, To) static inline To name(From v) { To r; memcpy(&r, &v, sizeof(r)); return r; }
HIPSIM_BITCAST(__float_as_uint, float, unsigned)
HIPSIM_BITCAST(__float_as_int, float, int)
HIPSIM_BITCAST(__uint_as_float, unsigned, float)
HIPSIM_BITCAST(__int_as_float, int, float)
HIPSIM_BITCAST(__double_as_longlong, double, long long)
HIPSIM_BITCAST(__longlong_as_double, long long, double)
static inline int __double2hiint(double v) { return (int)(uint32_t)((uint64_t)__double_as_longlong(v) >> 32); }
static inline int __double2loint(double v) { return (int)(uint32_t)(uint64_t)__double_as_longlong(v); }
static inline double __hiloint2double(int hi, int lo) { return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo)); }
static inline float __fmaf_rn(float a, float b, float c) { return std::fma(a, b, c); }
static inline double __fma_rn(double a, double b, double c) { return std::fma(a, b, c); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
using std::max;
using std::min;
static inline float __ldg(const float* p) { return *p; }

// ------------------------------------------------------------------------------------------------ atomics (blocks run on several OS threads)
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <class T> static inline T __hip_atomic_load(T* p, int, int) { hipsim::yield(); return __atomic_load_n(p, __ATOMIC_SEQ_CST); }  // a poll is a scheduling point
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { hipsim::yield(); return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
template <class T, class V> static inline void __hip_atomic_store(T* p, V v, int, int) { __atomic_store_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class V> static inline T __hip_atomic_fetch_add(T* p, V v, int, int) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class V> static inline T __hip_atomic_fetch_or(T* p, V v, int, int) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class V> static inline T __hip_atomic_fetch_and(T* p, V v, int, int) { return __atomic_fetch_and(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class V> static inline T __hip_atomic_exchange(T* p, V v, int, int) { return __atomic_exchange_n(p, (T)v, __ATOMIC_SEQ_CST); }
namespace hipsim {
template <class T, class F> static inline T atomic_rmw(T* p, F f) {
  using U = std::conditional_t<sizeof(T) == 8, uint64_t, uint32_t>;
  static_assert(sizeof(T) == sizeof(U), "atomic on a type that is neither 32 nor 64 bits");
  U* up = reinterpret_cast<U*>(p);
  U old = __atomic_load_n(up, __ATOMIC_SEQ_CST);
  for (;;) {
    T o;
    memcpy(&o, &old, sizeof(T));
    const T n = f(o);
    U nu;
    memcpy(&nu, &n, sizeof(T));
    if (__atomic_compare_exchange_n(up, &old, nu, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return o;
  }
}
}  // namespace hipsim
template <class T, class V> static inline T __hip_atomic_fetch_min(T* p, V v, int, int) { return hipsim::atomic_rmw(p, [=](T o) { return o < (T)v ? o : (T)v; }); }
template <class T, class V> static inline T __hip_atomic_fetch_max(T* p, V v, int, int) { return hipsim::atomic_rmw(p, [=](T o) { return o > (T)v ? o : (T)v; }); }
template <class T, class V> static inline T atomicAdd(T* p, V v) { return hipsim::atomic_rmw(p, [=](T o) { return (T)(o + (T)v); }); }
template <class T, class V> static inline T atomicSub(T* p, V v) { return hipsim::atomic_rmw(p, [=](T o) { return (T)(o - (T)v); }); }
template <class T, class V> static inline T atomicMin(T* p, V v) { return hipsim::atomic_rmw(p, [=](T o) { return o < (T)v ? o : (T)v; }); }
template <class T, class V> static inline T atomicMax(T* p, V v) { return hipsim::atomic_rmw(p, [=](T o) { return o > (T)v ? o : (T)v; }); }
template <class T, class V> static inline T atomicOr(T* p, V v) { return hipsim::atomic_rmw(p, [=](T o) { return (T)(o | (T)v); }); }
template <class T, class V> static inline T atomicAnd(T* p, V v) { return hipsim::atomic_rmw(p, [=](T o) { return (T)(o & (T)v); }); }
template <class T, class V> static inline T atomicXor(T* p, V v) { return hipsim::atomic_rmw(p, [=](T o) { return (T)(o ^ (T)v); }); }
template <class T, class V> static inline T atomicExch(T* p, V v) { return hipsim::atomic_rmw(p, [=](T) { return (T)v; }); }
template <class T, class V, class W> static inline T atomicCAS(T* p, V cmp, W v) { return hipsim::atomic_rmw(p, [=](T o) { return o == (T)cmp ? (T)v : o; }); }

// ------------------------------------------------------------------------------------------------ host runtime
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidConfiguration = 9, hipErrorNotReady = 600 };
struct hipsimStream { int id; };
typedef hipsimStream* hipStream_t;
struct hipsimEvent { double t_ms; };
typedef hipsimEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t {
  char name[256];
  int multiProcessorCount;
  size_t sharedMemPerBlock, maxSharedMemoryPerMultiProcessor, totalGlobalMem;
  int warpSize, clockRate;
  char gcnArchName[64];
};
static inline double hipsim_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "hipsim (host SIMT interpreter, test infrastructure)");
  strcpy(p->gcnArchName, "gfx950-sim");
  p->multiProcessorCount = 256; p->sharedMemPerBlock = 64 * 1024; p->maxSharedMemoryPerMultiProcessor = 160 * 1024; p->warpSize = 64; p->clockRate = 2400000;
  p->totalGlobalMem = (size_t)64 << 30;
  return hipSuccess;
}
namespace hipsim { hipError_t take_last_error(bool clear); }
static inline hipError_t hipGetLastError() { return hipsim::take_last_error(true); }
static inline hipError_t hipPeekAtLastError() { return hipsim::take_last_error(false); }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorInvalidConfiguration ? "hipErrorInvalidConfiguration (hipsim)" : "hipsim error"; }
namespace hipsim { int poison_byte(); }  // HIPSIM_POISON=<byte>: fresh device memory is filled with it (results must not depend on what hipMalloc returns)
static inline hipError_t hipMalloc(void** p, size_t n) {
  hipsim::count(hipsim::C_MALLOC);
  const size_t bytes = (n + 255) / 256 * 256 + 256;
  *p = aligned_alloc(256, bytes);
  if (*p && hipsim::poison_byte() >= 0) memset(*p, hipsim::poison_byte(), bytes);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), n, f); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { hipsim::count(hipsim::C_MEMCPY); if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
  hipsim::count(hipsim::C_MEMCPY);
  for (size_t r = 0; r < h; r++) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { hipsim::count(hipsim::C_MEMSET); if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipsimStream{1}; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { hipsim::count(hipsim::C_SYNC); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { hipsim::count(hipsim::C_SYNC); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipsimEvent{0.0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t_ms = hipsim_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { hipsim::count(hipsim::C_SYNC); return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int block, size_t lds) {
  int by_lds = lds ? (int)((160 * 1024) / lds) : 32, by_waves = 32 / std::max(1, (block + 63) / 64);
  *n = std::max(0, std::min(by_lds, by_waves));
  return hipSuccess;
}

namespace hipsim {
template <class F> static void closure_tramp(void* c) { (*static_cast<F*>(c))(); }
}
// hipLaunchKernelGGL as a function template: the kernel runs once per lane on that lane's fiber; the launch returns when the grid is done
template <class... KA, class... A>
static inline void hipLaunchKernelGGL(void (*kernel)(KA...), dim3 grid, dim3 block, size_t shmem, hipStream_t, A&&... args) {
  auto body = [&]() { kernel(static_cast<KA>(args)...); };
  hipsim::launch(grid, block, shmem, &hipsim::closure_tramp<decltype(body)>, &body, &hipsim_smem_global, &hipsim_smem_tu);
}
