// Hand-written device-wide primitives of the front end (gfx950): inclusive scan, select-by-flag and unique over a sorted run
// (round-5 verdict, missing #3 / next-round item 5: these were hipcub::DeviceSelect / rocPRIM calls -- two launches each plus a
// temporary-storage query, a look-back state kernel and runtime fill kernels in every profile).
// All three are the same three launches over tiles of TILE items per 256-thread workgroup:
//   (1) tile totals                     -- one pass over the input, wave ballots / DPP adds, one number per tile
//   (2) scan of the tile totals         -- a single workgroup (<= 2^28 / TILE = 64 K totals)
//   (3) tile-local scan + tile offset   -- second pass over the input, outputs written in order (stable, ascending)
// Reading the input twice costs less than the host round trips and look-back spinning it replaces at these sizes (1-60 M items, HBM
// bound either way), and the result is position-exact: item i of the output is the i-th flagged / distinct input.
#include "prims.h"

namespace {

constexpr int PT = 256;           // threads per workgroup
constexpr int PI = 16;            // items per thread
constexpr int TILE = PT * PI;     // 4096 items per tile

__device__ inline unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  return v;
}

// exclusive prefix of `v` over the workgroup (sh: 4 words); returns the prefix, *total = workgroup sum
__device__ inline unsigned block_excl_scan(unsigned v, unsigned* sh, unsigned* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned inc = wave_incl_scan(v, lane);
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < wave; w++) base += sh[w];
  *total = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return base + inc - v;
}

// ---- (1) tile totals.  MODE 0: sum of u32 values; 1: number of non-zero flag bytes; 2: number of run heads of a sorted u32 array
template <int MODE>
__global__ __launch_bounds__(PT) void k_tile_totals(const void* __restrict__ in, long long n, unsigned* __restrict__ totals) {
  __shared__ unsigned sh[4];
  const long long t0 = (long long)blockIdx.x * TILE;
  unsigned acc = 0;
#pragma unroll 4
  for (int k = 0; k < PI; k++) {
    const long long i = t0 + (long long)k * PT + threadIdx.x;
    if (i < n) {
      if (MODE == 0) acc += reinterpret_cast<const unsigned*>(in)[i];
      else if (MODE == 1) acc += reinterpret_cast<const unsigned char*>(in)[i] ? 1u : 0u;
      else {
        const unsigned* key = reinterpret_cast<const unsigned*>(in);
        acc += (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
      }
    }
  }
  unsigned tot;
  block_excl_scan(acc, sh, &tot);
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

// ---- (2) exclusive scan of the tile totals in place, one workgroup; totals[nt] = grand total (also to *count when given)
__global__ __launch_bounds__(PT) void k_scan_totals(unsigned* __restrict__ totals, int nt, int* __restrict__ count) {
  __shared__ unsigned sh[4];
  unsigned carry = 0;
  for (int base = 0; base < nt; base += PT) {
    const int i = base + threadIdx.x;
    const unsigned v = i < nt ? totals[i] : 0u;
    unsigned tot;
    const unsigned ex = block_excl_scan(v, sh, &tot);
    if (i < nt) totals[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    totals[nt] = carry;
    if (count) *count = (int)carry;
  }
}

// ---- (3a) inclusive scan in place.  A thread owns PI CONSECUTIVE items so that one prefix per thread suffices.
__global__ __launch_bounds__(PT) void k_scan_apply(unsigned* __restrict__ data, long long n, const unsigned* __restrict__ totals) {
  __shared__ unsigned sh[4];
  const long long t0 = (long long)blockIdx.x * TILE + (long long)threadIdx.x * PI;
  unsigned v[PI];
  unsigned acc = 0;
  const bool whole = t0 + PI <= n;  // 64 bytes per thread, 16-byte aligned (tiles and thread offsets are multiples of 16 items)
  if (whole) {
#pragma unroll
    for (int q = 0; q < PI / 4; q++) {
      const uint4 u = reinterpret_cast<const uint4*>(data + t0)[q];
      v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < PI; k++) v[k] = t0 + k < n ? data[t0 + k] : 0u;
  }
#pragma unroll
  for (int k = 0; k < PI; k++) acc += v[k];
  unsigned tot;
  unsigned run = totals[blockIdx.x] + block_excl_scan(acc, sh, &tot);
#pragma unroll
  for (int k = 0; k < PI; k++) {
    run += v[k];
    v[k] = run;
  }
  if (whole) {
#pragma unroll
    for (int q = 0; q < PI / 4; q++) reinterpret_cast<uint4*>(data + t0)[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < PI; k++)
      if (t0 + k < n) data[t0 + k] = v[k];
  }
}

// ---- (3b) select: out[rank] = i for every flagged (MODE 1) / run-head (MODE 2) position i, ascending; MODE 2 writes the KEY instead
// (unique).  Items are taken wave-contiguously (a wave instruction covers 64 consecutive items), ranks from ballots.
template <int MODE>
__global__ __launch_bounds__(PT) void k_select_apply(const void* __restrict__ in, long long n, const unsigned* __restrict__ totals, unsigned* __restrict__ out,
                                                     const unsigned* __restrict__ vals) {
  __shared__ unsigned sh_w[4][PI];  // per wave and round: hits of that wave in that round
  __shared__ unsigned sh_base[4][PI];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t0 = (long long)blockIdx.x * TILE;
  // item (round k, wave w, lane l) = t0 + (k * 4 + w) * 64 + l: ascending in (k, w, l)
  unsigned long long hit[PI];
#pragma unroll
  for (int k = 0; k < PI; k++) {
    const long long i = t0 + (long long)(k * 4 + wave) * 64 + lane;
    bool h = false;
    if (i < n) {
      if (MODE == 1) h = reinterpret_cast<const unsigned char*>(in)[i] != 0;
      else {
        const unsigned* key = reinterpret_cast<const unsigned*>(in);
        h = i == 0 || key[i] != key[i - 1];
      }
    }
    hit[k] = __ballot(h);
    if (lane == 0) sh_w[wave][k] = (unsigned)__popcll(hit[k]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // 64 numbers: serial prefix in (k, w) order
    unsigned run = totals[blockIdx.x];
    for (int k = 0; k < PI; k++)
      for (int w = 0; w < 4; w++) { sh_base[w][k] = run; run += sh_w[w][k]; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PI; k++) {
    if ((hit[k] >> lane) & 1ull) {
      const long long i = t0 + (long long)(k * 4 + wave) * 64 + lane;
      const unsigned pos = sh_base[wave][k] + (unsigned)__popcll(hit[k] & ((1ull << lane) - 1ull));
      out[pos] = MODE == 1 ? (vals ? vals[i] : (unsigned)i) : reinterpret_cast<const unsigned*>(in)[i];
    }
  }
}

int tiles_of(long long n) { return (int)((n + TILE - 1) / TILE); }

template <int MODE>
int select_impl(ghicp_ctx* ctx, const void* in, long long n, unsigned* out, int* d_count, const unsigned* vals = nullptr) {
  hipStream_t s = ctx->stream;
  if (n <= 0) {
    if (d_count) GH_HIP(hipMemsetAsync(d_count, 0, sizeof(int), s));
    return GHICP_OK;
  }
  const int nt = tiles_of(n);
  unsigned* totals;
  GH_TRY(ctx->reserve(B_PRIM_TMP, (size_t)nt + 2, &totals));
  hipLaunchKernelGGL((k_tile_totals<MODE>), dim3(nt), dim3(PT), 0, s, in, n, totals);
  hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(PT), 0, s, totals, nt, d_count);
  hipLaunchKernelGGL((k_select_apply<MODE>), dim3(nt), dim3(PT), 0, s, in, n, (const unsigned*)totals, out, vals);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

}  // namespace

int gh_scan_inclusive_u32(ghicp_ctx* ctx, unsigned* data, long long n) {
  if (n <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  const int nt = tiles_of(n);
  unsigned* totals;
  GH_TRY(ctx->reserve(B_PRIM_TMP, (size_t)nt + 2, &totals));
  hipLaunchKernelGGL((k_tile_totals<0>), dim3(nt), dim3(PT), 0, s, (const void*)data, n, totals);
  hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(PT), 0, s, totals, nt, (int*)nullptr);
  hipLaunchKernelGGL(k_scan_apply, dim3(nt), dim3(PT), 0, s, data, n, (const unsigned*)totals);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

int gh_select_flagged_iota(ghicp_ctx* ctx, const unsigned char* flags, long long n, int* out_idx, int* d_count) {
  return select_impl<1>(ctx, flags, n, reinterpret_cast<unsigned*>(out_idx), d_count);
}

int gh_select_flagged_u32(ghicp_ctx* ctx, const unsigned* vals, const unsigned char* flags, long long n, unsigned* out, int* d_count) {
  return select_impl<1>(ctx, flags, n, out, d_count, vals);
}

int gh_unique_sorted_u32(ghicp_ctx* ctx, const unsigned* keys, long long n, unsigned* out, int* d_count) {
  return select_impl<2>(ctx, keys, n, out, d_count);
}
