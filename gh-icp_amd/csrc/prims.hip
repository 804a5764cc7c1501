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

#include <algorithm>

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

// ================================================================================ stable LSD radix sort (round 6)
// Replaces rocPRIM's Onesweep for the voxel keys, the two grids' cell keys and the single-cloud NMS order (filter.hpp:66,
// keypoint_detect.hpp:119-130: the reference's std::sort calls; stability = "lowest input index leads its voxel / cell").
// One pass per 8-bit digit place, four launches per pass over the same 4096-item tiles as the scan / select above:
//   k_rs_hist     digit counts of every tile                      -> table[tile][digit]   (1 KB per tile, written coalesced)
//   k_rs_chunks   per digit, exclusive scan over a chunk of tiles -> table (in place), chunk totals part[chunk][digit]
//   k_rs_bases    one workgroup: chunk totals -> chunk bases per digit, digit totals -> digit bases
//   k_rs_scatter  ranks inside the tile from wave ballots (items in input order: wave-contiguous chunks, a wave's rounds in turn), the
//                 tile reordered by digit in LDS, then written out as runs -- consecutive threads store consecutive addresses of a run
// (three launches per place up to 64 tiles: k_rs_chunk_bases, and up to 32 chunks = 8 M items: the scatter sums the chunk rows itself; ONE launch for all
// places up to one tile: k_rs_small)
// No look-back, no spinning, no temporary-storage query; position-exact, so the result does not depend on the launch order of workgroups.
constexpr int RD = 256;   // digits per place
constexpr int RC = 64;    // tiles per chunk of k_rs_chunks

template <typename K>
__global__ __launch_bounds__(PT) void k_rs_hist(const K* __restrict__ keys, long long n, int shift, unsigned mask, unsigned* __restrict__ table) {
  __shared__ unsigned cnt[4][RD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < 4; w++) cnt[w][threadIdx.x] = 0u;
  __syncthreads();
  const long long t0 = (long long)blockIdx.x * TILE;
  K key[PI];  // all loads of the tile in flight before the first LDS atomic
#pragma unroll
  for (int k = 0; k < PI; k++) {
    const long long i = t0 + (long long)(wave * PI + k) * 64 + lane;
    key[k] = i < n ? keys[i] : (K)0;
  }
#pragma unroll
  for (int k = 0; k < PI; k++) {
    const long long i = t0 + (long long)(wave * PI + k) * 64 + lane;
    if (i < n) atomicAdd(&cnt[wave][(unsigned)(key[k] >> shift) & mask], 1u);
  }
  __syncthreads();
  table[(size_t)blockIdx.x * RD + threadIdx.x] = cnt[0][threadIdx.x] + cnt[1][threadIdx.x] + cnt[2][threadIdx.x] + cnt[3][threadIdx.x];
}

// thread = digit; a workgroup walks RC consecutive tiles (rows of 1 KB, coalesced)
__global__ __launch_bounds__(RD) void k_rs_chunks(unsigned* __restrict__ table, int nt, unsigned* __restrict__ part) {
  const int t0 = blockIdx.x * RC, t1 = t0 + RC < nt ? t0 + RC : nt;
  unsigned run = 0;
#pragma unroll 8
  for (int t = t0; t < t1; t++) {
    const unsigned v = table[(size_t)t * RD + threadIdx.x];
    table[(size_t)t * RD + threadIdx.x] = run;
    run += v;
  }
  part[(size_t)blockIdx.x * RD + threadIdx.x] = run;
}

__global__ __launch_bounds__(RD) void k_rs_bases(unsigned* __restrict__ part, int nchunk, unsigned* __restrict__ digit_base) {
  __shared__ unsigned sh[4];
  unsigned run = 0;
#pragma unroll 8
  for (int j = 0; j < nchunk; j++) {
    const unsigned v = part[(size_t)j * RD + threadIdx.x];
    part[(size_t)j * RD + threadIdx.x] = run;
    run += v;
  }
  unsigned tot;
  digit_base[threadIdx.x] = block_excl_scan(run, sh, &tot);
}

// One digit place of one tile, by the whole workgroup: the items (key[k], val[k] of item (wave * PI + k) * 64 + lane, `here` of them) end up in
// T.lk / T.lv in stable order of the digit, T.first[d] = first position of digit d.  Ranks from wave ballots: a wave owns PI * 64 consecutive
// items and takes them 64 at a time; the lanes of a round that share a digit find each other with eight ballots, the lowest of them bumps the
// wave's counter of that digit (LDS, that one lane only) and hands the old value round.  Ends with a barrier.
template <typename K, bool HASV>
struct RsTile {
  unsigned cnt[4][RD];  // per wave: running count of a digit, then the wave's base inside the digit's run of this tile
  unsigned first[RD];
  unsigned sh[4];
  K lk[TILE];
  unsigned lv[HASV ? TILE : 1];
};
template <typename K, bool HASV>
__device__ inline void rs_tile_sort(RsTile<K, HASV>& T, const K (&key)[PI], const unsigned (&val)[PI], int here, int shift, unsigned mask) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < 4; w++) T.cnt[w][threadIdx.x] = 0u;
  __syncthreads();
  unsigned rk[PI];
#pragma unroll
  for (int k = 0; k < PI; k++) {
    const bool valid = (wave * PI + k) * 64 + lane < here;
    const unsigned d = (unsigned)(key[k] >> shift) & mask;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int leader = valid ? (int)__ffsll(peers) - 1 : lane;
    unsigned base = 0;
    if (valid && lane == leader) {
      base = T.cnt[wave][d];
      T.cnt[wave][d] = base + (unsigned)__popcll(peers);
    }
    base = __shfl(base, leader, 64);
    rk[k] = base + (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  {  // thread = digit: the waves' bases inside the digit's run, the run's place in the tile
    const int d = threadIdx.x;
    const unsigned c0 = T.cnt[0][d], c1 = T.cnt[1][d], c2 = T.cnt[2][d], c3 = T.cnt[3][d];
    T.cnt[0][d] = 0u; T.cnt[1][d] = c0; T.cnt[2][d] = c0 + c1; T.cnt[3][d] = c0 + c1 + c2;
    unsigned tot;
    T.first[d] = block_excl_scan(c0 + c1 + c2 + c3, T.sh, &tot);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PI; k++) {
    if ((wave * PI + k) * 64 + lane < here) {
      const unsigned d = (unsigned)(key[k] >> shift) & mask;
      const unsigned p = T.first[d] + T.cnt[wave][d] + rk[k];
      T.lk[p] = key[k];
      if (HASV) T.lv[p] = val[k];
    }
  }
  __syncthreads();
}

// BASES: the launch has no k_rs_bases before it (up to RS_INLINE_CHUNKS chunks): `part` holds the chunks' raw totals and every workgroup sums the
// few rows it needs itself (thread = digit; the rows come out of L2) -- a launch less per digit place where launches are what a place costs.
constexpr int RS_INLINE_CHUNKS = 32;
template <typename K, bool HASV, bool BASES>
__global__ __launch_bounds__(PT) void k_rs_scatter(const K* __restrict__ kin, const unsigned* __restrict__ vin, K* __restrict__ kout, unsigned* __restrict__ vout,
                                                   long long n, int shift, unsigned mask, const unsigned* __restrict__ table, const unsigned* __restrict__ part,
                                                   const unsigned* __restrict__ digit_base, int nchunk) {
  __shared__ RsTile<K, HASV> T;
  __shared__ unsigned s_delta[RD];  // global position of the digit's run of this tile - its first position in the tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t0 = (long long)blockIdx.x * TILE;
  const int here = n - t0 < (long long)TILE ? (int)(n - t0) : TILE;
  K key[PI];
  unsigned val[PI];
#pragma unroll
  for (int k = 0; k < PI; k++) {
    const long long i = t0 + (long long)(wave * PI + k) * 64 + lane;
    key[k] = i < n ? kin[i] : (K)0;
    val[k] = (HASV && i < n) ? vin[i] : 0u;
  }
  unsigned run_at = table[(size_t)blockIdx.x * RD + threadIdx.x];  // thread = digit
  if (BASES) {
    const int mine = (int)blockIdx.x / RC;
    unsigned below = 0, total = 0;
    for (int j = 0; j < nchunk; j++) {
      const unsigned v = part[(size_t)j * RD + threadIdx.x];
      total += v;
      below += j < mine ? v : 0u;
    }
    unsigned tot;
    run_at += below + block_excl_scan(total, T.sh, &tot);
  } else {
    run_at += digit_base[threadIdx.x] + part[(size_t)(blockIdx.x / RC) * RD + threadIdx.x];
  }
  rs_tile_sort<K, HASV>(T, key, val, here, shift, mask);
  s_delta[threadIdx.x] = run_at - T.first[threadIdx.x];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PI; j++) {
    const int p = j * PT + threadIdx.x;
    if (p < here) {
      const K kk = T.lk[p];
      const unsigned pos = s_delta[(unsigned)(kk >> shift) & mask] + (unsigned)p;
      kout[pos] = kk;
      if (HASV) vout[pos] = T.lv[p];
    }
  }
}

// n <= TILE: every digit place in ONE launch of one workgroup (the single-cloud front end's NMS order, small grids): the items stay in
// registers between the places and go through the tile once per place.
template <typename K, bool HASV>
__global__ __launch_bounds__(PT) void k_rs_small(const K* __restrict__ kin, const unsigned* __restrict__ vin, K* __restrict__ kout, unsigned* __restrict__ vout, int n,
                                                 int bit_begin, int bit_end) {
  __shared__ RsTile<K, HASV> T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  K key[PI];
  unsigned val[PI];
#pragma unroll
  for (int k = 0; k < PI; k++) {
    const int i = (wave * PI + k) * 64 + lane;
    key[k] = i < n ? kin[i] : (K)0;
    val[k] = (HASV && i < n) ? vin[i] : 0u;
  }
  for (int shift = bit_begin; shift < bit_end; shift += 8) {
    const unsigned mask = (1u << (bit_end - shift < 8 ? bit_end - shift : 8)) - 1u;
    rs_tile_sort<K, HASV>(T, key, val, n, shift, mask);
    if (shift + 8 < bit_end) {
#pragma unroll
      for (int k = 0; k < PI; k++) {
        const int i = (wave * PI + k) * 64 + lane;
        if (i < n) {
          key[k] = T.lk[i];
          if (HASV) val[k] = T.lv[i];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < PI; j++) {
    const int p = j * PT + threadIdx.x;
    if (p < n) {
      kout[p] = T.lk[p];
      if (HASV) vout[p] = T.lv[p];
    }
  }
}

// nt <= RC: k_rs_chunks and k_rs_bases in one launch (one chunk: its base is 0)
__global__ __launch_bounds__(RD) void k_rs_chunk_bases(unsigned* __restrict__ table, int nt, unsigned* __restrict__ part, unsigned* __restrict__ digit_base) {
  __shared__ unsigned sh[4];
  unsigned run = 0;
#pragma unroll 8
  for (int t = 0; t < nt; t++) {
    const unsigned v = table[(size_t)t * RD + threadIdx.x];
    table[(size_t)t * RD + threadIdx.x] = run;
    run += v;
  }
  part[threadIdx.x] = 0u;
  unsigned tot;
  digit_base[threadIdx.x] = block_excl_scan(run, sh, &tot);
}

template <typename K>
int radix_sort_impl(ghicp_ctx* ctx, const K* kin, K* kout, const unsigned* vin, unsigned* vout, long long n, int bit_begin, int bit_end) {
  if (n <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  const bool hasv = vin != nullptr;
  const int passes = bit_end > bit_begin ? (bit_end - bit_begin + 7) / 8 : 0;
  if (passes == 0) {
    GH_HIP(hipMemcpyAsync(kout, kin, (size_t)n * sizeof(K), hipMemcpyDeviceToDevice, s));
    if (hasv) GH_HIP(hipMemcpyAsync(vout, vin, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, s));
    return GHICP_OK;
  }
  if (n <= TILE) {
    if (hasv) hipLaunchKernelGGL((k_rs_small<K, true>), dim3(1), dim3(PT), 0, s, kin, vin, kout, vout, (int)n, bit_begin, bit_end);
    else hipLaunchKernelGGL((k_rs_small<K, false>), dim3(1), dim3(PT), 0, s, kin, (const unsigned*)nullptr, kout, (unsigned*)nullptr, (int)n, bit_begin, bit_end);
    GH_HIP(hipGetLastError());
    return GHICP_OK;
  }
  const int nt = tiles_of(n), nchunk = (nt + RC - 1) / RC;
  unsigned* table;
  GH_TRY(ctx->reserve(B_PRIM_TMP, (size_t)nt * RD + (size_t)nchunk * RD + RD + 4, &table));
  unsigned* part = table + (size_t)nt * RD;
  unsigned* digit_base = part + (size_t)nchunk * RD;
  // the input is left alone: passes alternate between a spare buffer and the output so that the last one lands in the output
  K* spare_k = nullptr;
  unsigned* spare_v = nullptr;
  if (passes > 1) {
    char* sp;
    const size_t kb = (((size_t)n * sizeof(K) + 255) / 256) * 256;
    GH_TRY(ctx->reserve(B_GRID_TMP, kb + (hasv ? (size_t)n * sizeof(unsigned) : 0) + 16, &sp));
    spare_k = reinterpret_cast<K*>(sp);
    spare_v = reinterpret_cast<unsigned*>(sp + kb);
  }
  const K* src_k = kin;
  const unsigned* src_v = vin;
  for (int p = 0; p < passes; p++) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    K* dst_k = to_out ? kout : spare_k;
    unsigned* dst_v = to_out ? vout : spare_v;
    const int shift = bit_begin + 8 * p;
    const unsigned mask = (1u << std::min(8, bit_end - shift)) - 1u;
    hipLaunchKernelGGL((k_rs_hist<K>), dim3(nt), dim3(PT), 0, s, src_k, n, shift, mask, table);
    const bool inline_bases = nchunk > 1 && nchunk <= RS_INLINE_CHUNKS;
    if (nchunk == 1) hipLaunchKernelGGL(k_rs_chunk_bases, dim3(1), dim3(RD), 0, s, table, nt, part, digit_base);
    else {
      hipLaunchKernelGGL(k_rs_chunks, dim3(nchunk), dim3(RD), 0, s, table, nt, part);
      if (!inline_bases) hipLaunchKernelGGL(k_rs_bases, dim3(1), dim3(RD), 0, s, part, nchunk, digit_base);
    }
    const unsigned* tb = table;
    const unsigned* pt = part;
    const unsigned* db = digit_base;
    const unsigned* nov = nullptr;
    unsigned* novo = nullptr;
    if (hasv && inline_bases) hipLaunchKernelGGL((k_rs_scatter<K, true, true>), dim3(nt), dim3(PT), 0, s, src_k, src_v, dst_k, dst_v, n, shift, mask, tb, pt, db, nchunk);
    else if (hasv) hipLaunchKernelGGL((k_rs_scatter<K, true, false>), dim3(nt), dim3(PT), 0, s, src_k, src_v, dst_k, dst_v, n, shift, mask, tb, pt, db, nchunk);
    else if (inline_bases) hipLaunchKernelGGL((k_rs_scatter<K, false, true>), dim3(nt), dim3(PT), 0, s, src_k, nov, dst_k, novo, n, shift, mask, tb, pt, db, nchunk);
    else hipLaunchKernelGGL((k_rs_scatter<K, false, false>), dim3(nt), dim3(PT), 0, s, src_k, nov, dst_k, novo, n, shift, mask, tb, pt, db, nchunk);
    src_k = dst_k;
    src_v = dst_v;
  }
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}



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

int gh_radix_sort_u32(ghicp_ctx* ctx, const unsigned* keys_in, unsigned* keys_out, const unsigned* vals_in, unsigned* vals_out, long long n, int bit_begin,
                      int bit_end) {
  return radix_sort_impl<unsigned>(ctx, keys_in, keys_out, vals_in, vals_out, n, bit_begin, bit_end);
}

int gh_radix_sort_u64(ghicp_ctx* ctx, const unsigned long long* keys_in, unsigned long long* keys_out, const unsigned* vals_in, unsigned* vals_out, long long n,
                      int bit_begin, int bit_end) {
  return radix_sort_impl<unsigned long long>(ctx, keys_in, keys_out, vals_in, vals_out, n, bit_begin, bit_end);
}

extern "C" int ghicp_sort_pairs(ghicp_ctx* ctx, int key_bytes, const void* keys_in, void* keys_out, const uint32_t* vals_in, uint32_t* vals_out, int64_t n,
                                int bit_begin, int bit_end) {
  GH_ENTER(ctx);
  GH_ARG((key_bytes == 4 || key_bytes == 8) && n >= 0 && n < (1ll << 31) - 2 && bit_begin >= 0 && bit_begin <= bit_end && bit_end <= 8 * key_bytes);
  GH_ARG((vals_in == nullptr) == (vals_out == nullptr) && (n == 0 || (keys_in != nullptr && keys_out != nullptr && keys_in != keys_out)));
  if (n == 0) return GHICP_OK;
  Stager sg(ctx);
  const unsigned* vi = nullptr;
  unsigned* vo = nullptr;
  if (vals_in) {
    GH_TRY(sg.in(vals_in, (size_t)n, &vi));
    GH_TRY(sg.out(vals_out, (size_t)n, &vo));
  }
  if (key_bytes == 4) {
    const unsigned* ki;
    unsigned* ko;
    GH_TRY(sg.in(static_cast<const unsigned*>(keys_in), (size_t)n, &ki));
    GH_TRY(sg.out(static_cast<unsigned*>(keys_out), (size_t)n, &ko));
    GH_TRY(gh_radix_sort_u32(ctx, ki, ko, vi, vo, n, bit_begin, bit_end));
  } else {
    const unsigned long long* ki;
    unsigned long long* ko;
    GH_TRY(sg.in(static_cast<const unsigned long long*>(keys_in), (size_t)n, &ki));
    GH_TRY(sg.out(static_cast<unsigned long long*>(keys_out), (size_t)n, &ko));
    GH_TRY(gh_radix_sort_u64(ctx, ki, ko, vi, vo, n, bit_begin, bit_end));
  }
  return sg.finish();
}
