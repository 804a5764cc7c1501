// Uniform-grid build (cell keys -> stable radix sort -> cell table -> points in cell order) and the
// voxel down-sampling filter.  The sort, scan and select primitives are this library's own (prims.hip); rounds 1-5 called rocPRIM's
// Onesweep here (with its merge-sort limit lowered to 4096 items: the default dispatch cost ~17 launches per sort below 1 M items).
#include "grid.h"
#include "prims.h"

#include <algorithm>
#include <cmath>

namespace {

__global__ __launch_bounds__(256) void k_cell_keys(const float* __restrict__ xyz, long long n, int stride, GridDesc g, unsigned* __restrict__ keys,
                                                   unsigned* __restrict__ vals) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const int cx = gh_cell_coord(xyz[i * stride], g.mn[0], g.inv, g.dim[0]);
  const int cy = gh_cell_coord(xyz[i * stride + 1], g.mn[1], g.inv, g.dim[1]);
  const int cz = gh_cell_coord(xyz[i * stride + 2], g.mn[2], g.inv, g.dim[2]);
  keys[i] = ((unsigned)cx * g.dim[1] + cy) * g.dim[2] + cz;
  vals[i] = (unsigned)i;
}

// The cell table start[c] = lower_bound(keys, c), c = 0 .. ncell, FROM THE SORTED KEYS (round 5): the first point of every run of equal
// keys fills the cells (previous key, its key] with its position -- the table is written once, coalesced, and nothing is searched.
// (Rounds 1-4 ran a binary search over the keys per CELL: 23 dependent loads for each of the tens of millions of cells of a batch, most
// of them empty: 0.45 ms per 32 clouds, more bytes read than the whole PCA stage needs -- round-4 verdict, weak #5.)
// Short gaps are filled by the wave, one after the other; a gap of 2048 cells or more (empty space between surfaces, the boundary of two
// clouds of a batch) is filled by the whole workgroup; the cells beyond the last key by k_cell_start_tail.
__global__ __launch_bounds__(256) void k_cell_start_fill(const unsigned* __restrict__ keys, unsigned n, unsigned* __restrict__ start) {
  __shared__ unsigned s_lo[256], s_hi[256], s_ix[256];
  __shared__ int s_cnt;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  bool bd = false;
  unsigned lo = 0, hi = 0;
  if (i < n) {
    const unsigned k = keys[i];
    if (i == 0) bd = true;
    else {
      const unsigned kp = keys[i - 1];
      bd = k != kp;
      lo = kp + 1u;
    }
    hi = k;
  }
  const bool far = bd && (hi - lo >= 2048u);
  unsigned long long m = __ballot(bd && !far);
  while (m) {
    const int l = (int)__ffsll((long long)m) - 1;
    m &= m - 1ull;
    const unsigned glo = (unsigned)__builtin_amdgcn_readlane((int)lo, l), ghi = (unsigned)__builtin_amdgcn_readlane((int)hi, l);
    const unsigned gix = i - (unsigned)lane + (unsigned)l;
    for (unsigned c = glo + (unsigned)lane; c <= ghi; c += 64u) start[c] = gix;
  }
  if (far) {
    const int p = atomicAdd(&s_cnt, 1);
    s_lo[p] = lo; s_hi[p] = hi; s_ix[p] = i;
  }
  __syncthreads();
  const int ng = s_cnt;
  for (int g = 0; g < ng; g++) {
    const unsigned ghi = s_hi[g], gix = s_ix[g];
    for (unsigned long long c = (unsigned long long)s_lo[g] + threadIdx.x; c <= ghi; c += 256ull) start[c] = gix;
  }
}
__global__ __launch_bounds__(256) void k_cell_start_tail(const unsigned* __restrict__ keys, unsigned n, unsigned ncell, unsigned* __restrict__ start) {
  const unsigned long long first = n ? (unsigned long long)keys[n - 1] + 1ull : 0ull;
  for (unsigned long long c = first + blockIdx.x * 256ull + threadIdx.x; c <= ncell; c += (unsigned long long)gridDim.x * 256ull) start[c] = n;
}

__global__ __launch_bounds__(256) void k_gather_sorted(const float* __restrict__ xyz, int stride, const unsigned* __restrict__ vals, long long n,
                                                       float4* __restrict__ pts) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const long long s = vals[i];
  pts[i] = make_float4(xyz[s * stride], xyz[s * stride + 1], xyz[s * stride + 2], __uint_as_float((unsigned)s));
}

static int bits_for(unsigned long long maxv) {
  int b = 1;
  while (b < 64 && (maxv >> b) != 0ull) b++;
  return b;
}

}  // namespace

void gh_cell_start_launch(hipStream_t s, const unsigned* keys, unsigned n, unsigned ncell, unsigned* start) {
  if (n > 0) hipLaunchKernelGGL(k_cell_start_fill, dim3(cdiv(n, 256)), dim3(256), 0, s, keys, n, start);
  hipLaunchKernelGGL(k_cell_start_tail, dim3((unsigned)std::min<long long>(2048, cdiv((long long)ncell + 1, 256))), dim3(256), 0, s, keys, n, ncell, start);
}

int gh_grid_build(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float cell, const GridSlots& sl, DeviceGrid* out) {
  hipStream_t s = ctx->stream;
  float mm[6] = {0, 0, 0, 0, 0, 0};
  if (n > 0) GH_TRY(gh_bbox_dev(ctx, xyz, n, stride, mm));
  const GridDesc g = gh_grid_desc(mm, n, cell);
  unsigned *keys, *keys2, *vals, *vals2, *start;
  float4* pts;
  GH_TRY(ctx->reserve(sl.keys, (size_t)n + 1, &keys));
  GH_TRY(ctx->reserve(sl.keys2, (size_t)n + 1, &keys2));
  GH_TRY(ctx->reserve(sl.vals, (size_t)n + 1, &vals));
  GH_TRY(ctx->reserve(sl.vals2, (size_t)n + 1, &vals2));
  GH_TRY(ctx->reserve(sl.start, (size_t)g.ncell + 2, &start));
  GH_TRY(ctx->reserve(sl.pts, (size_t)n + 1, &pts));
  if (n > 0) {
    hipLaunchKernelGGL(k_cell_keys, dim3(cdiv(n, 256)), dim3(256), 0, s, xyz, n, stride, g, keys, vals);
    GH_TRY(gh_radix_sort_u32(ctx, keys, keys2, vals, vals2, n, 0, bits_for(g.ncell)));  // stable: a cell keeps its points in input order (prims.hip)
    hipLaunchKernelGGL(k_gather_sorted, dim3(cdiv(n, 256)), dim3(256), 0, s, xyz, stride, vals2, n, pts);
  }
  gh_cell_start_launch(s, keys2, (unsigned)n, g.ncell, start);
  GH_HIP(hipGetLastError());
  out->d = g;
  out->pts = pts;
  out->start = start;
  out->keys = keys2;
  return GHICP_OK;
}

// =============================================================================== voxel filter
namespace {

struct VoxDesc {
  float mn[3];
  float inv;
  unsigned long long mul_x, mul_y;
};

// filter.hpp:57-70: float (p - min) * inv, floor, cast to u64; key = vx*mul_vx + vy*mul_vy + vz
__global__ __launch_bounds__(256) void k_voxel_keys(const float* __restrict__ xyz, long long n, int stride, VoxDesc v,
                                                    unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const unsigned long long vx = (unsigned long long)floorf((xyz[i * stride] - v.mn[0]) * v.inv);
  const unsigned long long vy = (unsigned long long)floorf((xyz[i * stride + 1] - v.mn[1]) * v.inv);
  const unsigned long long vz = (unsigned long long)floorf((xyz[i * stride + 2] - v.mn[2]) * v.inv);
  keys[i] = vx * v.mul_x + vy * v.mul_y + vz;
  vals[i] = (unsigned)i;
}

// head of every voxel run whose key > 0; slot 0 of the output is the phantom group (filter.hpp:52,66,75-83)
__global__ __launch_bounds__(256) void k_voxel_flags(const unsigned long long* __restrict__ keys, long long n, unsigned char* __restrict__ flags) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  flags[i] = (k != 0ull && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

__global__ void k_set_first(int* keep) { keep[0] = 0; }

}  // namespace

int gh_voxel_filter_dev(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float voxel, int32_t* keep, long long* m_out) {
  hipStream_t s = ctx->stream;
  if (n <= 0) { *m_out = 0; return GHICP_OK; }
  float mm[6];
  GH_TRY(gh_bbox_dev(ctx, xyz, n, stride, mm));
  VoxDesc v;
  v.inv = 1.0f / voxel;  // filter.hpp:30
  unsigned long long maxv[3];
  for (int d = 0; d < 3; d++) {
    v.mn[d] = mm[d];
    const float gap = mm[3 + d] - mm[d];                              // Eigen::Vector4f gap_p = max_p - min_p
    maxv[d] = (unsigned long long)(std::ceil(gap * v.inv) + 1);       // filter.hpp:38-40
  }
  v.mul_x = maxv[1] * maxv[2];
  v.mul_y = maxv[2];
  const long double total = (long double)maxv[0] * (long double)maxv[1] * (long double)maxv[2];
  if (total >= 18446744073709551615.0L) return ctx->fail(GHICP_ERR_CAPACITY, "voxel filter: the number of boxes exceeds the limit");  // filter.hpp:42-46
  unsigned long long *keys, *keys2;
  unsigned *vals, *vals2;
  unsigned char* flags;
  int* dcount;
  GH_TRY(ctx->reserve(B_GRID_KEYS, (size_t)n * 2 + 2, (unsigned**)&keys));
  GH_TRY(ctx->reserve(B_GRID_KEYS2, (size_t)n * 2 + 2, (unsigned**)&keys2));
  GH_TRY(ctx->reserve(B_GRID_VALS, (size_t)n + 1, &vals));
  GH_TRY(ctx->reserve(B_GRID_VALS2, (size_t)n + 1, &vals2));
  GH_TRY(ctx->reserve(B_FE_FLAGS, (size_t)n + 16, &flags));
  GH_TRY(ctx->reserve(B_FE_SCAN, 16, &dcount));
  hipLaunchKernelGGL(k_voxel_keys, dim3(cdiv(n, 256)), dim3(256), 0, s, xyz, n, stride, v, keys, vals);
  const unsigned long long maxkey = (maxv[0] - 1) * v.mul_x + (maxv[1] - 1) * v.mul_y + (maxv[2] - 1);
  const int eb = bits_for(maxkey);
  hipEvent_t kev = ctx->kt_begin(KT_VOXEL_SORT);
  GH_TRY(gh_radix_sort_u64(ctx, keys, keys2, vals, vals2, n, 0, eb));  // stable: lowest input index leads its voxel (prims.hip)
  ctx->kt_end(KT_VOXEL_SORT, kev);
  hipLaunchKernelGGL(k_voxel_flags, dim3(cdiv(n, 256)), dim3(256), 0, s, keys2, n, flags);
  hipLaunchKernelGGL(k_set_first, dim3(1), dim3(1), 0, s, keep);
  GH_TRY(gh_select_flagged_u32(ctx, vals2, flags, n, reinterpret_cast<unsigned*>(keep + 1), dcount));  // the run heads' point indices, in voxel order (prims.hip)
  int* hc = reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->pinned) + 320);  // pinned: see gh_bbox_dev
  GH_HIP(hipMemcpyAsync(hc, dcount, sizeof(int), hipMemcpyDeviceToHost, s));
  GH_HIP(hipStreamSynchronize(s));
  *m_out = (long long)*hc + 1;
  return GHICP_OK;
}

extern "C" int ghicp_voxel_filter(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, float voxel, int32_t* keep_idx, int64_t* m) {
  GH_ENTER(ctx);
  GH_ARG(n >= 0 && n < (1ll << 31) - 2 && stride >= 3 && voxel > 0.f && m != nullptr);
  Stager sg(ctx);
  const float* d;
  int32_t* k;
  GH_TRY(sg.in_cloud(xyz, (size_t)n * stride, &d));
  GH_TRY(sg.out(keep_idx, (size_t)n + 1, &k));
  long long mm = 0;
  GH_TRY(gh_voxel_filter_dev(ctx, d, n, stride, voxel, k, &mm));
  *m = mm;
  return sg.finish();
}
