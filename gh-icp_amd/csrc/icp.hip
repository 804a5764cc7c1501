// Fine registration on gfx950: CRegistration<PointT>::icp_reg / ptplicp_reg / calOverlap / transformcloud / invTransform
// (reference include/common_reg.h:26-110, src/common_reg.cpp:45-107, 122-199, 294-317, 325-370).  The reference hands the
// loop to PCL (IterativeClosestPoint[WithNormals], CorrespondenceRejectorTrimmed, KdTreeFLANN); here it is
//   k_nn_fine / k_nn_coarse  exact 1-NN of every (transformed) source point in the target: thread-per-query ring search
//                            on a fine uniform grid, the unresolved tail (queries far from the target) handed to a
//                            wave-per-query search on an 8x coarser grid.  float L2, ties -> lower index.
//   k_sel_pass (radix select) the trimmed rejector (keep the floor(overlap * count) smallest by (d^2, source index))
//   k_acc_means / k_acc_cov  float Umeyama sums in f64 (N2), per-block partials reduced in a fixed order
//   k_acc_plane              point-to-plane LLS normal equations (6x6, f64 sums of float terms)
//   k_icp_step               the closed-form solve + pcl DefaultConvergenceCriteria, one thread
//   k_apply                  transformation_ applied to the working copy of the source
// One 64-byte status record per iteration is the only device->host traffic.  HBM-bound: per iteration the compulsory
// traffic is 16 B in + 16 B out per source point plus the target cells each query touches.
#include "grid.h"
#include "devmath.h"


#include <cmath>
#include <cstdlib>

int gh_knn_normals_dev(ghicp_ctx* ctx, const float* xyz, long long m, int stride, int k, float* normals);

namespace {

constexpr float kInf = 3.0e38f;
constexpr int RCAP = 2;     // rings searched per query on the fine grid before it is handed to the coarse grid
constexpr int NBLK = 512;   // partial-sum blocks of the accumulation kernels
constexpr int NPART = 32;   // doubles per partial record

struct NnGrid {
  GridDesc d;
  const float4* pts;
  const unsigned* start;
  float cell;
};

struct NnIndex {
  NnGrid fine, coarse;
};

struct IcpState {
  float T[16];    // transformation_ of this iteration
  float fin[16];  // final_transformation_
  double prev_mse, mse, eps_t, eps_e;
  float msf[3], mtf[3];
  int iterations, max_iter, converged, reason;
  unsigned count, nv;  // valid correspondences / kept after trimming
  int trimmed, metric;
  float ratio;
  unsigned pend;
  // radix select of the trimming threshold K* = (d2star, istar): correspondences with a smaller (d2 bits, index) are kept
  unsigned sel_prefix[6], sel_rank[6];
  unsigned d2star, istar;
  int sel_done;
};

// ------------------------------------------------------------------------------------------------ 1-NN search
// Lower bound on the distance from p to any point in a cell outside the block [c - r, c + r]^3 (sides clipped by the
// grid need no bound: nothing lies beyond them).  Cell assignment is a rounded float product, hence the margin.
__device__ inline float block_reach(const NnGrid& G, float px, float py, float pz, int cx, int cy, int cz, int r) {
  float m = kInf;
  const float p[3] = {px, py, pz};
  const int c[3] = {cx, cy, cz};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (c[a] - r > 0) m = fminf(m, p[a] - (G.d.mn[a] + (float)(c[a] - r) * G.cell));
    if (c[a] + r < G.d.dim[a] - 1) m = fminf(m, (G.d.mn[a] + (float)(c[a] + r + 1) * G.cell) - p[a]);
  }
  if (m >= kInf) return kInf;
  return m - 2e-3f * G.cell;
}

// squared distance from coordinate p to the cell interval [lo, hi] of one axis (cells as the grid assigns them, shrunk
// by the same rounding margin as block_reach)
__device__ inline float axis_gap2(const NnGrid& G, int a, float p, int lo, int hi) {
  const float m = 2e-3f * G.cell;
  const float l = G.d.mn[a] + (float)lo * G.cell + m, h = G.d.mn[a] + (float)(hi + 1) * G.cell - m;
  const float d = fmaxf(fmaxf(l - p, p - h), 0.f);
  return d * d;
}

// The cells of block r that are not in block rlo (rlo = -1: the whole block), as z-contiguous runs of the point array.
// bound(x) is called before every x slab and returns the squared distance beyond which a run cannot matter; runs whose
// box lies farther than that from P are skipped (a point at exactly the bound still ties, hence the strict test).
template <typename B, typename F>
__device__ inline void for_shell_runs(const NnGrid& G, float px, float py, float pz, int cx, int cy, int cz, int rlo, int r, B&& bound, F&& f) {
  const GridDesc& g = G.d;
  const unsigned* __restrict__ start = G.start;
  const int x0 = max(cx - r, 0), x1 = min(cx + r, g.dim[0] - 1);
  const int y0 = max(cy - r, 0), y1 = min(cy + r, g.dim[1] - 1);
  const int zl = max(cz - r, 0), zh = min(cz + r, g.dim[2] - 1);
  for (int x = x0; x <= x1; x++) {
    const float lim = bound(x);
    const float gx = axis_gap2(G, 0, px, x, x);
    if (gx > lim) continue;
    for (int y = y0; y <= y1; y++) {
      const float gxy = gx + axis_gap2(G, 1, py, y, y);
      if (gxy > lim) continue;
      const unsigned base = ((unsigned)x * g.dim[1] + y) * g.dim[2];
      if (max(abs(x - cx), abs(y - cy)) > rlo) {
        if (gxy + axis_gap2(G, 2, pz, zl, zh) > lim) continue;
        const unsigned b = start[base + zl], e = start[base + zh + 1];
        if (e > b) f(b, e);
      } else {
        const int a1 = min(cz - rlo - 1, g.dim[2] - 1);
        if (zl <= a1 && !(gxy + axis_gap2(G, 2, pz, zl, a1) > lim)) {
          const unsigned b = start[base + zl], e = start[base + a1 + 1];
          if (e > b) f(b, e);
        }
        const int b0 = max(cz + rlo + 1, 0);
        if (b0 <= zh && !(gxy + axis_gap2(G, 2, pz, b0, zh) > lim)) {
          const unsigned b = start[base + b0], e = start[base + zh + 1];
          if (e > b) f(b, e);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_nn_fine(NnGrid G, const float4* __restrict__ q, int nq, int* __restrict__ nn, float* __restrict__ nd,
                                                 unsigned* __restrict__ pend_list, unsigned* __restrict__ pend_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  const float4 P = q[i];
  const int cx = gh_cell_coord(P.x, G.d.mn[0], G.d.inv, G.d.dim[0]);
  const int cy = gh_cell_coord(P.y, G.d.mn[1], G.d.inv, G.d.dim[1]);
  const int cz = gh_cell_coord(P.z, G.d.mn[2], G.d.inv, G.d.dim[2]);
  float bd = kInf;
  int bi = -1;
  bool done = false;
  int rlo = -1;
  for (int r = 1; r <= RCAP; r++) {
    for_shell_runs(G, P.x, P.y, P.z, cx, cy, cz, rlo, r, [&](int) { return bd; }, [&](unsigned b, unsigned e) {
      for (unsigned t = b; t < e; t++) {
        const float4 Q = G.pts[t];
        const int qi = (int)__float_as_uint(Q.w);
        const float dx = P.x - Q.x, dy = P.y - Q.y, dz = P.z - Q.z;
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        if (d2 < bd || (d2 == bd && qi < bi)) { bd = d2; bi = qi; }
      }
    });
    rlo = r;
    const float reach = block_reach(G, P.x, P.y, P.z, cx, cy, cz, r);
    if (reach >= kInf || (reach > 0.f && bd < reach * reach)) { done = true; break; }
  }
  nn[i] = bi;
  nd[i] = bd;
  if (!done) pend_list[atomicAdd(pend_count, 1u)] = (unsigned)i;
}

__device__ inline unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;
    v = w < v ? w : v;
  }
  return v;
}

// One wave per unresolved query, seeded with the fine-grid candidate; ring expansion without a cap.
__global__ __launch_bounds__(256) void k_nn_coarse(NnGrid G, const float4* __restrict__ q, const unsigned* __restrict__ pend_list,
                                                   const unsigned* __restrict__ pend_count, int* __restrict__ nn, float* __restrict__ nd) {
  const int lane = threadIdx.x & 63;
  const unsigned nw = gridDim.x * 4u, np = *pend_count;
  for (unsigned w = blockIdx.x * 4u + (threadIdx.x >> 6); w < np; w += nw) {
    const unsigned i = pend_list[w];
    const float4 P = q[i];
    const int cx = gh_cell_coord(P.x, G.d.mn[0], G.d.inv, G.d.dim[0]);
    const int cy = gh_cell_coord(P.y, G.d.mn[1], G.d.inv, G.d.dim[1]);
    const int cz = gh_cell_coord(P.z, G.d.mn[2], G.d.inv, G.d.dim[2]);
    unsigned long long best = ((unsigned long long)__float_as_uint(nd[i]) << 32) | (unsigned)nn[i];
    for (int r = 0;; r++) {
      // Shell r as 2 slots per (x, y) column of the block: a rim column is one z run (slot 0), an interior column its two
      // cap cells.  The lanes look the slots up in parallel (box test against the wave's best, then the cell table), then
      // the wave walks the non-empty runs together: no serial chain of table lookups per column.
      const int side = 2 * r + 1, slots = 2 * side * side;
      for (int s0 = 0; s0 < slots; s0 += 64) {
        best = wave_min_u64(best);
        const float lim = __uint_as_float((unsigned)(best >> 32));
        unsigned rb = 0, re = 0;
        const int sl = s0 + lane;
        if (sl < slots) {
          const int c = sl >> 1, h = sl & 1;
          const int x = cx + c / side - r, y = cy + c % side - r;
          if (x >= 0 && x < G.d.dim[0] && y >= 0 && y < G.d.dim[1]) {
            const bool rim = max(abs(x - cx), abs(y - cy)) == r;
            int zl, zh;
            if (rim) { zl = cz - r; zh = h ? zl - 1 : cz + r; }
            else { zl = zh = h ? cz + r : cz - r; }
            if (r == 0 && h) zh = zl - 1;
            zl = max(zl, 0); zh = min(zh, G.d.dim[2] - 1);
            if (zl <= zh && !(axis_gap2(G, 0, P.x, x, x) + axis_gap2(G, 1, P.y, y, y) + axis_gap2(G, 2, P.z, zl, zh) > lim)) {
              const unsigned base = ((unsigned)x * G.d.dim[1] + y) * G.d.dim[2];
              rb = G.start[base + zl];
              re = G.start[base + zh + 1];
            }
          }
        }
        unsigned long long live = __ballot(re > rb);
        while (live) {
          const int l = __ffsll((long long)live) - 1;
          live &= live - 1;
          const unsigned b = __shfl(rb, l, 64), e = __shfl(re, l, 64);
          for (unsigned t = b + lane; t < e; t += 64) {
            const float4 Q = G.pts[t];
            const float dx = P.x - Q.x, dy = P.y - Q.y, dz = P.z - Q.z;
            float d2 = dx * dx;
            d2 += dy * dy;
            d2 += dz * dz;
            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(Q.w);
            best = key < best ? key : best;
          }
        }
      }
      best = wave_min_u64(best);
      const float bd = __uint_as_float((unsigned)(best >> 32));
      const float reach = block_reach(G, P.x, P.y, P.z, cx, cy, cz, r);
      if (reach >= kInf || (reach > 0.f && bd < reach * reach)) break;
    }
    if (lane == 0) {
      nn[i] = (int)(unsigned)best;
      nd[i] = __uint_as_float((unsigned)(best >> 32));
    }
  }
}

// adds the number of threads of a 256-thread block with `flag` set to *dst: one atomic per block
__device__ inline void block_count_add(bool flag, unsigned* dst) {
  __shared__ unsigned wsum[4];
  const unsigned long long b = __ballot(flag);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (t) atomicAdd(dst, t);
  }
}

__global__ __launch_bounds__(256) void k_count_occupied(const unsigned* __restrict__ keys, unsigned n, unsigned* __restrict__ out) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  const bool first = i < n && (i == 0 || keys[i] != keys[i - 1]);
  block_count_add(first, out);
}

__global__ __launch_bounds__(256) void k_pack4(const float* __restrict__ xyz, long long n, int stride, float4* __restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n) out[i] = make_float4(xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2], 0.f);
}

__global__ __launch_bounds__(256) void k_gather_query(const float4* __restrict__ tgt, const int* __restrict__ nn, int n, float4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = tgt[max(nn[i], 0)];
}

__global__ __launch_bounds__(256) void k_reciprocal(const int* __restrict__ back, int n, int* __restrict__ nn) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && back[i] != i) nn[i] = -1;
}

// ------------------------------------------------------------------------------------------------ overlap
__global__ __launch_bounds__(256) void k_overlap(NnGrid G, const float* __restrict__ xyz, long long n, int stride, float r2, unsigned* __restrict__ count) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  bool hit = false;
  if (i < n) {
    const float px = xyz[i * stride], py = xyz[i * stride + 1], pz = xyz[i * stride + 2];
    // a query outside the padded grid has no neighbour; inside, the 27 cells around it hold every candidate
    const float fx = (px - G.d.mn[0]) * G.d.inv, fy = (py - G.d.mn[1]) * G.d.inv, fz = (pz - G.d.mn[2]) * G.d.inv;
    if (fx >= -1.f && fy >= -1.f && fz >= -1.f && fx <= (float)G.d.dim[0] + 1.f && fy <= (float)G.d.dim[1] + 1.f && fz <= (float)G.d.dim[2] + 1.f) {
      const int cx = gh_cell_coord(px, G.d.mn[0], G.d.inv, G.d.dim[0]);
      const int cy = gh_cell_coord(py, G.d.mn[1], G.d.inv, G.d.dim[1]);
      const int cz = gh_cell_coord(pz, G.d.mn[2], G.d.inv, G.d.dim[2]);
      gh_for_runs(G.d, G.start, cx, cy, cz, [&](unsigned b, unsigned e) {
        for (unsigned t = b; t < e && !hit; t++) {
          const float4 Q = G.pts[t];
          const float dx = px - Q.x, dy = py - Q.y, dz = pz - Q.z;
          float d2 = dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          if (d2 < r2) hit = true;
        }
      });
    }
  }
  block_count_add(hit, count);
}

// ------------------------------------------------------------------------------------------------ correspondences
__global__ __launch_bounds__(256) void k_corr_count(const int* __restrict__ nn, int n, IcpState* __restrict__ st) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  block_count_add(i < n && nn[i] >= 0, &st->count);
}

// CorrespondenceRejectorTrimmed::getRemainingCorrespondences: floor(overlap_ratio * float(size))
__global__ void k_icp_prep(IcpState* st) {
  unsigned nv = st->count;
  if (st->trimmed) {
    const unsigned t = (unsigned)(int)floorf(st->ratio * (float)st->count);
    if (t < nv) nv = t;
  }
  st->nv = nv;
  st->sel_prefix[0] = 0;
  st->sel_rank[0] = nv;
  st->sel_done = nv >= st->count;  // nothing to trim: every valid correspondence is kept
  st->d2star = 0xffffffffu;
  st->istar = 0xffffffffu;
}

// ---- trimmed rejector without a sort.  The kept set is {key < K*} with key = (d2 bits, source index) and K* the key of
// rank nv: an MSD radix select, three digit passes (11 + 11 + 10 bits) over the distance bits, then -- only when ties at
// the threshold distance have to be split -- three over the index bits.  Every pass is one histogram kernel whose blocks
// first derive the prefix chosen so far from the previous pass's histogram.
constexpr int SEL_BINS = 2048;
__device__ inline int sel_bits(int p) { return (p % 3 == 2) ? 10 : 11; }
__device__ inline int sel_shift(int p) { return (p % 3 == 0) ? 21 : ((p % 3 == 1) ? 10 : 0); }

// bin of `hist` (SEL_BINS entries) that holds rank r; *below = entries in lower bins.  Block-cooperative, 256 threads.
__device__ inline unsigned sel_pick(const unsigned* __restrict__ hist, unsigned r, unsigned* below, int* scan_s, unsigned* pick_s) {
  unsigned c[8], sum = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { c[k] = hist[threadIdx.x * 8 + k]; sum += c[k]; }
  int tot;
  const unsigned ex = (unsigned)gh_block_excl_scan((int)sum, scan_s, &tot);
  if (r >= ex && r < ex + sum) {
    unsigned acc = ex;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (r >= acc && r < acc + c[k]) { pick_s[0] = threadIdx.x * 8 + k; pick_s[1] = acc; }
      acc += c[k];
    }
  }
  __syncthreads();
  *below = pick_s[1];
  return pick_s[0];
}

__global__ __launch_bounds__(256) void k_sel_pass(int pass, const int* __restrict__ nn, const float* __restrict__ nd, int n, IcpState* __restrict__ st,
                                                  unsigned* __restrict__ hist) {
  if (st->sel_done || (pass > 3 && st->istar == 0)) return;
  __shared__ unsigned lh[SEL_BINS];
  __shared__ int scan_s[20];
  __shared__ unsigned pick_s[2];
  unsigned prefix = 0, rank = st->sel_rank[0], d2star = st->d2star;
  if (pass > 0) {
    unsigned below;
    const unsigned bin = sel_pick(hist + (size_t)(pass - 1) * SEL_BINS, st->sel_rank[pass - 1], &below, scan_s, pick_s);
    prefix = (st->sel_prefix[pass - 1] << sel_bits(pass - 1)) | bin;
    rank = st->sel_rank[pass - 1] - below;
    if (pass == 3) {  // the distance is fixed: `rank` of its ties (lowest indices first) are kept
      d2star = prefix;
      prefix = 0;
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->d2star = d2star;
        if (rank == 0) st->istar = 0;
      }
      if (rank == 0) return;  // K* is the first tie: no index digits needed (k_sel_final sees rank 0 too)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->sel_prefix[pass] = prefix; st->sel_rank[pass] = rank; }
  }
  for (int k = threadIdx.x; k < SEL_BINS; k += 256) lh[k] = 0;
  __syncthreads();
  const int shift = sel_shift(pass), bits = sel_bits(pass);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if (nn[i] < 0) continue;
    const unsigned db = __float_as_uint(nd[i]);
    unsigned v;
    bool member;
    if (pass < 3) { v = db; member = pass == 0 || (v >> (shift + bits)) == prefix; }
    else { v = (unsigned)i; member = db == d2star && (pass == 3 || (v >> (shift + bits)) == prefix); }
    if (member) atomicAdd(&lh[(v >> shift) & ((1u << bits) - 1u)], 1u);
  }
  __syncthreads();
  unsigned* out = hist + (size_t)pass * SEL_BINS;
  for (int k = threadIdx.x; k < SEL_BINS; k += 256)
    if (lh[k]) atomicAdd(&out[k], lh[k]);
}

__global__ __launch_bounds__(256) void k_sel_final(IcpState* __restrict__ st, const unsigned* __restrict__ hist) {
  if (st->sel_done) return;
  __shared__ int scan_s[20];
  __shared__ unsigned pick_s[2];
  if (st->istar == 0) return;  // decided at pass 3
  unsigned below;
  const unsigned bin = sel_pick(hist + (size_t)5 * SEL_BINS, st->sel_rank[5], &below, scan_s, pick_s);
  if (threadIdx.x == 0) st->istar = (st->sel_prefix[5] << 10) | bin;
}

struct CorrView {
  const int* nn;
  const float* nd;
  const float4* cur;
  const float4* tgt;
  int ns;
};

// source point e -> its target j if the correspondence survives the rejectors, else -1
__device__ inline int corr_at(const CorrView& V, const IcpState* __restrict__ st, unsigned e, int* j) {
  *j = V.nn[e];
  if (*j < 0) return -1;
  const unsigned db = __float_as_uint(V.nd[e]);
  return (db < st->d2star || (db == st->d2star && e < st->istar)) ? (int)e : -1;
}

__device__ inline void store_partials(const double* v, int nv, double* red, double* __restrict__ part) {
  for (int d = 0; d < nv; d++) {
    const double s = gh_block_sum(v[d], red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * NPART + d] = s;
  }
}

__global__ __launch_bounds__(256) void k_acc_means(CorrView V, const IcpState* __restrict__ st, double* __restrict__ part) {
  __shared__ double red[16];
  const unsigned lim = (unsigned)V.ns;
  double m[7] = {0, 0, 0, 0, 0, 0, 0};
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < lim; e += NBLK * 256u) {
    int j;
    const int i = corr_at(V, st, e, &j);
    if (i < 0) continue;
    const float4 S = V.cur[i], D = V.tgt[j];
    m[0] += (double)S.x; m[1] += (double)S.y; m[2] += (double)S.z;
    m[3] += (double)D.x; m[4] += (double)D.y; m[5] += (double)D.z;
    m[6] += (double)V.nd[i];
  }
  store_partials(m, 7, red, part);
}

// sum of component d over the NBLK block partials by one wave: lane l adds blocks l, l+64, ... then a fixed shuffle tree
__device__ inline double wave_reduce_partials(const double* __restrict__ part, int d) {
  double s = 0;
  for (int b = threadIdx.x; b < NBLK; b += 64) s += part[(size_t)b * NPART + d];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

__global__ __launch_bounds__(64) void k_icp_means(IcpState* st, const double* __restrict__ part) {
  double m[7];
  for (int d = 0; d < 7; d++) m[d] = wave_reduce_partials(part, d);
  if (threadIdx.x != 0) return;
  const double c = (double)st->nv;
  for (int d = 0; d < 3; d++) { st->msf[d] = (float)(m[d] / c); st->mtf[d] = (float)(m[3 + d] / c); }
  st->mse = m[6] / c;  // DefaultConvergenceCriteria::calculateMSE over the remaining correspondences
}

__global__ __launch_bounds__(256) void k_acc_cov(CorrView V, const IcpState* __restrict__ st, double* __restrict__ part) {
  __shared__ double red[16];
  const unsigned lim = (unsigned)V.ns;
  const double ms[3] = {(double)st->msf[0], (double)st->msf[1], (double)st->msf[2]};
  const double mt[3] = {(double)st->mtf[0], (double)st->mtf[1], (double)st->mtf[2]};
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < lim; e += NBLK * 256u) {
    int j;
    const int i = corr_at(V, st, e, &j);
    if (i < 0) continue;
    const float4 S = V.cur[i], D = V.tgt[j];
    const double a[3] = {(double)D.x - mt[0], (double)D.y - mt[1], (double)D.z - mt[2]};
    const double b[3] = {(double)S.x - ms[0], (double)S.y - ms[1], (double)S.z - ms[2]};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int q = 0; q < 3; q++) H[r * 3 + q] += a[r] * b[q];
  }
  store_partials(H, 9, red, part);
}

// TransformationEstimationPointToPlaneLLS: rows [n x s ; n], rhs n.(d - s), float terms summed in f64
__global__ __launch_bounds__(256) void k_acc_plane(CorrView V, const float* __restrict__ tnrm, const IcpState* __restrict__ st, double* __restrict__ part) {
  __shared__ double red[16];
  const unsigned lim = (unsigned)V.ns;
  double acc[28];
#pragma unroll
  for (int d = 0; d < 28; d++) acc[d] = 0;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < lim; e += NBLK * 256u) {
    int j;
    const int i = corr_at(V, st, e, &j);
    if (i < 0) continue;
    const float4 S = V.cur[i], D = V.tgt[j];
    const float nx = tnrm[(size_t)j * 3], ny = tnrm[(size_t)j * 3 + 1], nz = tnrm[(size_t)j * 3 + 2];
    const double v[6] = {(double)(nz * S.y - ny * S.z), (double)(nx * S.z - nz * S.x), (double)(ny * S.x - nx * S.y), (double)nx, (double)ny, (double)nz};
    const double dd = (double)(((((nx * D.x + ny * D.y) + nz * D.z) - nx * S.x) - ny * S.y) - nz * S.z);
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; r++) {
#pragma unroll
      for (int q = r; q < 6; q++) acc[k++] += v[r] * v[q];
    }
#pragma unroll
    for (int r = 0; r < 6; r++) acc[21 + r] += v[r] * dd;
    acc[27] += (double)V.nd[i];
  }
  store_partials(acc, 28, red, part);
}

__device__ inline void mat4_mul(const float* a, const float* b, float* out) {
  float t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) t[r * 4 + c] = ((a[r * 4] * b[c] + a[r * 4 + 1] * b[4 + c]) + a[r * 4 + 2] * b[8 + c]) + a[r * 4 + 3] * b[12 + c];
  for (int d = 0; d < 16; d++) out[d] = t[d];
}

// closed-form solve + final_transformation_ update + DefaultConvergenceCriteria::hasConverged
__global__ __launch_bounds__(64) void k_icp_step(IcpState* st, const double* __restrict__ part) {
  const unsigned cnt = st->nv;
  double acc[28];
  const int nacc = st->metric == GHICP_ICP_POINT_TO_POINT ? 9 : 28;
  for (int d = 0; d < 28; d++) acc[d] = d < nacc ? wave_reduce_partials(part, d) : 0.0;
  if (threadIdx.x != 0) return;
  if (cnt < 3u) {  // min_number_correspondences_
    st->converged = 0; st->reason = GHICP_ICP_NO_CORRESPONDENCES; st->count = 0;
    return;
  }
  float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (st->metric == GHICP_ICP_POINT_TO_POINT) {
    double A[9], R[9];
    for (int d = 0; d < 9; d++) A[d] = acc[d] / (double)cnt;
    gh_quant_grid(A, 9);  // N2: umeyama's sigma is a Matrix3f
    gh_kabsch(A, R);
    float Rf[9];
    for (int d = 0; d < 9; d++) Rf[d] = (float)R[d];
    for (int r = 0; r < 3; r++) {
      for (int q = 0; q < 3; q++) T[r * 4 + q] = Rf[r * 3 + q];
      T[r * 4 + 3] = (float)((double)st->mtf[r] - (((double)Rf[r * 3] * (double)st->msf[0] + (double)Rf[r * 3 + 1] * (double)st->msf[1]) +
                                                   (double)Rf[r * 3 + 2] * (double)st->msf[2]));
    }
  } else {
    st->mse = acc[27] / (double)cnt;
    double A[6][6], bb[6], x[6];
    int k = 0;
    for (int r = 0; r < 6; r++)
      for (int q = r; q < 6; q++) { A[r][q] = acc[k]; A[q][r] = acc[k]; k++; }
    for (int r = 0; r < 6; r++) bb[r] = acc[21 + r];
    for (int c = 0; c < 6; c++) {  // elimination with partial pivoting (same sequence as the CPU restatement)
      int piv = c;
      for (int r = c + 1; r < 6; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
      if (piv != c) {
        for (int q = 0; q < 6; q++) { const double t = A[c][q]; A[c][q] = A[piv][q]; A[piv][q] = t; }
        const double t = bb[c]; bb[c] = bb[piv]; bb[piv] = t;
      }
      for (int r = c + 1; r < 6; r++) {
        const double f = A[r][c] / A[c][c];
        for (int q = c; q < 6; q++) A[r][q] -= f * A[c][q];
        bb[r] -= f * bb[c];
      }
    }
    for (int r = 5; r >= 0; r--) {
      double s = bb[r];
      for (int q = r + 1; q < 6; q++) s -= A[r][q] * x[q];
      x[r] = s / A[r][r];
    }
    const double al = x[0], be = x[1], ga = x[2];  // constructTransformationMatrix
    T[0] = (float)(cos(ga) * cos(be));
    T[1] = (float)(-sin(ga) * cos(al) + cos(ga) * sin(be) * sin(al));
    T[2] = (float)(sin(ga) * sin(al) + cos(ga) * sin(be) * cos(al));
    T[4] = (float)(sin(ga) * cos(be));
    T[5] = (float)(cos(ga) * cos(al) + sin(ga) * sin(be) * sin(al));
    T[6] = (float)(-cos(ga) * sin(al) + sin(ga) * sin(be) * cos(al));
    T[8] = (float)(-sin(be));
    T[9] = (float)(cos(be) * sin(al));
    T[10] = (float)(cos(be) * cos(al));
    T[3] = (float)x[3]; T[7] = (float)x[4]; T[11] = (float)x[5];
  }
  for (int d = 0; d < 16; d++) st->T[d] = T[d];
  mat4_mul(T, st->fin, st->fin);
  st->iterations++;
  st->count = 0;
  const double mse = st->mse, prev = st->prev_mse;
  if (st->iterations >= st->max_iter) { st->converged = 1; st->reason = GHICP_ICP_ITERATIONS; return; }
  const double cos_angle = 0.5 * ((double)T[0] + (double)T[5] + (double)T[10] - 1);
  const double tsq = (double)T[3] * T[3] + (double)T[7] * T[7] + (double)T[11] * T[11];
  if (cos_angle >= 1.0 - st->eps_t && tsq <= st->eps_t) { st->converged = 1; st->reason = GHICP_ICP_TRANSFORM; return; }
  if (fabs(mse - prev) < st->eps_e) { st->converged = 1; st->reason = GHICP_ICP_ABS_MSE; return; }
  if (fabs(mse - prev) / prev < 1e-5) { st->converged = 1; st->reason = GHICP_ICP_REL_MSE; return; }
  st->prev_mse = mse;
}

__global__ __launch_bounds__(256) void k_apply(float4* __restrict__ cur, int n, const IcpState* __restrict__ st) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (st->reason == GHICP_ICP_NO_CORRESPONDENCES) return;
  const float* M = st->T;
  const float4 P = cur[i];
  cur[i] = make_float4(((M[0] * P.x + M[1] * P.y) + M[2] * P.z) + M[3], ((M[4] * P.x + M[5] * P.y) + M[6] * P.z) + M[7],
                       ((M[8] * P.x + M[9] * P.y) + M[10] * P.z) + M[11], 0.f);
}

struct M16 { float m[16]; };
__global__ __launch_bounds__(256) void k_transform_f32(const float* __restrict__ xyz, long long n, int stride, M16 M, float* __restrict__ out3,
                                                       float4* __restrict__ out4) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[i * stride], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
  const float ox = ((M.m[0] * x + M.m[1] * y) + M.m[2] * z) + M.m[3];
  const float oy = ((M.m[4] * x + M.m[5] * y) + M.m[6] * z) + M.m[7];
  const float oz = ((M.m[8] * x + M.m[9] * y) + M.m[10] * z) + M.m[11];
  if (out3) { out3[i * 3] = ox; out3[i * 3 + 1] = oy; out3[i * 3 + 2] = oz; }
  if (out4) out4[i] = make_float4(ox, oy, oz, 0.f);
}

__global__ __launch_bounds__(256) void k_sum_f32(const float* __restrict__ v, int n, double* __restrict__ part) {
  __shared__ double red[16];
  double s = 0;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)n; e += NBLK * 256u) s += (double)v[e];
  s = gh_block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------ host side
const GridSlots kSlotsTF = {B_GRID_KEYS, B_GRID_KEYS2, B_GRID_VALS, B_GRID_VALS2, B_GRID_START, B_GRID_PTS};
const GridSlots kSlotsTC = {B_ICP_TC_KEYS, B_ICP_TC_KEYS2, B_ICP_TC_VALS, B_ICP_TC_VALS2, B_ICP_TC_START, B_ICP_TC_PTS};
const GridSlots kSlotsSF = {B_GRID2_KEYS, B_GRID2_KEYS2, B_GRID2_VALS, B_GRID2_VALS2, B_GRID2_START, B_GRID2_PTS};
const GridSlots kSlotsSC = {B_ICP_SC_KEYS, B_ICP_SC_KEYS2, B_ICP_SC_VALS, B_ICP_SC_VALS2, B_ICP_SC_START, B_ICP_SC_PTS};

NnGrid as_nn(const DeviceGrid& g) { return NnGrid{g.d, g.pts, g.start, 1.0f / g.d.inv}; }

// Builds the fine + coarse grids over xyz.  cell <= 0: pick the cell from the data -- start from a volume guess and halve
// it while the occupied cells hold more than ~8 points on average (surface scans fill a small share of their bounding box).
int build_index(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float cell, const GridSlots& sf, const GridSlots& sc, NnIndex* out,
                float* cell_used) {
  hipStream_t s = ctx->stream;
  DeviceGrid gf, gc;
  if (cell > 0.f) {
    GH_TRY(gh_grid_build(ctx, xyz, n, stride, cell, sf, &gf));
  } else {
    float mm[6];
    GH_TRY(gh_bbox_dev(ctx, xyz, n, stride, mm));
    const double vol = fmax(1e-9, (double)(mm[3] - mm[0] + 1e-3) * (mm[4] - mm[1] + 1e-3) * (mm[5] - mm[2] + 1e-3));
    cell = fmaxf((float)cbrt(vol / (double)n * 8.0), 0.02f);
    unsigned* cnt;
    GH_TRY(ctx->reserve(B_ICP_PEND, (size_t)n + 4, &cnt));
    for (int attempt = 0;; attempt++) {
      GH_TRY(gh_grid_build(ctx, xyz, n, stride, cell, sf, &gf));
      if (attempt >= 5) break;
      const double nc_next = (double)gf.d.dim[0] * gf.d.dim[1] * gf.d.dim[2] * 8.0;
      if (nc_next > (double)(1u << 26) || 1.0f / gf.d.inv > cell * 1.01f) break;  // next halving would not fit / was already coarsened
      GH_HIP(hipMemsetAsync(cnt, 0, 4, s));
      hipLaunchKernelGGL(k_count_occupied, dim3(cdiv(n, 256)), dim3(256), 0, s, gf.keys, (unsigned)n, cnt);
      unsigned occ = 0;
      GH_HIP(hipMemcpyAsync(&occ, cnt, 4, hipMemcpyDeviceToHost, s));
      GH_HIP(hipStreamSynchronize(s));
      if (occ == 0 || (double)n / occ <= 8.0) break;
      cell *= 0.5f;
    }
  }
  const float cf = 1.0f / gf.d.inv;
  GH_TRY(gh_grid_build(ctx, xyz, n, stride, cf * 8.0f, sc, &gc));
  out->fine = as_nn(gf);
  out->coarse = as_nn(gc);
  if (cell_used) *cell_used = cf;
  return GHICP_OK;
}

int nn_search(ghicp_ctx* ctx, const NnIndex& X, const float4* q, int nq, int* nn, float* nd) {
  if (nq <= 0) return GHICP_OK;
  hipStream_t s = ctx->stream;
  unsigned* pend;
  GH_TRY(ctx->reserve(B_ICP_PEND, (size_t)nq + 4, &pend));
  GH_HIP(hipMemsetAsync(pend, 0, 4, s));
  hipLaunchKernelGGL(k_nn_fine, dim3(cdiv(nq, 256)), dim3(256), 0, s, X.fine, q, nq, nn, nd, pend + 1, pend);
  hipLaunchKernelGGL(k_nn_coarse, dim3(min(cdiv(nq, 4), 4096)), dim3(256), 0, s, X.coarse, q, pend + 1, pend, nn, nd);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

int overlap_dev(ghicp_ctx* ctx, const float* d1, long long n1, int s1, const float* d2, long long n2, int s2, float thre_dis, float* ratio) {
  if (n1 <= 0) { *ratio = 0.f; return GHICP_OK; }
  unsigned cnt_h = 0;
  if (n2 > 0) {
    DeviceGrid g;
    GH_TRY(gh_grid_build(ctx, d2, n2, s2, thre_dis, kSlotsSF, &g));
    unsigned* cnt;
    GH_TRY(ctx->reserve(B_ICP_STATE, 64, &cnt));
    GH_HIP(hipMemsetAsync(cnt, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_overlap, dim3(cdiv(n1, 256)), dim3(256), 0, ctx->stream, as_nn(g), d1, n1, s1, thre_dis * thre_dis, cnt);
    GH_HIP(hipMemcpyAsync(&cnt_h, cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(hipStreamSynchronize(ctx->stream));
  }
  *ratio = (float)((0.01 + (int)cnt_h) / (double)n1);  // common_reg.cpp:313
  return GHICP_OK;
}

}  // namespace

extern "C" void ghicp_icp_params_default(ghicp_icp_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->max_iter = 50;
  p->metric = GHICP_ICP_POINT_TO_POINT;
  p->thre_dis = 0.5f;
  p->min_overlap = 0.1f;
  p->covariance_k = 15;
  p->transformation_epsilon = 1e-8;
  p->euclidean_fitness_epsilon = 1e-5;
}

extern "C" void ghicp_inv_transform(const float* T, float* inv) {
  float t[16];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[r * 4 + c] = T[c * 4 + r];
  t[3] = -T[3]; t[7] = -T[7]; t[11] = -T[11];
  t[12] = t[13] = t[14] = 0.f;
  t[15] = 1.f;
  memcpy(inv, t, sizeof(t));
}

extern "C" int ghicp_cal_overlap(ghicp_ctx* ctx, const float* xyz1, int64_t n1, int stride1, const float* xyz2, int64_t n2, int stride2,
                                 float thre_dis, float* ratio) {
  GH_ENTER(ctx);
  GH_ARG(n1 >= 0 && n2 >= 0 && n1 < (1ll << 31) - 2 && n2 < (1ll << 31) - 2 && stride1 >= 3 && stride2 >= 3 && thre_dis > 0.f && ratio != nullptr);
  Stager sg(ctx);
  const float *d1, *d2;
  GH_TRY(sg.in_cloud(xyz1, (size_t)n1 * stride1, &d1));
  GH_TRY(sg.in_cloud(xyz2, (size_t)n2 * stride2, &d2));
  return overlap_dev(ctx, d1, n1, stride1, d2, n2, stride2, thre_dis, ratio);
}

extern "C" int ghicp_transform_cloud_f32(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, const float* T16, float* out) {
  GH_ENTER(ctx);
  GH_ARG(n >= 0 && stride >= 3 && T16 != nullptr);
  Stager sg(ctx);
  const float* d;
  float* o;
  GH_TRY(sg.in_cloud(xyz, (size_t)n * stride, &d));
  GH_TRY(sg.out(out, (size_t)n * 3, &o));
  if (n > 0) {
    M16 M;
    memcpy(M.m, T16, sizeof(M.m));
    hipLaunchKernelGGL(k_transform_f32, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d, (long long)n, stride, M, o, (float4*)nullptr);
    GH_HIP(hipGetLastError());
  }
  return sg.finish();
}

extern "C" int ghicp_knn_normals(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, int k, float* normals) {
  GH_ENTER(ctx);
  GH_ARG(n >= 0 && n < (1ll << 31) - 2 && stride >= 3 && normals != nullptr);
  Stager sg(ctx);
  const float* d;
  float* o;
  GH_TRY(sg.in_cloud(xyz, (size_t)n * stride, &d));
  GH_TRY(sg.out(normals, (size_t)n * 3, &o));
  GH_TRY(gh_knn_normals_dev(ctx, d, n, stride, k, o));
  return sg.finish();
}

extern "C" int ghicp_nn_search(ghicp_ctx* ctx, const float* query, int64_t nq, int strideQ, const float* xyzT, int64_t nt, int strideT,
                               int32_t* idx, float* d2) {
  GH_ENTER(ctx);
  GH_ARG(nq >= 0 && nt > 0 && nq < (1ll << 31) - 2 && nt < (1ll << 31) - 2 && strideQ >= 3 && strideT >= 3);
  Stager sg(ctx);
  const float *dq, *dt;
  int32_t* di;
  float* dd;
  GH_TRY(sg.in(query, (size_t)nq * strideQ, &dq));
  GH_TRY(sg.in_cloud(xyzT, (size_t)nt * strideT, &dt));
  GH_TRY(sg.out(idx, (size_t)nq, &di));
  GH_TRY(sg.out(d2, (size_t)nq, &dd));
  if (nq > 0) {
    NnIndex X;
    GH_TRY(build_index(ctx, dt, nt, strideT, 0.f, kSlotsTF, kSlotsTC, &X, nullptr));
    float4* q4;
    GH_TRY(ctx->reserve(B_ICP_Q, (size_t)nq + 1, &q4));
    hipLaunchKernelGGL(k_pack4, dim3(cdiv(nq, 256)), dim3(256), 0, ctx->stream, dq, (long long)nq, strideQ, q4);
    GH_TRY(nn_search(ctx, X, q4, (int)nq, di, dd));
  }
  return sg.finish();
}

extern "C" int ghicp_icp(ghicp_ctx* ctx, const float* xyzS, int64_t ns, int strideS, const float* xyzT, int64_t nt, int strideT,
                         const ghicp_icp_params* P, float* T16, float* transformed, ghicp_icp_stats* stats) {
  GH_ENTER(ctx);
  GH_ARG(P != nullptr && T16 != nullptr && stats != nullptr && ns >= 0 && nt >= 0 && ns < (1ll << 31) - 2 && nt < (1ll << 31) - 2 && strideS >= 3 &&
         strideT >= 3);
  GH_ARG(P->metric == GHICP_ICP_POINT_TO_POINT || P->metric == GHICP_ICP_POINT_TO_PLANE);
  hipStream_t s = ctx->stream;
  memset(stats, 0, sizeof(*stats));
  Stager sg(ctx);
  const float *dS, *dT;
  float* dOut;
  GH_TRY(sg.in_cloud(xyzS, (size_t)ns * strideS, &dS));
  GH_TRY(sg.in_cloud(xyzT, (size_t)nt * strideT, &dT));
  GH_TRY(sg.out(transformed, (size_t)ns * 3, &dOut));

  float ratio = 1.0f;
  int trimmed = 0;
  if (P->use_trimmed) {  // common_reg.cpp:64-74
    GH_ARG(P->thre_dis > 0.f);
    GH_TRY(overlap_dev(ctx, dS, ns, strideS, dT, nt, strideT, P->thre_dis, &ratio));
    stats->overlap = ratio;
    if (ratio < P->min_overlap) { sg.outs.clear(); return GHICP_OK; }  // "This registration would not be done"
    trimmed = ratio < 1.0f;
  }
  stats->done = 1;
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  IcpState hst;
  memset(&hst, 0, sizeof(hst));
  memcpy(hst.fin, I16, sizeof(I16));
  memcpy(hst.T, I16, sizeof(I16));
  if (ns == 0 || nt == 0) {
    stats->reason = GHICP_ICP_NO_CORRESPONDENCES;
    memcpy(T16, I16, sizeof(I16));
    if (ns > 0 && dOut) {
      M16 M;
      memcpy(M.m, I16, sizeof(I16));
      hipLaunchKernelGGL(k_transform_f32, dim3(cdiv(ns, 256)), dim3(256), 0, s, dS, (long long)ns, strideS, M, dOut, (float4*)nullptr);
    }
    return sg.finish();
  }

  float4 *cur, *tgt4, *q4 = nullptr;
  int *nn, *nn2 = nullptr;
  float *nd, *nd2 = nullptr, *tnrm = nullptr;
  unsigned* hist = nullptr;
  double* part;
  IcpState* st;
  GH_TRY(ctx->reserve(B_ICP_CUR, (size_t)ns + 1, &cur));
  GH_TRY(ctx->reserve(B_ICP_OUT, (size_t)nt + 1, &tgt4));
  GH_TRY(ctx->reserve(B_ICP_NN, (size_t)ns + 1, &nn));
  GH_TRY(ctx->reserve(B_ICP_ND, (size_t)ns + 1, &nd));
  GH_TRY(ctx->reserve(B_ICP_PART, (size_t)NBLK * NPART, &part));
  GH_TRY(ctx->reserve(B_ICP_STATE, 1, &st));
  if (P->use_reciprocal) {
    GH_TRY(ctx->reserve(B_ICP_Q, (size_t)ns + 1, &q4));
    GH_TRY(ctx->reserve(B_ICP_NN2, (size_t)ns + 1, &nn2));
    GH_TRY(ctx->reserve(B_ICP_ND2, (size_t)ns + 1, &nd2));
  }
  if (trimmed) GH_TRY(ctx->reserve(B_ICP_KEYS, (size_t)6 * SEL_BINS, &hist));
  if (P->metric == GHICP_ICP_POINT_TO_PLANE) {  // common_reg.cpp:146-147 (only the target normals enter the LLS solve)
    GH_TRY(ctx->reserve(B_ICP_TNRM, (size_t)nt * 3 + 3, &tnrm));
    GH_TRY(gh_knn_normals_dev(ctx, dT, nt, strideT, P->covariance_k, tnrm));
  }
  hipLaunchKernelGGL(k_pack4, dim3(cdiv(ns, 256)), dim3(256), 0, s, dS, (long long)ns, strideS, cur);
  hipLaunchKernelGGL(k_pack4, dim3(cdiv(nt, 256)), dim3(256), 0, s, dT, (long long)nt, strideT, tgt4);
  NnIndex XT, XS;
  float cell = 0.f;
  GH_TRY(build_index(ctx, dT, nt, strideT, 0.f, kSlotsTF, kSlotsTC, &XT, &cell));

  hst.prev_mse = 1.7976931348623157e308;
  hst.eps_t = P->transformation_epsilon;
  hst.eps_e = P->euclidean_fitness_epsilon;
  hst.max_iter = P->max_iter;
  hst.trimmed = trimmed;
  hst.metric = P->metric;
  hst.ratio = ratio;
  GH_HIP(hipMemcpyAsync(st, &hst, sizeof(hst), hipMemcpyHostToDevice, s));
  IcpState* pin = reinterpret_cast<IcpState*>(ctx->pinned);
  static_assert(sizeof(IcpState) <= 4096, "status record must fit the pinned scratch");
  const int gS = cdiv(ns, 256);
  const CorrView V = {nn, nd, cur, tgt4, (int)ns};
  const int gSel = min(cdiv(ns, 2048), 512);
  for (;;) {
    GH_TRY(nn_search(ctx, XT, cur, (int)ns, nn, nd));
    if (P->use_reciprocal) {  // determineReciprocalCorrespondences
      GH_TRY(build_index(ctx, reinterpret_cast<const float*>(cur), ns, 4, cell, kSlotsSF, kSlotsSC, &XS, nullptr));
      hipLaunchKernelGGL(k_gather_query, dim3(gS), dim3(256), 0, s, tgt4, nn, (int)ns, q4);
      GH_TRY(nn_search(ctx, XS, q4, (int)ns, nn2, nd2));
      hipLaunchKernelGGL(k_reciprocal, dim3(gS), dim3(256), 0, s, nn2, (int)ns, nn);
    }
    hipLaunchKernelGGL(k_corr_count, dim3(gS), dim3(256), 0, s, nn, (int)ns, st);
    hipLaunchKernelGGL(k_icp_prep, dim3(1), dim3(1), 0, s, st);
    if (trimmed) {
      GH_HIP(hipMemsetAsync(hist, 0, (size_t)6 * SEL_BINS * sizeof(unsigned), s));
      for (int pass = 0; pass < 6; pass++) hipLaunchKernelGGL(k_sel_pass, dim3(gSel), dim3(256), 0, s, pass, nn, nd, (int)ns, st, hist);
      hipLaunchKernelGGL(k_sel_final, dim3(1), dim3(256), 0, s, st, hist);
    }
    if (P->metric == GHICP_ICP_POINT_TO_POINT) {
      hipLaunchKernelGGL(k_acc_means, dim3(NBLK), dim3(256), 0, s, V, st, part);
      hipLaunchKernelGGL(k_icp_means, dim3(1), dim3(64), 0, s, st, part);
      hipLaunchKernelGGL(k_acc_cov, dim3(NBLK), dim3(256), 0, s, V, st, part);
    } else {
      hipLaunchKernelGGL(k_acc_plane, dim3(NBLK), dim3(256), 0, s, V, tnrm, st, part);
    }
    hipLaunchKernelGGL(k_icp_step, dim3(1), dim3(64), 0, s, st, part);
    hipLaunchKernelGGL(k_apply, dim3(gS), dim3(256), 0, s, cur, (int)ns, st);
    GH_HIP(hipMemcpyAsync(pin, st, sizeof(IcpState), hipMemcpyDeviceToHost, s));
    GH_HIP(hipStreamSynchronize(s));
    if (pin->converged || pin->reason == GHICP_ICP_NO_CORRESPONDENCES) break;
  }
  hst = *pin;
  stats->iterations = hst.iterations;
  stats->converged = hst.converged;
  stats->reason = hst.reason;
  stats->correspondences = hst.nv;
  stats->mse = hst.mse;
  memcpy(T16, hst.fin, sizeof(hst.fin));
  // output = final_transformation_ * input, then getFitnessScore() on it
  M16 M;
  memcpy(M.m, hst.fin, sizeof(M.m));
  hipLaunchKernelGGL(k_transform_f32, dim3(gS), dim3(256), 0, s, dS, (long long)ns, strideS, M, dOut, cur);
  GH_TRY(nn_search(ctx, XT, cur, (int)ns, nn, nd));
  hipLaunchKernelGGL(k_sum_f32, dim3(NBLK), dim3(256), 0, s, nd, (int)ns, part);
  double hp[NBLK];
  GH_HIP(hipMemcpyAsync(hp, part, sizeof(hp), hipMemcpyDeviceToHost, s));
  GH_HIP(hipStreamSynchronize(s));
  double f = 0;
  for (int b = 0; b < NBLK; b++) f += hp[b];
  stats->fitness = f / (double)ns;
  return sg.finish();
}
