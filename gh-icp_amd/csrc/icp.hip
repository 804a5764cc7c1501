// Fine registration on gfx950: CRegistration<PointT>::icp_reg / ptplicp_reg / calOverlap / transformcloud / invTransform
// (reference include/common_reg.h:26-110, src/common_reg.cpp:45-107, 122-199, 294-317, 325-370).  The reference hands the
// loop to PCL (IterativeClosestPoint[WithNormals], CorrespondenceRejectorTrimmed, KdTreeFLANN); here it is
//   k_nn_fine / k_nn_coarse  exact 1-NN of every (transformed) source point in the target: thread-per-query ring search
//                            on a fine uniform grid, the unresolved tail (queries far from the target) handed to a
//                            wave-per-query search on an 8x coarser grid.  float L2, ties -> lower index.
//   k_corr_keys + radix sort the trimmed rejector (keep the floor(overlap * count) smallest by (d^2, source index))
//   k_acc_means / k_acc_cov  float Umeyama sums in f64 (N2), per-block partials reduced in a fixed order
//   k_acc_plane              point-to-plane LLS normal equations (6x6, f64 sums of float terms)
//   k_icp_step               the closed-form solve + pcl DefaultConvergenceCriteria, one thread
//   k_apply                  transformation_ applied to the working copy of the source
// One 64-byte status record per iteration is the only device->host traffic.  HBM-bound: per iteration the compulsory
// traffic is 16 B in + 16 B out per source point plus the target cells each query touches.
#include "grid.h"
#include "devmath.h"

#include <hipcub/hipcub.hpp>

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

// The cells of block r that are not in block rlo (rlo = -1: the whole block), as z-contiguous runs of the point array.
template <typename F>
__device__ inline void for_shell_runs(const GridDesc& g, const unsigned* __restrict__ start, int cx, int cy, int cz, int rlo, int r, F&& f) {
  const int x0 = max(cx - r, 0), x1 = min(cx + r, g.dim[0] - 1);
  const int y0 = max(cy - r, 0), y1 = min(cy + r, g.dim[1] - 1);
  const int zl = max(cz - r, 0), zh = min(cz + r, g.dim[2] - 1);
  for (int x = x0; x <= x1; x++)
    for (int y = y0; y <= y1; y++) {
      const unsigned base = ((unsigned)x * g.dim[1] + y) * g.dim[2];
      if (max(abs(x - cx), abs(y - cy)) > rlo) {
        const unsigned b = start[base + zl], e = start[base + zh + 1];
        if (e > b) f(b, e);
      } else {
        const int a1 = min(cz - rlo - 1, g.dim[2] - 1);
        if (zl <= a1) {
          const unsigned b = start[base + zl], e = start[base + a1 + 1];
          if (e > b) f(b, e);
        }
        const int b0 = max(cz + rlo + 1, 0);
        if (b0 <= zh) {
          const unsigned b = start[base + b0], e = start[base + zh + 1];
          if (e > b) f(b, e);
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
    for_shell_runs(G.d, G.start, cx, cy, cz, rlo, r, [&](unsigned b, unsigned e) {
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
    int rlo = -1;
    for (int r = 0;; r++) {
      for_shell_runs(G.d, G.start, cx, cy, cz, rlo, r, [&](unsigned b, unsigned e) {
        for (unsigned t = b + lane; t < e; t += 64) {
          const float4 Q = G.pts[t];
          const float dx = P.x - Q.x, dy = P.y - Q.y, dz = P.z - Q.z;
          float d2 = dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(Q.w);
          best = key < best ? key : best;
        }
      });
      rlo = r;
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

__global__ __launch_bounds__(256) void k_count_occupied(const unsigned* __restrict__ keys, unsigned n, unsigned* __restrict__ out) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  const bool first = i < n && (i == 0 || keys[i] != keys[i - 1]);
  const unsigned long long b = __ballot(first);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned)__popcll(b));
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
  const unsigned long long b = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned)__popcll(b));
}

// ------------------------------------------------------------------------------------------------ correspondences
__global__ __launch_bounds__(256) void k_corr_keys(const int* __restrict__ nn, const float* __restrict__ nd, int n, unsigned long long* __restrict__ keys,
                                                   IcpState* __restrict__ st) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < n && nn[i] >= 0;
  if (i < n && keys) keys[i] = valid ? (((unsigned long long)__float_as_uint(nd[i]) << 32) | (unsigned)i) : ~0ull;
  const unsigned long long b = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&st->count, (unsigned)__popcll(b));
}

// CorrespondenceRejectorTrimmed::getRemainingCorrespondences: floor(overlap_ratio * float(size))
__global__ void k_icp_prep(IcpState* st) {
  unsigned nv = st->count;
  if (st->trimmed) {
    const unsigned t = (unsigned)(int)floorf(st->ratio * (float)st->count);
    if (t < nv) nv = t;
  }
  st->nv = nv;
}

struct CorrView {
  const unsigned long long* sorted;  // trimmed: keys in ascending (d2, i) order; else NULL
  const int* nn;
  const float* nd;
  const float4* cur;
  const float4* tgt;
  int ns;
};

// entry e of the correspondence list -> (i, j) or i = -1
__device__ inline int corr_at(const CorrView& V, unsigned e, int* j) {
  const int i = V.sorted ? (int)(unsigned)V.sorted[e] : (int)e;
  *j = V.nn[i];
  return *j >= 0 ? i : -1;
}

__device__ inline void store_partials(const double* v, int nv, double* red, double* __restrict__ part) {
  for (int d = 0; d < nv; d++) {
    const double s = gh_block_sum(v[d], red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * NPART + d] = s;
  }
}

__global__ __launch_bounds__(256) void k_acc_means(CorrView V, const IcpState* __restrict__ st, double* __restrict__ part) {
  __shared__ double red[16];
  const unsigned lim = V.sorted ? st->nv : (unsigned)V.ns;
  double m[7] = {0, 0, 0, 0, 0, 0, 0};
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < lim; e += NBLK * 256u) {
    int j;
    const int i = corr_at(V, e, &j);
    if (i < 0) continue;
    const float4 S = V.cur[i], D = V.tgt[j];
    m[0] += (double)S.x; m[1] += (double)S.y; m[2] += (double)S.z;
    m[3] += (double)D.x; m[4] += (double)D.y; m[5] += (double)D.z;
    m[6] += (double)V.nd[i];
  }
  store_partials(m, 7, red, part);
}

__global__ void k_icp_means(IcpState* st, const double* __restrict__ part) {
  double m[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < NBLK; b++)
    for (int d = 0; d < 7; d++) m[d] += part[(size_t)b * NPART + d];
  const double c = (double)st->nv;
  for (int d = 0; d < 3; d++) { st->msf[d] = (float)(m[d] / c); st->mtf[d] = (float)(m[3 + d] / c); }
  st->mse = m[6] / c;  // DefaultConvergenceCriteria::calculateMSE over the remaining correspondences
}

__global__ __launch_bounds__(256) void k_acc_cov(CorrView V, const IcpState* __restrict__ st, double* __restrict__ part) {
  __shared__ double red[16];
  const unsigned lim = V.sorted ? st->nv : (unsigned)V.ns;
  const double ms[3] = {(double)st->msf[0], (double)st->msf[1], (double)st->msf[2]};
  const double mt[3] = {(double)st->mtf[0], (double)st->mtf[1], (double)st->mtf[2]};
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < lim; e += NBLK * 256u) {
    int j;
    const int i = corr_at(V, e, &j);
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
  const unsigned lim = V.sorted ? st->nv : (unsigned)V.ns;
  double acc[28];
#pragma unroll
  for (int d = 0; d < 28; d++) acc[d] = 0;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < lim; e += NBLK * 256u) {
    int j;
    const int i = corr_at(V, e, &j);
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
__global__ void k_icp_step(IcpState* st, const double* __restrict__ part) {
  const unsigned cnt = st->nv;
  if (cnt < 3u) {  // min_number_correspondences_
    st->converged = 0; st->reason = GHICP_ICP_NO_CORRESPONDENCES; st->count = 0;
    return;
  }
  float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (st->metric == GHICP_ICP_POINT_TO_POINT) {
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, R[9];
    for (int b = 0; b < NBLK; b++)
      for (int d = 0; d < 9; d++) A[d] += part[(size_t)b * NPART + d];
    for (int d = 0; d < 9; d++) A[d] /= (double)cnt;
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
    double acc[28];
    for (int d = 0; d < 28; d++) acc[d] = 0;
    for (int b = 0; b < NBLK; b++)
      for (int d = 0; d < 28; d++) acc[d] += part[(size_t)b * NPART + d];
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
    if (const char* e = getenv("GHICP_ICP_CELL")) cell = (float)atof(e);
    unsigned* cnt;
    GH_TRY(ctx->reserve(B_ICP_PEND, (size_t)n + 4, &cnt));
    for (int attempt = 0;; attempt++) {
      GH_TRY(gh_grid_build(ctx, xyz, n, stride, cell, sf, &gf));
      if (attempt >= 5 || getenv("GHICP_ICP_CELL")) break;
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
  if (!ctx) return GHICP_ERR_ARG;
  GH_ARG(n1 >= 0 && n2 >= 0 && n1 < (1ll << 31) - 2 && n2 < (1ll << 31) - 2 && stride1 >= 3 && stride2 >= 3 && thre_dis > 0.f && ratio != nullptr);
  Stager sg(ctx);
  const float *d1, *d2;
  GH_TRY(sg.in(xyz1, (size_t)n1 * stride1, &d1));
  GH_TRY(sg.in(xyz2, (size_t)n2 * stride2, &d2));
  return overlap_dev(ctx, d1, n1, stride1, d2, n2, stride2, thre_dis, ratio);
}

extern "C" int ghicp_transform_cloud_f32(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, const float* T16, float* out) {
  if (!ctx) return GHICP_ERR_ARG;
  GH_ARG(n >= 0 && stride >= 3 && T16 != nullptr);
  Stager sg(ctx);
  const float* d;
  float* o;
  GH_TRY(sg.in(xyz, (size_t)n * stride, &d));
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
  if (!ctx) return GHICP_ERR_ARG;
  GH_ARG(n >= 0 && n < (1ll << 31) - 2 && stride >= 3 && normals != nullptr);
  Stager sg(ctx);
  const float* d;
  float* o;
  GH_TRY(sg.in(xyz, (size_t)n * stride, &d));
  GH_TRY(sg.out(normals, (size_t)n * 3, &o));
  GH_TRY(gh_knn_normals_dev(ctx, d, n, stride, k, o));
  return sg.finish();
}

extern "C" int ghicp_nn_search(ghicp_ctx* ctx, const float* query, int64_t nq, int strideQ, const float* xyzT, int64_t nt, int strideT,
                               int32_t* idx, float* d2) {
  if (!ctx) return GHICP_ERR_ARG;
  GH_ARG(nq >= 0 && nt > 0 && nq < (1ll << 31) - 2 && nt < (1ll << 31) - 2 && strideQ >= 3 && strideT >= 3);
  Stager sg(ctx);
  const float *dq, *dt;
  int32_t* di;
  float* dd;
  GH_TRY(sg.in(query, (size_t)nq * strideQ, &dq));
  GH_TRY(sg.in(xyzT, (size_t)nt * strideT, &dt));
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
  if (!ctx) return GHICP_ERR_ARG;
  GH_ARG(P != nullptr && T16 != nullptr && stats != nullptr && ns >= 0 && nt >= 0 && ns < (1ll << 31) - 2 && nt < (1ll << 31) - 2 && strideS >= 3 &&
         strideT >= 3);
  GH_ARG(P->metric == GHICP_ICP_POINT_TO_POINT || P->metric == GHICP_ICP_POINT_TO_PLANE);
  hipStream_t s = ctx->stream;
  memset(stats, 0, sizeof(*stats));
  Stager sg(ctx);
  const float *dS, *dT;
  float* dOut;
  GH_TRY(sg.in(xyzS, (size_t)ns * strideS, &dS));
  GH_TRY(sg.in(xyzT, (size_t)nt * strideT, &dT));
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
  unsigned long long *keys = nullptr, *keys2 = nullptr;
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
  size_t sort_bytes = 0;
  char* sort_tmp = nullptr;
  if (trimmed) {
    GH_TRY(ctx->reserve(B_ICP_KEYS, (size_t)ns + 1, &keys));
    GH_TRY(ctx->reserve(B_ICP_KEYS2, (size_t)ns + 1, &keys2));
    GH_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys, keys2, (int)ns, 0, 64, s));
    GH_TRY(ctx->reserve(B_ICP_SORTTMP, sort_bytes + 16, &sort_tmp));
  }
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
  const CorrView V = {trimmed ? keys2 : nullptr, nn, nd, cur, tgt4, (int)ns};
  for (;;) {
    GH_TRY(nn_search(ctx, XT, cur, (int)ns, nn, nd));
    if (P->use_reciprocal) {  // determineReciprocalCorrespondences
      GH_TRY(build_index(ctx, reinterpret_cast<const float*>(cur), ns, 4, cell, kSlotsSF, kSlotsSC, &XS, nullptr));
      hipLaunchKernelGGL(k_gather_query, dim3(gS), dim3(256), 0, s, tgt4, nn, (int)ns, q4);
      GH_TRY(nn_search(ctx, XS, q4, (int)ns, nn2, nd2));
      hipLaunchKernelGGL(k_reciprocal, dim3(gS), dim3(256), 0, s, nn2, (int)ns, nn);
    }
    hipLaunchKernelGGL(k_corr_keys, dim3(gS), dim3(256), 0, s, nn, nd, (int)ns, keys, st);
    hipLaunchKernelGGL(k_icp_prep, dim3(1), dim3(1), 0, s, st);
    if (trimmed) GH_HIP(hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, keys, keys2, (int)ns, 0, 64, s));
    if (P->metric == GHICP_ICP_POINT_TO_POINT) {
      hipLaunchKernelGGL(k_acc_means, dim3(NBLK), dim3(256), 0, s, V, st, part);
      hipLaunchKernelGGL(k_icp_means, dim3(1), dim3(1), 0, s, st, part);
      hipLaunchKernelGGL(k_acc_cov, dim3(NBLK), dim3(256), 0, s, V, st, part);
    } else {
      hipLaunchKernelGGL(k_acc_plane, dim3(NBLK), dim3(256), 0, s, V, tnrm, st, part);
    }
    hipLaunchKernelGGL(k_icp_step, dim3(1), dim3(1), 0, s, st, part);
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
