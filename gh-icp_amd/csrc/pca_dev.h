// PCA of the query points of ONE occupied grid cell (pca.h:133-165, 202-250); shared by the single-cloud kernel (pca.hip) and the
// batched front end (batch.hip).  One wave; sC = PCA_CHUNK float4 of LDS.
#pragma once
#include "grid.h"
#include "devmath.h"

#include <cstdlib>

constexpr int PCA_CHUNK = 512;  // LDS tile of neighbour points (8 KB; a block stays resident in it up to 448 points: PCA_PAD)

// CHUNK only sets how many neighbour points are staged per barrier pair.
// Lane layout: a 0.5 m cell holds ~16 points, so one lane per query point would leave three quarters of the wave idle in the candidate
// loops.  The wave is therefore split into np query points x g lanes (g = the largest power of two with np * g <= 64): lane (qi, sl)
// tests the staged candidates t = sl, sl + g, ... against query qi, and the g partial sums of a query are combined by an xor butterfly
// (fixed order: deterministic, and the same in the single-cloud and the batched front end).  The sums are f64 and go through N2's
// rounding onto the f32 grid, so the summation order shows up as an ulp on a few points in 10^5, as before.
__device__ inline double gh_group_sum(double x, int g) {
  for (int o = 1; o < g; o <<= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// Radius tests of up to 32 staged candidates pc[0], pc[1 << LG], ... against the query P: bit j of the result = candidate j lies
// within the radius.  LG (log2 of the lanes that share a query) is a template parameter so that the four LDS loads of a step carry
// their offsets as immediates; they are issued together, and the step has no branch.  The caller masks the bits of candidates past
// the end of the block (the tile is readable up to PCA_PAD entries beyond it).
template <int LG>
__device__ inline unsigned gh_pca_tests32(const float4* __restrict__ pc, int nj, const float4 P, float r2) {
  unsigned m = 0u;
  int j = 0;
  for (; j + 4 <= nj; j += 4) {
    const float4* q = pc + (j << LG);
    const float4 c0 = q[0], c1 = q[1 << LG], c2 = q[2 << LG], c3 = q[3 << LG];
    float a0 = P.x - c0.x, a1 = P.x - c1.x, a2 = P.x - c2.x, a3 = P.x - c3.x;
    float d0 = a0 * a0, d1 = a1 * a1, d2 = a2 * a2, d3 = a3 * a3;
    a0 = P.y - c0.y; a1 = P.y - c1.y; a2 = P.y - c2.y; a3 = P.y - c3.y;
    d0 += a0 * a0; d1 += a1 * a1; d2 += a2 * a2; d3 += a3 * a3;
    a0 = P.z - c0.z; a1 = P.z - c1.z; a2 = P.z - c2.z; a3 = P.z - c3.z;
    d0 += a0 * a0; d1 += a1 * a1; d2 += a2 * a2; d3 += a3 * a3;
    const unsigned nib = (d0 < r2 ? 1u : 0u) | (d1 < r2 ? 2u : 0u) | (d2 < r2 ? 4u : 0u) | (d3 < r2 ? 8u : 0u);
    m |= nib << j;
  }
  for (; j < nj; j++) {
    const float4 c = pc[j << LG];
    const float a0 = P.x - c.x, a1 = P.y - c.y, a2 = P.z - c.z;
    float d = a0 * a0;
    d += a1 * a1;
    d += a2 * a2;
    m |= (d < r2 ? 1u : 0u) << j;
  }
  return m;
}
__device__ inline unsigned gh_pca_tests32_lg(int lg, const float4* __restrict__ pc, int nj, const float4 P, float r2) {
  switch (lg) {
    case 0: return gh_pca_tests32<0>(pc, nj, P, r2);
    case 1: return gh_pca_tests32<1>(pc, nj, P, r2);
    case 2: return gh_pca_tests32<2>(pc, nj, P, r2);
    case 3: return gh_pca_tests32<3>(pc, nj, P, r2);
    case 4: return gh_pca_tests32<4>(pc, nj, P, r2);
    case 5: return gh_pca_tests32<5>(pc, nj, P, r2);
    default: return gh_pca_tests32<6>(pc, nj, P, r2);
  }
}
// Which occupied cells a workgroup of a PCA launch takes: runs of 8 CONSECUTIVE cells of the (z-fastest) cell list -- a cell's block
// shares two thirds of its nine runs with its z-neighbour's, so the second staging of a run comes out of L1 / L2 instead of the fabric --
// and 64 consecutive runs (one neighbourhood of the scan) go to workgroups of ONE XCD (block b runs on XCD b % 8, observed: for speed only),
// whose L2 then holds the (x +- 1, y +- 1) columns the runs share.  Rounds 3-4 dealt single cells round-robin over all workgroups: every
// XCD fetched every point, counter traffic 9 x the algorithmic bytes (profiles/r04_pmc_*).  Results do not depend on the deal.
// f(c0, cnt): the cells c0 .. c0 + cnt - 1 (cnt <= 8).  `ctr`: eight zeroed counters (one per XCD) -- the runs of an XCD's share are
// handed out one at a time to whichever of its workgroups asks next.  A cell's cost is points x candidates and has a long tail (dense
// patches near the scanner, 50 x 450 against a mean of 14 x 130), and dense cells are neighbours: with a static deal the workgroup that
// drew them finished at 2.3 x the mean workgroup's time, and that was the kernel's time -- the counters of round 5 (SQ_WAVE_CYCLES: 44 %
// of the launched waves resident on average, VALU issue 56 % of peak all the same; profiles/r05_fe_pmc_call3.txt).  One returning atomic
// per 8 cells (round 2 popped single cells from ONE counter: 450 k serialised pops were the kernel's run time).
template <typename F>
__device__ inline void gh_pca_for_my_runs(int nc, int* __restrict__ ctr, F&& f) {
  const int grid = (int)gridDim.x, b = (int)blockIdx.x;
  const int nrun = (nc + 7) >> 3;
  if (grid < 8) {
    for (int run = b; run < nrun; run += grid) f(run << 3, min(8, nc - (run << 3)));
    return;
  }
  const int nlb = grid >> 3, xcd = b & 7, lb = b >> 3;
  if (lb >= nlb) return;  // grid not a multiple of 8: the last few workgroups stay idle
  for (int r = lb;; r += nlb) {
    if (ctr) {
      int t = 0;
      if (threadIdx.x == 0) t = atomicAdd(&ctr[xcd], 1);
      r = __builtin_amdgcn_readfirstlane(t);
    }
    if (((r >> 6) << 9) >= nrun) break;
    const int run = ((r >> 6) << 9) | (xcd << 6) | (r & 63);
    if (run >= nrun) continue;
    f(run << 3, min(8, nc - (run << 3)));
  }
}
constexpr int PCA_PAD = 64;  // a block is kept resident in the tile when it leaves this many entries free: the test steps may read (never use) up to g - 1 entries past its end

// What a cell's pass needs from the cell table: the cell's own points [qb, qe) and, in lanes 0..8, the 9 runs of its 27-cell block.
// gh_pca_meta only ISSUES the loads (one table lookup per lane, all independent); a caller that walks consecutive cells asks for the
// next cell's record before it computes the current one, so that the lookups -- each a trip to L2 or HBM: the table has tens of
// millions of entries per batch -- are in flight during the arithmetic instead of in front of it.
struct PcaMeta {
  unsigned rb_l, re_l, qb, qe;
};
__device__ inline PcaMeta gh_pca_meta(const GridArgs& G, unsigned key, int lane) {
  const int cz = key % G.d.dim[2];
  const int cy = (key / G.d.dim[2]) % G.d.dim[1];
  const int cx = key / (G.d.dim[2] * G.d.dim[1]);
  PcaMeta m;
  m.qb = G.start[key];
  m.qe = G.start[key + 1];
  m.rb_l = 0u; m.re_l = 0u;
  if (lane < 9) {
    const int x = cx - 1 + lane / 3, y = cy - 1 + lane % 3;
    if (x >= 0 && x < G.d.dim[0] && y >= 0 && y < G.d.dim[1]) {
      const int z0 = max(cz - 1, 0), z1 = min(cz + 1, G.d.dim[2] - 1);
      const unsigned base = ((unsigned)x * G.d.dim[1] + y) * G.d.dim[2];
      m.rb_l = G.start[base + z0];
      m.re_l = G.start[base + z1 + 1];
    }
  }
  return m;
}

template <int CHUNK>
__device__ inline void gh_pca_cell_body(const GridArgs& G, unsigned key, const PcaMeta M, float r2, double* __restrict__ scat, int* __restrict__ count, float4* sC,
                                        int lane) {
  const int cz = key % G.d.dim[2];
  const int cy = (key / G.d.dim[2]) % G.d.dim[1];
  const int cx = key / (G.d.dim[2] * G.d.dim[1]);
  const unsigned qb = M.qb, qe = M.qe;
  // The candidates of a cell are the points of its 27-cell block: 9 runs, ~130 points at TLS density (up to ~500).  When they fit the tile
  // they are staged ONCE, run after run, and every sweep of every pass over the cell reads them from LDS; larger blocks go through the
  // tile chunk by chunk.  Either way a lane meets the candidates in the same order.
  const unsigned rb_l = M.rb_l, re_l = M.re_l;
  const unsigned len_l = re_l - rb_l;
  unsigned off_l = len_l;  // inclusive prefix over the lanes (lanes >= 9 add nothing)
  for (int o = 1; o < 16; o <<= 1) {
    const unsigned v = __shfl_up(off_l, o, 64);
    if (lane >= o) off_l += v;
  }
  const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)off_l, 8);
  off_l -= len_l;  // exclusive
  const bool resident = total + (unsigned)PCA_PAD <= (unsigned)CHUNK;
  // the first 64 query points' coordinates: asked for before the staging loads are waited for (one memory round trip, not two)
  float4 P0 = make_float4(0, 0, 0, 0);
  {
    const int np0 = (int)min(64u, qe - qb);
    int g0 = 1;
    while (np0 > 0 && g0 * 2 * np0 <= 64) g0 *= 2;  // (np0 == 0, an empty query cell, never comes from Unique() of occupied keys: any other caller's guard)
    if (lane / g0 < np0) P0 = G.pts[qb + lane / g0];
  }
  if (resident) {
    // staging: tile entry idx comes from run r(idx) = the last run whose offset is <= idx; the source address is idx + (rb_r - off_r).
    // Four loads per lane are in flight at a time (rounds 3-4 walked the runs one after the other: nine dependent load -> store rounds)
    int dsel[9], osel[9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
      osel[r] = __builtin_amdgcn_readlane((int)off_l, r);
      dsel[r] = __builtin_amdgcn_readlane((int)rb_l, r) - osel[r];
    }
    for (int base = 0; base < (int)total; base += 256) {
      float4 v[4];
      int ix[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        ix[u] = base + u * 64 + lane;
        int d = dsel[0];
#pragma unroll
        for (int r = 1; r < 9; r++) d = ix[u] >= osel[r] ? dsel[r] : d;  // (offsets never decrease, so the LAST run with offset <= idx < total is never an empty one)
        v[u] = make_float4(0, 0, 0, 0);
        if (ix[u] < (int)total) v[u] = G.pts[ix[u] + d];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (ix[u] < (int)total) sC[ix[u]] = v[u];
    }
    __syncthreads();
  }
  for (unsigned q0 = qb; q0 < qe; q0 += 64) {
    const int np = (int)min(64u, qe - q0);
    int g = 1, lg = 0;  // g = 1 << lg
    while (g * 2 * np <= 64) { g *= 2; lg++; }
    const int qi = lane / g, sl = lane % g;
    const unsigned q = q0 + qi;
    const bool live = qi < np;
    float4 P = P0;
    if (q0 != qb) {
      P = make_float4(0, 0, 0, 0);
      if (live) P = G.pts[q];
    }
    // ---- sweep 1: neighbour count and centroid (pca.h:151, pcl::PCA mean)
    int k = 0;
    double sx = 0, sy = 0, sz = 0;
    // Resident block (round 5): the sweep is split into TESTS and SUMS.  The tests are f32 only and run four candidates per step with
    // their LDS loads issued together (the rolled loop of rounds 3-4 waited a full LDS round trip per candidate and spent two thirds of
    // its ~33 instructions per candidate on loop control and on the f64 block, which a wave executes whenever ANY of its lanes hits):
    // they leave one bit per candidate a lane tested, the first 128 of them in four registers.  The centroid sums and the scatter sums
    // then walk the set bits (~15 % of the candidates).  A lane meets its neighbours in the same ascending order as before, so the sums
    // are the same bits; the count is the population count of the masks.
    unsigned hm0 = 0u, hm1 = 0u, hm2 = 0u, hm3 = 0u;
    const int per = ((int)total + g - 1) >> lg;                           // candidates a lane tests at most (wave-uniform)
    const int mine = live ? max(0, ((int)total - sl + g - 1) >> lg) : 0;  // ... and how many of them exist for THIS lane (t = sl + (j << lg) < total)
#define GH_PCA_CENTROID(HM, W)                                                           \
    for (unsigned m = HM; m; m &= m - 1u) {                                              \
      const int t = sl + (((W) * 32 + (__ffs((int)m) - 1)) << lg);                       \
      const float4 Cc = sC[t];                                                           \
      sx += (double)Cc.x; sy += (double)Cc.y; sz += (double)Cc.z;                        \
    }
    if (resident) {
      const int nw = min(4, (per + 31) >> 5);
#pragma unroll 1
      for (int W = 0; W < nw; W++) {  // (a run-time loop: one copy of the test steps per lane split, not four)
        unsigned m_ = gh_pca_tests32_lg(lg, sC + sl + ((W * 32) << lg), min(32, per - W * 32), P, r2);
        const int v_ = mine - W * 32;
        m_ &= v_ >= 32 ? ~0u : (v_ <= 0 ? 0u : (1u << v_) - 1u);
        hm0 = W == 0 ? m_ : hm0; hm1 = W == 1 ? m_ : hm1; hm2 = W == 2 ? m_ : hm2; hm3 = W == 3 ? m_ : hm3;
      }
      k = __popc(hm0) + __popc(hm1) + __popc(hm2) + __popc(hm3);
      GH_PCA_CENTROID(hm0, 0)
      GH_PCA_CENTROID(hm1, 1)
      GH_PCA_CENTROID(hm2, 2)
      GH_PCA_CENTROID(hm3, 3)
      if (live) {
        for (int t = sl + (128 << lg); t < (int)total; t += g) {
          const float4 Cc = sC[t];
          const float dx = P.x - Cc.x, dy = P.y - Cc.y, dz = P.z - Cc.z;
          float d2 = dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          if (d2 < r2) { k++; sx += (double)Cc.x; sy += (double)Cc.y; sz += (double)Cc.z; }
        }
      }
#undef GH_PCA_CENTROID
    } else
    gh_for_runs(G.d, G.start, cx, cy, cz, [&](unsigned rb, unsigned re) {
      for (unsigned base = rb; base < re; base += CHUNK) {
        const int cnt = min((unsigned)CHUNK, re - base);
        for (int t = lane; t < cnt; t += 64) sC[t] = G.pts[base + t];
        __syncthreads();
        if (live)
          for (int t = sl; t < cnt; t += g) {
            const float4 Cc = sC[t];
            const float dx = P.x - Cc.x, dy = P.y - Cc.y, dz = P.z - Cc.z;
            float d2 = dx * dx;
            d2 += dy * dy;
            d2 += dz * dz;
            if (d2 < r2) { k++; sx += (double)Cc.x; sy += (double)Cc.y; sz += (double)Cc.z; }
          }
        __syncthreads();
      }
    });
    for (int o = 1; o < g; o <<= 1) k += __shfl_xor(k, o, 64);
    sx = gh_group_sum(sx, g); sy = gh_group_sum(sy, g); sz = gh_group_sum(sz, g);
    const double mx = sx / (double)k, my = sy / (double)k, mz = sz / (double)k;
    // ---- sweep 2: de-meaned scatter
    double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
#define GH_PCA_SWEEP2(HM, W)                                                           \
    for (unsigned m = HM; m; m &= m - 1u) {                                            \
      const int t = sl + (((W) * 32 + (__ffs((int)m) - 1)) << lg);                     \
      const float4 Cc = sC[t];                                                         \
      const double ex = (double)Cc.x - mx, ey = (double)Cc.y - my, ez = (double)Cc.z - mz; \
      s00 += ex * ex; s01 += ex * ey; s02 += ex * ez;                                  \
      s11 += ey * ey; s12 += ey * ez; s22 += ez * ez;                                  \
    }
    if (resident) {
      if (live && k >= 3) {
        GH_PCA_SWEEP2(hm0, 0)
        GH_PCA_SWEEP2(hm1, 1)
        GH_PCA_SWEEP2(hm2, 2)
        GH_PCA_SWEEP2(hm3, 3)
        for (int t = sl + (128 << lg); t < (int)total; t += g) {
          const float4 Cc = sC[t];
          const float dx = P.x - Cc.x, dy = P.y - Cc.y, dz = P.z - Cc.z;
          float d2 = dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          if (d2 < r2) {
            const double ex = (double)Cc.x - mx, ey = (double)Cc.y - my, ez = (double)Cc.z - mz;
            s00 += ex * ex; s01 += ex * ey; s02 += ex * ez;
            s11 += ey * ey; s12 += ey * ez; s22 += ez * ez;
          }
        }
      }
#undef GH_PCA_SWEEP2
    } else
    gh_for_runs(G.d, G.start, cx, cy, cz, [&](unsigned rb, unsigned re) {
      for (unsigned base = rb; base < re; base += CHUNK) {
        const int cnt = min((unsigned)CHUNK, re - base);
        for (int t = lane; t < cnt; t += 64) sC[t] = G.pts[base + t];
        __syncthreads();
        if (live && k >= 3)
          for (int t = sl; t < cnt; t += g) {
            const float4 Cc = sC[t];
            const float dx = P.x - Cc.x, dy = P.y - Cc.y, dz = P.z - Cc.z;
            float d2 = dx * dx;
            d2 += dy * dy;
            d2 += dz * dz;
            if (d2 < r2) {
              const double ex = (double)Cc.x - mx, ey = (double)Cc.y - my, ez = (double)Cc.z - mz;
              s00 += ex * ex; s01 += ex * ey; s02 += ex * ez;
              s11 += ey * ey; s12 += ey * ez; s22 += ez * ez;
            }
          }
        __syncthreads();
      }
    });
    s00 = gh_group_sum(s00, g); s01 = gh_group_sum(s01, g); s02 = gh_group_sum(s02, g);
    s11 = gh_group_sum(s11, g); s12 = gh_group_sum(s12, g); s22 = gh_group_sum(s22, g);
    if (live && sl == 0) {  // the eigen-solve of the scatter is a kernel of its own (gh_pca_eigen_point): here only np lanes of 64 hold a point
      const unsigned orig = __float_as_uint(P.w);
      double* o = scat + (size_t)orig * 6;
      o[0] = s00; o[1] = s01; o[2] = s02; o[3] = s11; o[4] = s12; o[5] = s22;
      count[orig] = k;
    }
  }
}

// One cell, lookups and pass in a row (the single-cloud kernel's and any other caller's entry point)
template <int CHUNK>
__device__ inline void gh_pca_cell(const GridArgs& G, unsigned key, float r2, double* __restrict__ scat, int* __restrict__ count, float4* sC, int lane) {
  const PcaMeta M = gh_pca_meta(G, key, lane);
  gh_pca_cell_body<CHUNK>(G, key, M, r2, scat, count, sC, lane);
}

// pcl::PCA's eigenvalues of one point from its neighbourhood's scatter sums (pca.h:218-223) and the curvature of pca.h:232-239: one
// THREAD per point.  (Inside the per-cell kernel this part -- N2's rounding, 24 Jacobi rotations with their f64 divisions and square
// roots -- ran on the ~16 lanes of a wave that hold a point and was the kernel's run time: neither the lane split of the candidate
// loops, nor the resident tile, nor dropping the work counter moved it by 2 %.)
__device__ inline void gh_pca_eigen_point(const double* __restrict__ scat, const int* __restrict__ count, long long i, float* __restrict__ lambda,
                                          double* __restrict__ curvature) {
  const int k = count[i];
  float l1 = 0.f, l2 = 0.f, l3 = 0.f;
  double cv = 0.0;
  if (k >= 3) {  // pca.h:209
    double S[6];
#pragma unroll
    for (int t = 0; t < 6; t++) S[t] = scat[(size_t)i * 6 + t];
    gh_quant_grid(S, 6);  // N2: pcl::PCA's Matrix3f
    double a00 = (double)(float)S[0], a01 = (double)(float)S[1], a02 = (double)(float)S[2], a11 = (double)(float)S[3],
           a12 = (double)(float)S[4], a22 = (double)(float)S[5];
    double V[9];
    gh_jacobi3(a00, a01, a02, a11, a12, a22, V);
    double e0 = a00, e1 = a11, e2 = a22, t;  // ascending sort
    if (e0 > e1) { t = e0; e0 = e1; e1 = t; }
    if (e1 > e2) { t = e1; e1 = e2; e2 = t; }
    if (e0 > e1) { t = e0; e0 = e1; e1 = t; }
    l1 = (float)e2; l2 = (float)e1; l3 = (float)e0;
    const double d1 = (double)l1, d2_ = (double)l2, d3 = (double)l3;
    cv = ((d1 + d2_ + d3) == 0.0) ? 0.0 : d3 / (d1 + d2_ + d3);  // pca.h:240-247
  }
  lambda[(size_t)i * 3] = l1;
  lambda[(size_t)i * 3 + 1] = l2;
  lambda[(size_t)i * 3 + 2] = l3;
  curvature[i] = cv;
}
