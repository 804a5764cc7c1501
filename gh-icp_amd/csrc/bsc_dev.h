// Binary Shape Context of ONE keypoint by one 256-thread workgroup (bfe:603-837); shared by the single-cloud kernel (bsc.hip) and the
// batched front end (batch.hip).  feat: [variant][K][56] of the keypoint's cloud, lcs: [K][12] (origin in 9..11 on entry).
#pragma once
#include "grid.h"
#include "devmath.h"
#include "bsc_exp_table.h"

struct BscConst {
  float R, r2s, u, den, r2c, area;
  double radius_w;
  double dunit, dinv;  // 2^(24 - e) (depth units per metre, e = binary exponent of R / 2) and 2^(e - 54) (metres per unit of depth x weight): the exact depth sums below
  float centre[7];
  int pattern[98];
  int K, nvar;
};

constexpr int BT = 256;
// The sphere's points are compacted into LDS once (x, y, z, squared distance; after the covariance sweep: their LCS coordinates), and
// the cell sweep works on (point, plane, 3 x 3 neighbour cell) items so that the f64 exp of the Gaussian weights -- where the kernel's
// time goes -- runs on nearly full waves (see the sweep).  Measured on 32 cfg2 clouds (profiles/r03_kernel_stats_fe_one_stream*.txt):
// point-centric with a 7 x 7 scan per point (rounds 1-2) 7.0 ms, lane-per-cell with register sums 6.4 ms -- both evaluate the exp under
// ~1-in-6 divergence.  Spheres with more points than BSC_CAP keep the point-centric global sweeps.
constexpr int BSC_CAP = 2048;
// The compacted sphere goes through LDS in CHUNKS (round 4): 8 KB instead of 32 KB, 13 KB per workgroup in all, so eight workgroups fit a
// CU (registers) where four did (LDS): 3.97 -> 3.64 ms per 32 clouds -- and a front-end kernel this small can sit beside the solve slots' LDS
// (DESIGN.md §8).  A chunk is
// a multiple of the workgroup size, so thread t meets the points t, t + 256, ... of the list in ascending order exactly as before.
constexpr int BSC_CHUNK = 512;

int gh_bsc_make_const(ghicp_ctx* ctx, float R, int dof, const int32_t* pattern_host, BscConst* out, float* r_search);  // bsc.hip

// Numerics contract N4: the Gaussian cell weight expf(-dd / den) of bfe:239 is evaluated as exp(-j / 32) * exp(s), x = -j / 32 + s
// with j = the nearest grid point (so |s| <= 1 / 64 and the split is EXACT: x is a float, j / 32 has few bits), the first factor from a table
// of correctly rounded doubles, the second a degree-6 Taylor polynomial in Horner form (truncation 4.5e-17), from * and + only, rounded once
// to f32.  The CPU restatement the parity tests check against evaluates the same expression, so both sides hold the same bits; against a
// correctly rounded expf it differs on < 1e-7 of the arguments by one ulp.  x in [-4.5, 0] (dd < (1.5 u)^2, den = u^2 / 2).
static __device__ const double gh_bsc_exp_tab[145] = {GH_BSC_EXP_TABLE};
__device__ inline float gh_bsc_expf(float xf, const double* __restrict__ tab) {
  const double x = (double)xf;
  int j = (int)(x * -32.0 + 0.5);
  j = j < 0 ? 0 : (j > 144 ? 144 : j);
  const double s = x + (double)j * 0.03125;
  double p = s * (1.0 / 720.0) + (1.0 / 120.0);
  p = s * p + (1.0 / 24.0);
  p = s * p + (1.0 / 6.0);
  p = s * p + 0.5;
  p = s * p + 1.0;
  p = s * p + 1.0;
  return (float)(tab[j] * p);
}

// The depth sum of a cell, sum(depth * weight) (bfe:247), EXACTLY: depth = fl(loc + R) is a multiple of ulp(R / 2) = 2^(e - 24) (e = binary
// exponent of R / 2) below 2^27 of them, a weight lies in (2^-7, 1] and is a multiple of 2^-30, so with the depth shifted by 2^27 units every
// product is a non-negative INTEGER below 2^58.  The products are added into a 64-bit LDS word that may wrap; the returning atomic tells
// when it did, and only then (once in a few hundred terms) a carry word is bumped: two LDS atomics per weight as in round 3 (the cell's
// weight sum and this), no rounding, no order.  The shift is taken out again through the weight sum (exact: multiples of 2^-30), and the
// integer is rounded to f64 ONCE.  (Round 3 summed f64 in arrival order, and the GPU test tolerated one bit on 0.5 % of the keypoints.)
__device__ inline void gh_bsc_depth_add(unsigned long long* __restrict__ lo64, unsigned* __restrict__ carry, float depth, float ew, double dunit) {
  const unsigned d_off = (unsigned)((int)((double)depth * dunit) + (1 << 27));  // exact: depth is a multiple of 1 / dunit
  const unsigned w_int = (unsigned)((double)ew * 1073741824.0);                 // exact: ew is a multiple of 2^-30, <= 1
  const unsigned long long term = (unsigned long long)d_off * (unsigned long long)w_int;
  const unsigned long long old = atomicAdd(lo64, term);
  if (old + term < old) atomicAdd(carry, 1u);
}
__device__ inline double gh_bsc_depth_sum(unsigned long long lo64, unsigned carry, double pnum, double dinv) {
  const unsigned __int128 acc = ((unsigned __int128)carry << 64) | (unsigned __int128)lo64;
  const unsigned __int128 shift = (unsigned __int128)(unsigned long long)(pnum * 1073741824.0) << 27;  // 2^27 units x the integer weight sum
  const bool neg = acc < shift;
  const unsigned __int128 mag = neg ? shift - acc : acc - shift;  // < 2^70
  const double d = (double)(unsigned long long)(mag >> 32) * 4294967296.0 + (double)(unsigned long long)(mag & 0xFFFFFFFFull);  // both exact: ONE rounding
  return (neg ? -d : d) * dinv;
}

__device__ inline int rearr_src(int type, int k) {  // bfe:700-739
  switch (type) {
    case 1: return 48 - k;
    case 2: return (6 - k / 7) * 7 + k % 7;
    default: return (k / 7) * 7 + 6 - k % 7;
  }
}

__device__ inline void gh_bsc_keypoint(const GridArgs& G, const BscConst& C, int kk, int K, uint8_t* __restrict__ feat, float* __restrict__ lcs) {

  __shared__ double red[16];
  __shared__ double s_pnum[147], s_exp[145];
  __shared__ unsigned long long s_dlo[147];
  __shared__ unsigned s_dcarry[147];
  __shared__ float s_weight[147], s_depth[147];
  __shared__ float s_axes[9];
  __shared__ double s_stat[3][4];  // per plane: avg_d, sd_d, avg_w, sd_w
  __shared__ unsigned s_bits[4][16];
  __shared__ float4 s_pts[BSC_CHUNK];
  __shared__ int s_scan[17];
  // (round 5, measured and dropped: the three sweeps over the block's points through an iterator that issues four loads of the thread's own
  // sequence at a time -- same order, same bits -- 3.61 -> 3.70 ms per 32 clouds: eight workgroups per CU hide the load latency already,
  // and the kernel is bound by its instruction count, profiles/r05_fe_pmc_call3.txt)
  __shared__ float s_centre[8];  // C.centre, indexed per lane in the cell sweep: out of the kernel arguments (a global load + wait per work item) into LDS
  const int tid = threadIdx.x;
  // the keypoint itself: kp holds ORIGINAL indices; find its coordinates through the original cloud copy kept in pts? -> passed via lcs origin
  const float qx = lcs[(size_t)kk * 12 + 9], qy = lcs[(size_t)kk * 12 + 10], qz = lcs[(size_t)kk * 12 + 11];
  const int cx = gh_cell_coord(qx, G.d.mn[0], G.d.inv, G.d.dim[0]);
  const int cy = gh_cell_coord(qy, G.d.mn[1], G.d.inv, G.d.dim[1]);
  const int cz = gh_cell_coord(qz, G.d.mn[2], G.d.inv, G.d.dim[2]);

  // ---- sweep A
  double sx = 0, sy = 0, sz = 0, sw = 0;
  int cnt = 0;
  gh_for_runs(G.d, G.start, cx, cy, cz, [&](unsigned b, unsigned e) {
    for (unsigned q = b + tid; q < e; q += BT) {
      const float4 P = G.pts[q];
      const float dx = qx - P.x, dy = qy - P.y, dz = qz - P.z;
      float d2 = dx * dx;
      d2 += dy * dy;
      d2 += dz * dz;
      if (d2 < C.r2s) {
        cnt++;
        sx += (double)P.x; sy += (double)P.y; sz += (double)P.z;
        sw += C.radius_w - (double)sqrtf(d2);
      }
    }
  });
  sx = gh_block_sum(sx, red); sy = gh_block_sum(sy, red); sz = gh_block_sum(sz, red); sw = gh_block_sum(sw, red);
  int mm;
  const int my_base = gh_block_excl_scan(cnt, s_scan, &mm);  // thread-major order of the sphere's points: deterministic
  const double mx = sx / (double)mm, my = sy / (double)mm, mz = sz / (double)mm;
  const bool packed = mm <= BSC_CAP;
  // chunk [c0, c0 + BSC_CHUNK) of the sphere's point list into LDS: the same enumeration as the counting sweep, a hit lands at its list
  // position (the thread's base + its running count) when that falls into the chunk
  auto fill_chunk = [&](int c0) {
    int w = my_base;
    gh_for_runs(G.d, G.start, cx, cy, cz, [&](unsigned b, unsigned e) {
      for (unsigned q = b + tid; q < e; q += BT) {
        const float4 P = G.pts[q];
        const float dx = qx - P.x, dy = qy - P.y, dz = qz - P.z;
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        if (d2 < C.r2s) {
          if (w >= c0 && w < c0 + BSC_CHUNK) s_pts[w - c0] = make_float4(P.x, P.y, P.z, d2);
          w++;
        }
      }
    });
    __syncthreads();
  };

  // ---- sweep B
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
  if (mm >= 3 && packed) {
    for (int c0 = 0; c0 < mm; c0 += BSC_CHUNK) {
      fill_chunk(c0);
      const int cn = min(BSC_CHUNK, mm - c0);
      for (int t = tid; t < cn; t += BT) {
        const float4 P = s_pts[t];
        const double w = (double)(float)(C.radius_w - (double)sqrtf(P.w));  // float weight (bfe:975-976), negative beyond sqrt2*R
        const double ex = (double)P.x - mx, ey = (double)P.y - my, ez = (double)P.z - mz;
        c00 += w * ex * ex; c01 += w * ex * ey; c02 += w * ex * ez;
        c11 += w * ey * ey; c12 += w * ey * ez; c22 += w * ez * ez;
      }
      __syncthreads();
    }
  } else if (mm >= 3) {
    gh_for_runs(G.d, G.start, cx, cy, cz, [&](unsigned b, unsigned e) {
      for (unsigned q = b + tid; q < e; q += BT) {
        const float4 P = G.pts[q];
        const float dx = qx - P.x, dy = qy - P.y, dz = qz - P.z;
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        if (d2 < C.r2s) {
          const double w = (double)(float)(C.radius_w - (double)sqrtf(d2));  // float weight (bfe:975-976), negative beyond sqrt2*R
          const double ex = (double)P.x - mx, ey = (double)P.y - my, ez = (double)P.z - mz;
          c00 += w * ex * ex; c01 += w * ex * ey; c02 += w * ex * ez;
          c11 += w * ey * ey; c12 += w * ey * ez; c22 += w * ez * ez;
        }
      }
    });
  }
  if (mm >= 3) {
    c00 = gh_block_sum(c00, red); c01 = gh_block_sum(c01, red); c02 = gh_block_sum(c02, red);
    c11 = gh_block_sum(c11, red); c12 = gh_block_sum(c12, red); c22 = gh_block_sum(c22, red);
  }
  if (tid == 0) {
    float X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1};
    if (mm >= 3) {
      const float da = (float)sw;
      double Cq[6] = {c00, c01, c02, c11, c12, c22};
      gh_quant_grid(Cq, 6);  // N2: Matrix3f covariance
      double a00 = (double)((float)Cq[0] / da), a01 = (double)((float)Cq[1] / da), a02 = (double)((float)Cq[2] / da),
             a11 = (double)((float)Cq[3] / da), a12 = (double)((float)Cq[4] / da), a22 = (double)((float)Cq[5] / da);
      double V[9];
      gh_jacobi3(a00, a01, a02, a11, a12, a22, V);
      const double ev[3] = {a00, a11, a22};
      int imax = 0, imin = 0;
      for (int i = 0; i < 3; i++) {
        if ((float)ev[i] > (float)ev[imax]) imax = i;
        if ((float)ev[i] < (float)ev[imin]) imin = i;
      }
      float P[3], N[3];
      for (int which = 0; which < 2; which++) {
        const int col = which == 0 ? imax : imin;
        const double e[3] = {V[0 * 3 + col], V[1 * 3 + col], V[2 * 3 + col]};
        int b = 0;
        if (fabs(e[1]) > fabs(e[b])) b = 1;
        if (fabs(e[2]) > fabs(e[b])) b = 2;
        const double sg = (e[b] < 0) ? -1.0 : 1.0;
        float* o = which == 0 ? P : N;
        for (int d = 0; d < 3; d++) o[d] = (float)(sg * e[d]);
      }
      const float M0 = P[1] * N[2] - P[2] * N[1], M1 = P[2] * N[0] - P[0] * N[2], M2 = P[0] * N[1] - P[1] * N[0];
      X[0] = P[0]; X[1] = P[1]; X[2] = P[2];
      Y[0] = M0; Y[1] = M1; Y[2] = M2;
      Z[0] = X[1] * Y[2] - X[2] * Y[1];
      Z[1] = X[2] * Y[0] - X[0] * Y[2];
      Z[2] = X[0] * Y[1] - X[1] * Y[0];
      const float nx = sqrtf((X[0] * X[0] + X[1] * X[1]) + X[2] * X[2]);
      const float ny = sqrtf((Y[0] * Y[0] + Y[1] * Y[1]) + Y[2] * Y[2]);
      for (int d = 0; d < 3; d++) { X[d] = X[d] / nx; Y[d] = Y[d] / ny; }
    }
    for (int d = 0; d < 3; d++) { s_axes[d] = X[d]; s_axes[3 + d] = Y[d]; s_axes[6 + d] = Z[d]; }
    float* o = &lcs[(size_t)kk * 12];
    for (int d = 0; d < 3; d++) { o[d] = X[d]; o[3 + d] = Y[d]; o[6 + d] = Z[d]; }
  }
  for (int i = tid; i < 147; i += BT) { s_pnum[i] = 0.0; s_dlo[i] = 0ull; s_dcarry[i] = 0u; }
  for (int i = tid; i < 145; i += BT) s_exp[i] = gh_bsc_exp_tab[i];
  if (tid < 7) s_centre[tid] = C.centre[tid];
  if (tid < 64) s_bits[tid >> 4][tid & 15] = 0u;
  __syncthreads();
  const float X0 = s_axes[0], X1 = s_axes[1], X2 = s_axes[2], Y0 = s_axes[3], Y1 = s_axes[4], Y2 = s_axes[5], Z0 = s_axes[6], Z1 = s_axes[7],
              Z2 = s_axes[8];

  // ---- sweep C: one point of the sphere into the 3 x 7 x 7 cells (bfe:196-373)
  auto splat = [&](float px, float py, float pz) {
    const float d0 = px - qx, d1 = py - qy, d2v = pz - qz;  // bfe:178-180
    const float loc[3] = {(X0 * d0 + X1 * d1) + X2 * d2v, (Y0 * d0 + Y1 * d1) + Y2 * d2v, (Z0 * d0 + Z1 * d1) + Z2 * d2v};
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
      const float a = loc[pl == 2 ? 1 : 0], bb = loc[pl == 0 ? 1 : 2];
      const float depth = loc[pl == 0 ? 2 : (pl == 1 ? 1 : 0)] + C.R;
      for (int j = 0; j < 7; j++) {
        const float dy = bb - C.centre[j];
        const float dy2 = dy * dy;
        if (!(dy2 < C.r2c)) continue;  // dx*dx + dy2 >= dy2 in f32, so this skip is exact
        for (int i = 0; i < 7; i++) {
          const float dx = a - C.centre[i];
          float dd = dx * dx;
          dd += dy2;
          if (dd < C.r2c) {
            const float ew = gh_bsc_expf(-dd / C.den, s_exp);  // expf (bfe:239), contract N4
            atomicAdd(&s_pnum[i + 7 * j + 49 * pl], (double)ew);  // exact in f64 in any order: multiples of 2^-30 below 2^16
            gh_bsc_depth_add(&s_dlo[i + 7 * j + 49 * pl], &s_dcarry[i + 7 * j + 49 * pl], depth, ew, C.dunit);
          }
        }
      }
    }
  };
  if (packed) {
    const int wave = tid >> 6, lane = tid & 63;
    const int sub = lane / 9, slot = lane % 9;
    const int di = slot % 3 - 1, dj = slot / 3 - 1;
    const float inv_u = 1.0f / C.u;
    // Work item = (point, plane, one of the 3 x 3 cells around the point's own cell): a point only reaches cells within 1.5 u of it, i.e.
    // the cell it projects into and that cell's neighbours, so 9 candidates per plane replace the 49-cell scan and ~7 of the 9 lanes
    // of an item evaluate a Gaussian weight.  7 items per wave (63 lanes).  The cell sums are exact, so the chunking does not reach them.
    for (int c0 = 0; c0 < mm; c0 += BSC_CHUNK) {
    fill_chunk(c0);
    const int cn = min(BSC_CHUNK, mm - c0);
    for (int t = tid; t < cn; t += BT) {  // into the LCS (bfe:178-180), in place
      const float4 P = s_pts[t];
      const float d0 = P.x - qx, d1 = P.y - qy, d2v = P.z - qz;
      s_pts[t] = make_float4((X0 * d0 + X1 * d1) + X2 * d2v, (Y0 * d0 + Y1 * d1) + Y2 * d2v, (Z0 * d0 + Z1 * d1) + Z2 * d2v, 0.f);
    }
    __syncthreads();
    const int nitem = cn * 3;
    for (int base = 0; base < nitem; base += 28) {
      const int w = base + wave * 7 + sub;
      if (lane < 63 && w < nitem) {
        const int t = w / 3, pl = w - 3 * t;
        const float4 L = s_pts[t];
        const float a = pl == 2 ? L.y : L.x, bb = pl == 0 ? L.y : L.z;
        const float depth = (pl == 0 ? L.z : (pl == 1 ? L.y : L.x)) + C.R;
        const float ta = (a + C.R) * inv_u, tb = (bb + C.R) * inv_u;
        const float fla = floorf(ta), flb = floorf(tb);
        const int bi = (int)fla, bj = (int)flb;
        auto cell = [&](int ci, int cj) {  // the reference's own test for one cell (bfe:229-247)
          if (ci < 0 || ci >= 7 || cj < 0 || cj >= 7) return;
          const float dy = bb - s_centre[cj];
          const float dy2 = dy * dy;
          const float dx = a - s_centre[ci];
          float dd = dx * dx;
          dd += dy2;
          if (dy2 < C.r2c && dd < C.r2c) {
            const float ew = gh_bsc_expf(-dd / C.den, s_exp);  // expf (bfe:239), contract N4
            atomicAdd(&s_pnum[ci + 7 * cj + 49 * pl], (double)ew);  // exact in f64 in any order: multiples of 2^-30 below 2^16
            gh_bsc_depth_add(&s_dlo[ci + 7 * cj + 49 * pl], &s_dcarry[ci + 7 * cj + 49 * pl], depth, ew, C.dunit);
          }
        };
        cell(bi + di, bj + dj);
        // The ring two cells from the base cell.  In exact arithmetic a point reaches three cells per axis; a point within an ulp of a
        // cell edge can pass the f32 test of a FOURTH (distance 1.5 u, weight e^-4.5), and the reciprocal multiply above can put the base
        // cell one off -- the reference's 7 x 7 scan finds those cells, so the window is completed here.  Only a point whose fractional
        // cell position is within 1e-4 of an edge can reach the ring (the error of ta / tb is ~1e-6), so one compare per axis keeps the
        // ring tests out of all but ~1 % of the waves (measured: the unconditional ring cost 0.47 ms per 32 clouds, profiles/r04_bsc_variants.txt).
        const float fa = ta - fla, fb = tb - flb;
        if (fa < 1e-4f || fa > 0.9999f || fb < 1e-4f || fb > 0.9999f) {
          bool fx = false, fy = false;
          const int i2 = bi + 2 * di, j2 = bj + 2 * dj;
          if (di != 0 && i2 >= 0 && i2 < 7) { const float dx = a - s_centre[i2]; fx = dx * dx < C.r2c; }
          if (dj != 0 && j2 >= 0 && j2 < 7) { const float dy = bb - s_centre[j2]; fy = dy * dy < C.r2c; }
          if (fx) cell(i2, bj + dj);
          if (fy) cell(bi + di, j2);
          if (fx && fy) cell(i2, j2);
        }
      }
    }
    __syncthreads();  // the chunk is consumed: the next one overwrites it
    }
  } else {
    gh_for_runs(G.d, G.start, cx, cy, cz, [&](unsigned b, unsigned e) {
      for (unsigned q = b + tid; q < e; q += BT) {
        const float4 P = G.pts[q];
        const float tx = qx - P.x, ty = qy - P.y, tz = qz - P.z;
        float d2 = tx * tx;
        d2 += ty * ty;
        d2 += tz * tz;
        if (!(d2 < C.r2s)) continue;
        splat(P.x, P.y, P.z);
      }
    });
  }
  __syncthreads();

  // ---- cell quantities (bfe:333-372)
  if (tid < 147) {
    const float ndens = (float)mm / C.area;
    float avg = (float)gh_bsc_depth_sum(s_dlo[tid], s_dcarry[tid], s_pnum[tid], C.dinv);
    avg = (s_pnum[tid] == 0.0) ? 0.0f : (float)((double)avg / s_pnum[tid]);
    const float garea = C.u * C.u;
    const float gdens = (float)(s_pnum[tid] / (double)garea);
    s_weight[tid] = (ndens != 0.0f) ? gdens / ndens : 0.0f;
    s_depth[tid] = avg;
  }
  __syncthreads();
  // ---- per-plane statistics over the 49 pairs, sequential f64 exactly as bfe:500-525
  if (tid < 3) {
    const int off = 49 * tid;
    double avg_d = 0, var_d = 0, avg_w = 0, var_w = 0;
    for (int i = 0; i < 49; i++) {
      const int a = C.pattern[2 * i] + off, b = C.pattern[2 * i + 1] + off;
      avg_d += (double)(s_depth[a] - s_depth[b]);
      avg_w += (double)(s_weight[a] - s_weight[b]);
    }
    avg_d /= 49; avg_w /= 49;
    for (int i = 0; i < 49; i++) {
      const int a = C.pattern[2 * i] + off, b = C.pattern[2 * i + 1] + off;
      const double dd = (double)(s_depth[a] - s_depth[b]), dw = (double)(s_weight[a] - s_weight[b]);
      var_d += (dd - avg_d) * (dd - avg_d);
      var_w += (dw - avg_w) * (dw - avg_w);
    }
    var_d /= 49; var_w /= 49;
    s_stat[tid][0] = avg_d; s_stat[tid][1] = sqrt(var_d); s_stat[tid][2] = avg_w; s_stat[tid][3] = sqrt(var_w);
  }
  __syncthreads();
  // ---- bits (A.2 layout): variant 0 = 147 occupancy + 3 x 49 x (depth, density); variants 1..3 = Q3 layout
  for (int base = 0; base < 448; base += BT) {
    const int k = base + tid;
    for (int v = 0; v < C.nvar; v++) {
      bool bit = false;
      if (v == 0) {
        if (k < 147) bit = s_weight[k] > 0.1f;
        else if (k < 441) {
          const int r = k - 147, pl = r / 98, p = (r % 98) >> 1, off = 49 * pl;
          const int a = C.pattern[2 * p], b = C.pattern[2 * p + 1];
          if ((r & 1) == 0) {
            const double dd = (double)(s_depth[a + off] - s_depth[b + off]);
            bit = fabs(dd - s_stat[pl][0]) > s_stat[pl][1];
          } else if (!(s_weight[a] < 0.1f && s_weight[b] < 0.1f)) {  // Q4: no plane offset (bfe:543)
            const double dw = (double)(s_weight[a + off] - s_weight[b + off]);
            bit = fabs(dw - s_stat[pl][2]) > s_stat[pl][3];
          }
        }
      } else if (k >= 147 && k < 294) {
        static const int TR[4][3] = {{0, 0, 0}, {1, 2, 2}, {3, 2, 1}, {2, 1, 3}};  // bfe:795, 808, 817
        const int r = k - 147, pl = r / 49, kk2 = r % 49;
        bit = s_weight[49 * pl + rearr_src(TR[v][pl], kk2)] > 0.1f;
      }
      const unsigned long long bal = __ballot(bit);
      if ((tid & 63) == 0 && k < 448) {
        s_bits[v][k >> 5] = (unsigned)bal;
        if (k + 32 < 448) s_bits[v][(k >> 5) + 1] = (unsigned)(bal >> 32);
      }
    }
  }
  __syncthreads();
  if (tid < 14 * C.nvar) {
    const int v = tid / 14, w = tid % 14;
    reinterpret_cast<unsigned*>(feat)[((size_t)v * K + kk) * 14 + w] = s_bits[v][w];
  }
}
