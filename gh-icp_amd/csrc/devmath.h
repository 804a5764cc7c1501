// Small linear algebra (host + device: the same source serves the kernels and the host-side entry points, so that both
// follow one numerics contract) and device-side wave/block reductions for gfx950 (wave64).
// Build flag contract: -ffp-contract=off (no implicit FMA), IEEE f32/f64 div & sqrt.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#define GH_WAVE 64

// ---- 3x3 symmetric eigen-decomposition, cyclic Jacobi in f64 (DESIGN.md "numerics contract" N3):
// pivots (0,1),(0,2),(1,2), 8 sweeps, exactly-zero pivots skipped, t = sgn(th)/(|th|+sqrt(th^2+1)).
// m = {a00,a01,a02,a11,a12,a22} in/out (diagonal ends as eigenvalues); v row-major eigenvectors in columns.
__host__ __device__ inline void gh_jacobi3(double& a00, double& a01, double& a02, double& a11, double& a12, double& a22, double v[9]) {
  v[0] = 1; v[1] = 0; v[2] = 0; v[3] = 0; v[4] = 1; v[5] = 0; v[6] = 0; v[7] = 0; v[8] = 1;
#define GH_ROT(app, aqq, apq, arp, arq, P, Q)                  \
  if (apq != 0.0) {                                            \
    const double th = (aqq - app) / (2.0 * apq);               \
    double t = 1.0 / (fabs(th) + sqrt(th * th + 1.0));         \
    if (th < 0.0) t = -t;                                      \
    const double c = 1.0 / sqrt(t * t + 1.0);                  \
    const double s = t * c;                                    \
    app = app - t * apq;                                       \
    aqq = aqq + t * apq;                                       \
    const double rp = arp, rq = arq;                           \
    arp = c * rp - s * rq;                                     \
    arq = s * rp + c * rq;                                     \
    apq = 0.0;                                                 \
    _Pragma("unroll") for (int i = 0; i < 3; i++) {            \
      const double vp = v[i * 3 + P], vq = v[i * 3 + Q];       \
      v[i * 3 + P] = c * vp - s * vq;                          \
      v[i * 3 + Q] = s * vp + c * vq;                          \
    }                                                          \
  }
  for (int sweep = 0; sweep < 8; sweep++) {
    GH_ROT(a00, a11, a01, a02, a12, 0, 1)  // (p,q)=(0,1), r=2: a[r][p]=a02, a[r][q]=a12
    GH_ROT(a00, a22, a02, a01, a12, 0, 2)  // (0,2), r=1: a[r][p]=a01, a[r][q]=a12
    GH_ROT(a11, a22, a12, a01, a02, 1, 2)  // (1,2), r=0: a[r][p]=a01, a[r][q]=a02
  }
#undef GH_ROT
}

// N2: round n f64 sums onto the f32 grid of the matrix scale: nearest multiple of 2^(e-23), e = exponent of
// the largest |entry| (ties to even).  See DESIGN.md "numerics contract".
__host__ __device__ inline void gh_quant_grid(double* v, int n) {
  double mx = 0;
  for (int i = 0; i < n; i++) mx = fmax(mx, fabs(v[i]));
  if (!(mx > 0) || !(mx <= 1.7976931348623157e308)) return;  // zero, infinite (or all-NaN) input: left untouched
  int e;
  frexp(mx, &e);
  const double q = ldexp(1.0, e - 1 - 23);
  for (int i = 0; i < n; i++) v[i] = rint(v[i] / q) * q;
}

// Closest rotation to the 3x3 cross-covariance A (row-major), Kabsch via Jacobi on A^T A:
// right singular vectors sorted by descending eigenvalue (stable), u1 = A v1/|.|, u2 = GS(A v2),
// u3 = u1 x u2, R = [u1 u2 u3] diag(1,1,sign det V) V^T.   (N5)
__host__ __device__ inline void gh_kabsch(const double A[9], double R[9]) {
  double m[6];
  {
    double ata[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += A[k * 3 + i] * A[k * 3 + j];
        ata[i * 3 + j] = s;
      }
    m[0] = ata[0]; m[1] = ata[1]; m[2] = ata[2]; m[3] = ata[4]; m[4] = ata[5]; m[5] = ata[8];
  }
  double V[9];
  gh_jacobi3(m[0], m[1], m[2], m[3], m[4], m[5], V);
  const double ev[3] = {m[0], m[3], m[5]};
  int ord[3] = {0, 1, 2};
  for (int i = 1; i < 3; i++)
    for (int j = i; j > 0 && ev[ord[j]] > ev[ord[j - 1]]; j--) { int t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t; }
  double v[3][3];
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < 3; i++) v[c][i] = V[i * 3 + ord[c]];
  const double detV = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                      v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
  double u[3][3];
  for (int c = 0; c < 2; c++)
    for (int i = 0; i < 3; i++) u[c][i] = A[i * 3 + 0] * v[c][0] + A[i * 3 + 1] * v[c][1] + A[i * 3 + 2] * v[c][2];
  const double n0 = sqrt(u[0][0] * u[0][0] + u[0][1] * u[0][1] + u[0][2] * u[0][2]);
  if (n0 > 0) {
    for (int i = 0; i < 3; i++) u[0][i] /= n0;
  } else {
    u[0][0] = 1; u[0][1] = 0; u[0][2] = 0;
  }
  const double d01 = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
  for (int i = 0; i < 3; i++) u[1][i] -= d01 * u[0][i];
  const double n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
  if (n1 > 1e-300) {
    for (int i = 0; i < 3; i++) u[1][i] /= n1;
  } else {
    int k = 0;
    if (fabs(u[0][1]) < fabs(u[0][k])) k = 1;
    if (fabs(u[0][2]) < fabs(u[0][k])) k = 2;
    double e[3] = {0, 0, 0};
    e[k] = 1;
    const double d = u[0][k];
    for (int i = 0; i < 3; i++) u[1][i] = e[i] - d * u[0][i];
    const double nn = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
    for (int i = 0; i < 3; i++) u[1][i] /= nn;
  }
  u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
  u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
  u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  const double sgn = (detV < 0) ? -1.0 : 1.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i * 3 + j] = u[0][i] * v[0][j] + u[1][i] * v[1][j] + sgn * u[2][i] * v[2][j];
}

// ---- N7: atan2f of the SPFH angle feature (pcl::computePairFeatures: f1 = atan2f(w . n2, u . n2)).  glibc's atan2f and the device
// library's differ by an ulp on some inputs, and an ulp moves a pair feature across a histogram bin edge.  Both sides of the parity
// test therefore evaluate THIS function (the oracle holds its own transcription): atan2 in f64 from +, -, *, / only -- octant
// reduction, t' = (t - 1) / (t + 1) above tan(pi / 8), 19-term odd series in Horner form, |t'| <= 0.4143 -> truncation < 1e-16 --
// rounded once to f32.  Against a correctly rounded atan2f it differs by at most one f32 ulp, on ~1e-5 of the inputs
// (tests/test_oracle_cpu.py); IEEE special cases (signed zeros, infinities, NaN) follow C99 F.9.1.4.
__host__ __device__ inline float gh_atan2f(float yf, float xf) {
  if (yf != yf || xf != xf) return yf + xf;
  const double PI = 3.14159265358979323846, PI_2 = 1.57079632679489661923, PI_4 = 0.78539816339744830962;
  const double y = (double)yf, x = (double)xf;
  const double ay = fabs(y), ax = fabs(x);
  double r;
  if (ay == 0.0 && ax == 0.0) {
    r = 0.0;
  } else if (ax == INFINITY && ay == INFINITY) {
    r = PI_4;
  } else {
    const double hi = ax > ay ? ax : ay, lo = ax > ay ? ay : ax;
    double t = hi == INFINITY ? 0.0 : lo / hi, base = 0.0;
    if (t > 0.41421356237309503) { t = (t - 1.0) / (t + 1.0); base = PI_4; }
    const double z = t * t;
    double p = 1.0 / 37.0;
#pragma unroll
    for (int k = 35; k >= 3; k -= 2) p = 1.0 / (double)k - z * p;
    r = base + (t - t * z * p);
    if (ay > ax) r = PI_2 - r;
  }
  if (xf < 0.0f || (xf == 0.0f && copysignf(1.0f, xf) < 0.0f)) r = PI - r;
  const float rf = (float)r;
  return copysignf(1.0f, yf) < 0.0f ? -rf : rf;
}

// ---- reductions (fixed tree order => run-to-run deterministic)
__device__ inline double gh_wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}
__device__ inline int gh_wave_sum_i(int x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}
// block sum for blockDim.x <= 1024; scratch must hold 16 doubles; result valid in every thread.
__device__ inline double gh_block_sum(double x, double* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  x = gh_wave_sum(x);
  __syncthreads();  // protect scratch reuse
  if (lane == 0) scratch[wid] = x;
  __syncthreads();
  double r = 0;
  for (int w = 0; w < nw; w++) r += scratch[w];
  return r;
}
__device__ inline double gh_block_min(double x, double* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_down(x, o, 64));
  __syncthreads();
  if (lane == 0) scratch[wid] = x;
  __syncthreads();
  double r = scratch[0];
  for (int w = 1; w < nw; w++) r = fmin(r, scratch[w]);
  return r;
}
// exclusive block scan of int flags/counts (blockDim.x <= 1024); scratch: 17 ints. returns exclusive prefix, *total = block total
__device__ inline int gh_block_excl_scan(int x, int* scratch, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  int incl = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(incl, o, 64);
    if (lane >= o) incl += y;
  }
  __syncthreads();
  if (lane == 63) scratch[wid] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < nw; w++) {
    if (w < wid) base += scratch[w];
    tot += scratch[w];
  }
  *total = tot;
  return base + incl - x;
}
