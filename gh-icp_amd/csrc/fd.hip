// Feature-distance matrices (once per pair), gfx950.
//   k_fd_bsc  : GHRegistration::calFD_BSC (src/ghicp_reg.cpp:143-200) with
//               StereoBinaryFeature::hammingDistance (src/stereo_binary_feature.cpp:87-104):
//               FD[i][j] = min_v popcount(S[v][i] xor T[j]), 441-bit strings = 14 dwords. Integer exact.
//   k_fd_fpfh : GHRegistration::calFD_FPFH (src/ghicp_reg.cpp:202-214) with
//               FPFHfeature::compute_fpfh_distance (include/fpfh.hpp:135-165): |Pearson r| of two
//               33-bin histograms, evaluated per pair in the reference's sequential f32 order, so the
//               result is bit-identical to the scalar loop (an MFMA GEMM of z-scored rows would
//               re-associate the sums).
// Both are HBM-write-bound: 56(V ks + kt) resp. 132(ks + kt) bytes in, 2 resp. 4 bytes per pair out.
#include "ctx.h"

namespace {

constexpr int TI = 64, TJ = 64;

__global__ __launch_bounds__(256) void k_fd_bsc(const uint32_t* __restrict__ fS, int ks, int V, const uint32_t* __restrict__ fT, int kt,
                                                uint16_t* __restrict__ FD) {
  __shared__ uint32_t sS[4 * TI * 14];
  const int i0 = blockIdx.y * TI, j0 = blockIdx.x * TJ;
  const int ni = min(TI, ks - i0);
  for (int t = threadIdx.x; t < V * ni * 14; t += 256) {
    const int v = t / (ni * 14), r = t % (ni * 14);
    sS[v * TI * 14 + r] = fS[((size_t)v * ks + i0) * 14 + r];
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = j0 + tx;
  if (j >= kt) return;
  uint32_t b[14];
#pragma unroll
  for (int q = 0; q < 14; q++) b[q] = fT[(size_t)j * 14 + q];
  for (int ii = ty; ii < ni; ii += 4) {
    int best = 1 << 30;
    for (int v = 0; v < V; v++) {
      const uint32_t* a = &sS[v * TI * 14 + ii * 14];
      int h = 0;
#pragma unroll
      for (int q = 0; q < 14; q++) h += __popc(a[q] ^ b[q]);
      best = min(best, h);
    }
    FD[(size_t)(i0 + ii) * kt + j] = (uint16_t)best;
  }
}

__global__ __launch_bounds__(256) void k_fd_fpfh(const float* __restrict__ hS, int ks, const float* __restrict__ hT, int kt,
                                                 float* __restrict__ FD) {
  __shared__ float sS[TI * 33];
  __shared__ float sM[TI];
  const int i0 = blockIdx.y * TI, j0 = blockIdx.x * TJ;
  const int ni = min(TI, ks - i0);
  for (int t = threadIdx.x; t < ni * 33; t += 256) sS[t] = hS[(size_t)i0 * 33 + t];
  __syncthreads();
  if (threadIdx.x < ni) {
    float m = 0;
    for (int q = 0; q < 33; q++) m += sS[threadIdx.x * 33 + q];
    sM[threadIdx.x] = m / 33;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = j0 + tx;
  if (j >= kt) return;
  float h2[33];
  float m2 = 0;
#pragma unroll
  for (int q = 0; q < 33; q++) { h2[q] = hT[(size_t)j * 33 + q]; m2 += h2[q]; }
  m2 /= 33;
  for (int ii = ty; ii < ni; ii += 4) {
    const float m1 = sM[ii];
    float up = 0, d1 = 0, d2 = 0;
#pragma unroll
    for (int q = 0; q < 33; q++) {
      const float a = sS[ii * 33 + q] - m1, b = h2[q] - m2;
      up += a * b;
      d1 += a * a;
      d2 += b * b;
    }
    FD[(size_t)(i0 + ii) * kt + j] = fabsf(up / sqrtf(d1 * d2));
  }
}

}  // namespace

int gh_fd_bsc_dev(ghicp_ctx* ctx, const uint8_t* featS, int ks, int V, const uint8_t* featT, int kt, uint16_t* FD) {
  if (ks <= 0 || kt <= 0) return GHICP_OK;
  dim3 g(cdiv(kt, TJ), cdiv(ks, TI));
  hipEvent_t kev = ctx->kt_begin(KT_FD_BSC);
  hipLaunchKernelGGL(k_fd_bsc, g, dim3(256), 0, ctx->stream, (const uint32_t*)featS, ks, V, (const uint32_t*)featT, kt, FD);
  ctx->kt_end(KT_FD_BSC, kev);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

int gh_fd_fpfh_dev(ghicp_ctx* ctx, const float* histS, int ks, const float* histT, int kt, float* FD) {
  if (ks <= 0 || kt <= 0) return GHICP_OK;
  dim3 g(cdiv(kt, TJ), cdiv(ks, TI));
  hipLaunchKernelGGL(k_fd_fpfh, g, dim3(256), 0, ctx->stream, histS, ks, histT, kt, FD);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

extern "C" int ghicp_fd_bsc(ghicp_ctx* ctx, const uint8_t* featS, int64_t ks, int V, const uint8_t* featT, int64_t kt, uint16_t* FD) {
  GH_ENTER(ctx);
  GH_ARG(ks >= 0 && kt >= 0 && V >= 1 && V <= 4 && ks < (1 << 24) && kt < (1 << 24));
  Stager sg(ctx);
  const uint8_t *dS, *dT;
  uint16_t* dF;
  GH_TRY(sg.in(featS, (size_t)V * ks * 56, &dS));
  GH_TRY(sg.in(featT, (size_t)kt * 56, &dT));
  GH_TRY(sg.out(FD, (size_t)ks * kt, &dF));
  GH_TRY(gh_fd_bsc_dev(ctx, dS, (int)ks, V, dT, (int)kt, dF));
  return sg.finish();
}

extern "C" int ghicp_fd_fpfh(ghicp_ctx* ctx, const float* histS, int64_t ks, const float* histT, int64_t kt, float* FD) {
  GH_ENTER(ctx);
  GH_ARG(ks >= 0 && kt >= 0 && ks < (1 << 24) && kt < (1 << 24));
  Stager sg(ctx);
  const float *dS, *dT;
  float* dF;
  GH_TRY(sg.in(histS, (size_t)ks * 33, &dS));
  GH_TRY(sg.in(histT, (size_t)kt * 33, &dT));
  GH_TRY(sg.out(FD, (size_t)ks * kt, &dF));
  if (ks > 0 && kt > 0) {
    dim3 g(cdiv(kt, TJ), cdiv(ks, TI));
    hipLaunchKernelGGL(k_fd_fpfh, g, dim3(256), 0, ctx->stream, dS, (int)ks, dT, (int)kt, dF);
    GH_HIP(hipGetLastError());
  }
  return sg.finish();
}
