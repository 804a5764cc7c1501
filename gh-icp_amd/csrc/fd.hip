// Feature-distance matrices (once per pair), gfx950.
//   k_fd_bsc_mfma : GHRegistration::calFD_BSC (src/ghicp_reg.cpp:143-200) with
//               StereoBinaryFeature::hammingDistance (src/stereo_binary_feature.cpp:87-104):
//               FD[i][j] = min_v H(S[v][i], T[j]) over the 56 bytes of a string (441 feature bits + 7 pad bits, which the
//               reference's byte-wise xor counts like any other).  The strings are expanded to +-1 int8 vectors (a set bit is +1,
//               a clear bit -1), so that dot(a, b) = #equal - #different = 448 - 2 H and H = (448 - dot) / 2 EXACTLY in i32:
//               the K = 448 dot products of a 64 x 64 tile run on the matrix cores
//               (v_mfma_i32_32x32x32_i8, 14 K-steps per variant).  ONE launch covers every pair of a batch (a flat tile index
//               over the jobs); the tile is written twice from LDS, row-major and transposed, both coalesced -- the loop's column
//               sweep reads the transposed copy (loop.hip).  Integer exact.
//   k_fd_fpfh : GHRegistration::calFD_FPFH (src/ghicp_reg.cpp:202-214) with
//               FPFHfeature::compute_fpfh_distance (include/fpfh.hpp:135-165): |Pearson r| of two
//               33-bin histograms, evaluated per pair in the reference's sequential f32 order, so the
//               result is bit-identical to the scalar loop (an MFMA GEMM of z-scored rows would
//               re-associate the sums).
// Both are HBM-write-bound: 56(V ks + kt) resp. 132(ks + kt) bytes in, 2 (x 2 with the transposed copy) resp. 4 bytes per pair out.
#include "ctx.h"

namespace {

constexpr int TI = 64, TJ = 64;
constexpr int FB_ROW = 448 + 16;  // bytes of an expanded string in LDS: 448 int8 + 16 of padding (rows 116 dwords apart)

typedef int fd_v4i __attribute__((vector_size(16)));
typedef int fd_v16i __attribute__((vector_size(64)));

// 32 bits of a string -> 32 int8 (+1 set, -1 clear; all 0 when `live` is false), as eight dwords of four bytes each
__device__ __forceinline__ void fd_expand32(uint32_t w, bool live, uint32_t* __restrict__ out) {
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const uint32_t nib = (w >> (4 * g)) & 0xFu;
    const uint32_t x = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);  // 0x01 per set bit
    const uint32_t pm = x | ((x ^ 0x01010101u) * 0xFFu);                                         // 0x01 / 0xFF per byte
    out[g] = live ? pm : 0u;
  }
}

// rows [r0, r0 + 64) of `strings` (14 dwords each, `nrows` in all) expanded into a 64 x FB_ROW LDS tile; rows beyond the end are zero
__device__ __forceinline__ void fd_expand_tile(const uint32_t* __restrict__ strings, int r0, int nrows, uint8_t* __restrict__ tile) {
  for (int t = threadIdx.x; t < 64 * 14; t += 256) {
    const int row = t / 14, q = t - row * 14;
    const bool live = r0 + row < nrows;
    const uint32_t w = live ? strings[(size_t)(r0 + row) * 14 + q] : 0u;
    uint32_t e[8];
    fd_expand32(w, live, e);
    uint4* dst = reinterpret_cast<uint4*>(tile + row * FB_ROW + q * 32);
    dst[0] = make_uint4(e[0], e[1], e[2], e[3]);
    dst[1] = make_uint4(e[4], e[5], e[6], e[7]);
  }
}

struct FdBscJob {
  const uint32_t* fS;  // V x ks x 14 dwords
  const uint32_t* fT;  // kt x 14 dwords
  uint16_t* FD;        // ks x kt
  uint16_t* FDt;       // kt x ks, or nullptr
  int ks, kt, V, tile0;  // tile0: index of the job's first tile in the launch's flat tile order
};

// One workgroup = one 64 x 64 tile of one job; wave w computes the 32 x 32 quadrant (w >> 1, w & 1).
// MFMA operands: lane l supplies 16 consecutive int8 of row (l & 31), K-offset (l >> 5) * 16 -- the SAME map for A and B, so the K order
// inside a step cancels in the dot product; C/D: column = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5) for accumulator register r.
template <bool BATCH>
__global__ __launch_bounds__(256) void k_fd_bsc_mfma(const FdBscJob* __restrict__ jobs, int njobs, FdBscJob one) {
  __shared__ __attribute__((aligned(16))) uint8_t sA[64 * FB_ROW];
  __shared__ __attribute__((aligned(16))) uint8_t sB[64 * FB_ROW];
  FdBscJob J = one;
  int tile = blockIdx.x;
  if (BATCH) {
    int lo = 0, hi = njobs - 1;  // last job whose tile0 <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    J = jobs[lo];
    tile = blockIdx.x - J.tile0;
  }
  const int tjn = (J.kt + TJ - 1) / TJ;
  const int i0 = (tile / tjn) * TI, j0 = (tile % tjn) * TJ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  fd_expand_tile(J.fT, j0, J.kt, sB);
  int best[16];
#pragma unroll
  for (int r = 0; r < 16; r++) best[r] = 1 << 30;
  const uint8_t* pa = sA + (wr * 32 + (lane & 31)) * FB_ROW + (lane >> 5) * 16;
  const uint8_t* pb = sB + (wc * 32 + (lane & 31)) * FB_ROW + (lane >> 5) * 16;
  for (int v = 0; v < J.V; v++) {
    __syncthreads();  // the previous variant's operand reads are done
    fd_expand_tile(J.fS + (size_t)v * J.ks * 14, i0, J.ks, sA);
    __syncthreads();
    fd_v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 14; q++) {
      const fd_v4i a = *reinterpret_cast<const fd_v4i*>(pa + q * 32);
      const fd_v4i b = *reinterpret_cast<const fd_v4i*>(pb + q * 32);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int h = (448 - acc[r]) >> 1;
      best[r] = h < best[r] ? h : best[r];
    }
  }
  __syncthreads();
  uint16_t(*sO)[TJ + 2] = reinterpret_cast<uint16_t(*)[TJ + 2]>(sA);  // the tile, staged for the two coalesced writes (8.4 KB over sA)
#pragma unroll
  for (int r = 0; r < 16; r++) sO[wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][wc * 32 + (lane & 31)] = (uint16_t)best[r];
  __syncthreads();
  for (int e = threadIdx.x; e < TI * TJ; e += 256) {
    const int i = e >> 6, j = e & 63;
    if (i0 + i < J.ks && j0 + j < J.kt) J.FD[(size_t)(i0 + i) * J.kt + j0 + j] = sO[i][j];
  }
  if (J.FDt)
    for (int e = threadIdx.x; e < TI * TJ; e += 256) {
      const int j = e >> 6, i = e & 63;
      if (i0 + i < J.ks && j0 + j < J.kt) J.FDt[(size_t)(j0 + j) * J.ks + i0 + i] = sO[i][j];
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
  FdBscJob J = {(const uint32_t*)featS, (const uint32_t*)featT, FD, nullptr, ks, kt, V, 0};
  hipEvent_t kev = ctx->kt_begin(KT_FD_BSC);
  hipLaunchKernelGGL(k_fd_bsc_mfma<false>, dim3(cdiv(kt, TJ) * cdiv(ks, TI)), dim3(256), 0, ctx->stream, (const FdBscJob*)nullptr, 1, J);
  ctx->kt_end(KT_FD_BSC, kev);
  GH_HIP(hipGetLastError());
  return GHICP_OK;
}

// The feature-distance matrices of a whole batch of pairs in ONE launch (+ their transposed copies where FDt[p] is given).
int gh_fd_bsc_batch_dev(ghicp_ctx* ctx, int nb, const gh_fd_bsc_job* jobs) {
  std::vector<FdBscJob> h;
  h.reserve((size_t)nb);
  long long tiles = 0;
  for (int p = 0; p < nb; p++) {
    const gh_fd_bsc_job& j = jobs[p];
    if (j.ks <= 0 || j.kt <= 0) continue;
    if (tiles + (long long)cdiv(j.kt, TJ) * cdiv(j.ks, TI) >= (1ll << 31) - 1) return ctx->fail(GHICP_ERR_ARG, "feature-distance batch: more than 2^31 tiles in one call");
    h.push_back({(const uint32_t*)j.featS, (const uint32_t*)j.featT, j.FD, j.FDt, j.ks, j.kt, j.V, (int)tiles});
    tiles += (long long)cdiv(j.kt, TJ) * cdiv(j.ks, TI);
  }
  if (h.empty()) return GHICP_OK;
  FdBscJob* d;
  GH_TRY(ctx->reserve(B_FD_JOBS, h.size(), &d));
  GH_TRY(ctx->upload_table(h.data(), h.size() * sizeof(FdBscJob), d));  // through the pinned job buffer: no stream synchronisation
  hipEvent_t kev = ctx->kt_begin(KT_FD_BSC);
  hipLaunchKernelGGL(k_fd_bsc_mfma<true>, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, (const FdBscJob*)d, (int)h.size(), h[0]);
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
