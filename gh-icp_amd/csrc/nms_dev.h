// Greedy non-maximum suppression of ONE cloud's rank-ordered candidates by one 256-thread workgroup (keypoint_detect.hpp:149-191);
// shared by the single-cloud kernel (nms.hip) and the batched front end (batch.hip).
#pragma once
#include "grid.h"
#include "devmath.h"

// Exact greedy NMS in ONE launch: a single 256-thread workgroup walks the rank-ordered candidates in chunks of 256.
//   (1) every candidate of the chunk is tested against the keypoints selected so far that lie in the 3 x 3 COLUMNS (x, y cells of side
//       R, any z) around it: per-column linked lists in LDS, ~2 keypoints to test instead of the several hundred of the whole list
//       (round 2 swept the whole LDS list: 3.6 ms per cloud at cfg2).  Clouds with more columns than NMS_COL_CAP sweep the list as
//       before, and beyond NMS_SEL_CAP keypoints the per-cell linked lists in global memory take over;
//   (2) the four waves then take turns in rank order: a wave first drops lanes that lie within R of a keypoint a
//       previous wave of this chunk just added, then resolves its own 64 lanes greedily with ballots -- the lowest
//       surviving lane is selected, its coordinates are broadcast with v_readlane, lanes within R die, repeat;
//   (3) winners are appended (rank order) to the output, the LDS list and the global grid.
//   (Round 5, measured and dropped: chunks of 1024 candidates with ONE wave walking the sixteen groups of survivors in rank order after a
//   parallel pre-test -- two barriers per 1024 candidates instead of five per 256 -- 1.62 -> 2.48 ms per 32 clouds: the second look into the
//   column lists and the serial groups cost more than the barriers they replace; profiles/r05_kernel_stats_fe_one_stream_call7.txt.)
constexpr int NMS_T = 256;
constexpr int NMS_SEL_CAP = 3072;  // selected keypoints kept in LDS as float4 (48 KB)
constexpr int NMS_COL_CAP = 16384; // (x, y) columns with an LDS list head (32 KB): 190 m x 190 m at R = 1.5 m
constexpr unsigned short NMS_NONE = 0xFFFF;
// cand[ord[r]] - idx_sub = the point index that is written for rank r (idx_sub: offset of the cloud in a batch's concatenated arrays)
__device__ inline void gh_nms_greedy_cloud(const float* __restrict__ cpts, int c, const GridDesc& g, float r2, int* __restrict__ head,
                                           int* __restrict__ next, const int* __restrict__ cand, const int* __restrict__ ord,
                                           int* __restrict__ kp, int* __restrict__ kcount, int idx_sub) {

  __shared__ float4 sel_pts[NMS_SEL_CAP];
  __shared__ unsigned short col_head[NMS_COL_CAP], sel_next[NMS_SEL_CAP];
  __shared__ int s_nsel;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncol = g.dim[0] * g.dim[1];
  const bool cols = ncol <= NMS_COL_CAP;
  if (tid == 0) s_nsel = 0;
  if (cols)
    for (int i = tid; i < ncol; i += NMS_T) col_head[i] = NMS_NONE;
  __syncthreads();
  // keypoints of the 3 x 3 columns around (px, py): anything within R of the point lies there (column side >= R)
  auto col_hit = [&](float px, float py, float pz) -> bool {
    const int cx = gh_cell_coord(px, g.mn[0], g.inv, g.dim[0]);
    const int cy = gh_cell_coord(py, g.mn[1], g.inv, g.dim[1]);
    for (int x = max(cx - 1, 0); x <= min(cx + 1, g.dim[0] - 1); x++)
      for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dim[1] - 1); y++)
        for (unsigned j = col_head[x * g.dim[1] + y]; j != NMS_NONE; j = sel_next[j]) {
          const float4 q = sel_pts[j];
          const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
          float d2 = dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          if (d2 < r2) return true;
        }
    return false;
  };
  auto grid_hit = [&](float px, float py, float pz) -> bool {
    const int cx = gh_cell_coord(px, g.mn[0], g.inv, g.dim[0]);
    const int cy = gh_cell_coord(py, g.mn[1], g.inv, g.dim[1]);
    const int cz = gh_cell_coord(pz, g.mn[2], g.inv, g.dim[2]);
    for (int x = max(cx - 1, 0); x <= min(cx + 1, g.dim[0] - 1); x++)
      for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dim[1] - 1); y++)
        for (int z = max(cz - 1, 0); z <= min(cz + 1, g.dim[2] - 1); z++)
          for (int j = __hip_atomic_load(&head[((unsigned)x * g.dim[1] + y) * g.dim[2] + z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); j >= 0;
               j = __hip_atomic_load(&next[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const float dx = cpts[(size_t)j * 3] - px, dy = cpts[(size_t)j * 3 + 1] - py, dz = cpts[(size_t)j * 3 + 2] - pz;
            float d2 = dx * dx;
            d2 += dy * dy;
            d2 += dz * dz;
            if (d2 < r2) return true;
          }
    return false;
  };
  for (int base = 0; base < c; base += NMS_T) {
    const int r = base + tid;
    bool alive = r < c;
    float px = 0, py = 0, pz = 0;
    if (alive) { px = cpts[(size_t)r * 3]; py = cpts[(size_t)r * 3 + 1]; pz = cpts[(size_t)r * 3 + 2]; }
    const int nsel0 = s_nsel;  // keypoints selected before this chunk
    if (alive) {
      if (cols && nsel0 <= NMS_SEL_CAP) {
        alive = !col_hit(px, py, pz);
      } else if (nsel0 <= NMS_SEL_CAP) {
        // no early exit inside a group of 8: the broadcast ds_read_b128 of a group are issued back to back
        bool hit = false;
        for (int j0 = 0; j0 < nsel0 && !hit; j0 += 8) {
#pragma unroll
          for (int t = 0; t < 8; t++) {
            const float4 q = sel_pts[min(j0 + t, nsel0 - 1)];
            const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
            float d2 = dx * dx;
            d2 += dy * dy;
            d2 += dz * dz;
            hit |= d2 < r2;
          }
        }
        alive = !hit;
      } else {
        alive = !grid_hit(px, py, pz);
      }
    }
    for (int gw = 0; gw < NMS_T / 64; gw++) {
      __syncthreads();
      if (wave == gw) {
        const int nsel1 = s_nsel;  // includes what earlier waves of this chunk added
        if (alive) {
          if (nsel1 <= NMS_SEL_CAP) {
            for (int j = nsel0; j < nsel1; j++) {
              const float dx = sel_pts[j].x - px, dy = sel_pts[j].y - py, dz = sel_pts[j].z - pz;
              float d2 = dx * dx;
              d2 += dy * dy;
              d2 += dz * dz;
              if (d2 < r2) { alive = false; break; }
            }
          } else if (nsel1 > nsel0) {
            alive = !grid_hit(px, py, pz);
          }
        }
        // in-wave greedy over rank-ordered lanes
        unsigned long long m = __ballot(alive);
        unsigned long long picked = 0ull;
        while (m) {
          const int k = (int)__ffsll((long long)m) - 1;
          picked |= 1ull << k;
          const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px), k));
          const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py), k));
          const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz), k));
          if (alive && lane > k) {
            const float dx = sx - px, dy = sy - py, dz = sz - pz;
            float d2 = dx * dx;
            d2 += dy * dy;
            d2 += dz * dz;
            if (d2 < r2) alive = false;
          }
          m = __ballot(alive) & ~((2ull << k) - 1ull);
        }
        const bool sel = (picked >> lane) & 1ull;
        const int pos = nsel1 + __popcll(picked & ((1ull << lane) - 1ull));
        if (sel) {
          kp[pos] = cand[ord[r]] - idx_sub;
          if (pos < NMS_SEL_CAP) sel_pts[pos] = make_float4(px, py, pz, 0.f);
          const int cx = gh_cell_coord(px, g.mn[0], g.inv, g.dim[0]);
          const int cy = gh_cell_coord(py, g.mn[1], g.inv, g.dim[1]);
          const int cz = gh_cell_coord(pz, g.mn[2], g.inv, g.dim[2]);
          const int old = atomicExch(&head[((unsigned)cx * g.dim[1] + cy) * g.dim[2] + cz], r);
          __hip_atomic_store(&next[r], old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cols) {  // the winners of this wave into their column lists, one at a time (two winners may share a column)
          for (unsigned long long pk = picked; pk; pk &= pk - 1ull) {
            const int k = (int)__ffsll((long long)pk) - 1;
            if (lane == k && pos < NMS_SEL_CAP) {
              const int col = gh_cell_coord(px, g.mn[0], g.inv, g.dim[0]) * g.dim[1] + gh_cell_coord(py, g.mn[1], g.inv, g.dim[1]);
              sel_next[pos] = col_head[col];
              col_head[col] = (unsigned short)pos;
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
        __threadfence_block();
        if (lane == 0) s_nsel = nsel1 + __popcll(picked);
      }
    }
    __syncthreads();
  }
  if (tid == 0) *kcount = s_nsel;
}
