// Uniform-grid spatial index on device (stands in for every pcl::KdTreeFLANN the reference builds:
// pca.h:138-139, keypoint_detect.hpp:162-163, binary_feature_extraction.hpp:623-624).
// Points are bucketed by cell, cells keyed z-fastest so that the 27-cell neighbourhood of a query
// is 9 contiguous runs of the sorted point array (3 z-neighbours are adjacent keys).
#pragma once
#include "ctx.h"

#include <cmath>

struct GridDesc {
  float mn[3];
  float inv;       // 1 / cell
  int dim[3];      // cells per axis
  int n;           // points
  unsigned ncell;  // dim0*dim1*dim2
};

struct DeviceGrid {
  GridDesc d;
  const float4* pts;      // n points in cell order: (x, y, z, bits = original index)
  const unsigned* start;  // ncell+1 offsets into pts
  const unsigned* keys;   // n sorted cell keys
};

// what a kernel needs of a grid (by value in the launch arguments, or one entry per cloud of a batch)
struct GridArgs {
  GridDesc d;
  const float4* pts;
  const unsigned* start;
};

// Cell table over the box mm[0..2] .. mm[3..5] for a search radius `cell` (callers pass radius * 1.0001).  The float rounding of
// (v - mn) * inv grows with the cell coordinate (ulp(1000) = 6e-5), so beyond a few hundred cells per axis the margin is widened with
// the extent: two points closer than the radius never end up two cells apart (below 256 cells the cell size -- and with it every
// enumeration order -- is what it always was).  Then the cell is coarsened until the dense table is affordable (a larger cell is
// still exact: superset search).
inline GridDesc gh_grid_desc(const float* mm, long long n, float cell) {
  GridDesc g;
  memset(&g, 0, sizeof(g));
  g.n = (int)n;
  float ext = 0.f;
  for (int d = 0; d < 3; d++) ext = ext > mm[3 + d] - mm[d] ? ext : mm[3 + d] - mm[d];
  const float dims = ext / cell;
  if (dims > 256.f) cell *= 1.0f + 4e-7f * dims;
  for (;;) {
    g.inv = 1.0f / cell;
    unsigned long long nc = 1;
    for (int d = 0; d < 3; d++) {
      g.mn[d] = mm[d];
      g.dim[d] = (int)floorf((mm[3 + d] - mm[d]) * g.inv) + 1;
      if (g.dim[d] < 1) g.dim[d] = 1;
      nc *= (unsigned long long)g.dim[d];
    }
    if (nc <= (1ull << 26)) { g.ncell = (unsigned)nc; break; }
    cell *= 1.5f;
  }
  return g;
}

struct GridSlots {
  BufSlot keys, keys2, vals, vals2, start, pts;
};

// Builds a grid with `cell` >= search radius over xyz (device, n x stride floats). Synchronises once
// (bounding box -> host) to size the cell table.
int gh_grid_build(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float cell, const GridSlots& slots, DeviceGrid* out);
// start[c] = lower_bound(keys, c) for c = 0 .. ncell from the n sorted cell keys (start: ncell + 1 entries); grid.hip
void gh_cell_start_launch(hipStream_t s, const unsigned* keys, unsigned n, unsigned ncell, unsigned* start);
int gh_bbox_dev(ghicp_ctx* ctx, const float* xyz, long long n, int stride, float* mm_host6);

__device__ inline int gh_cell_coord(float v, float mn, float inv, int dim) {
  int c = (int)floorf((v - mn) * inv);
  return min(max(c, 0), dim - 1);
}

// Enumerates the 9 contiguous candidate runs around cell (cx,cy,cz): calls f(begin, end) for each.
template <typename F>
__device__ inline void gh_for_runs(const GridDesc& g, const unsigned* __restrict__ start, int cx, int cy, int cz, F&& f) {
  const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.dim[2] - 1);
  for (int x = max(cx - 1, 0); x <= min(cx + 1, g.dim[0] - 1); x++)
    for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dim[1] - 1); y++) {
      const unsigned base = ((unsigned)x * g.dim[1] + y) * g.dim[2];
      const unsigned b = start[base + z0], e = start[base + z1 + 1];
      if (e > b) f(b, e);
    }
}
