// Internal: the per-cloud front-end cache handle (cloud.hip) -- shared with the batched front end (batch.hip).
#pragma once
#include "ctx.h"

struct ghicp_cloud {
  ghicp_ctx* ctx = nullptr;
  ghicp_pair_config cfg;
  long long n = 0, m = 0, k = 0, cand = 0;
  float bbx = 0.f;
  int V = 1;
  DevBuf ds;    // m float4 (down-sampled points; empty for handles rebuilt from stored features)
  DevBuf kp;    // k int32: keypoint ids into ds
  DevBuf kpx;   // k x 3 f64
  DevBuf feat;  // BSC: 4 x k x 56 bytes (variants 0..V-1 filled) | FPFH: k x 33 f32 | None: empty
};

inline bool same_front_end(const ghicp_pair_config& a, const ghicp_pair_config& b) {
  return a.reg.feature == b.reg.feature && a.reg.dof == b.reg.dof && a.reg.radius_nonmax == b.reg.radius_nonmax && a.voxel == b.voxel &&
         a.neighborhood_radius == b.neighborhood_radius && a.ratio_max == b.ratio_max && a.min_neighbors == b.min_neighbors &&
         (a.reg.feature != GHICP_FEATURE_BSC || memcmp(a.pattern, b.pattern, sizeof(a.pattern)) == 0);
}
