/* =====================================================================================
 *  ghicp_c.h -- C ABI of libghicp_hip.so: the MI355X (gfx950) GH-ICP registration hot path.
 *
 *  This is the drop-in boundary.  The reference has no FFI; its boundary is the
 *  public C++ class API.  Each entry point below names the reference method(s) it replaces
 *  (paths under the reference checkout).  The C++ headers next to this file (ghicp_reg.h, km.h,
 *  keypoint_detect.hpp, binary_feature_extraction.hpp, common_reg.h ...) keep the reference's
 *  class / method names and forward here.
 *
 *  Conventions
 *    - every function returns GHICP_OK (0) or an error code; ghicp_last_error() has the text.
 *    - bulk array arguments are DEVICE pointers by default.  After
 *      ghicp_ctx_set_host_pointers(ctx, 1) they are HOST pointers and the library stages them
 *      (this is what the C++ drop-in classes use).  Scalars / small outputs marked [host] are
 *      always host memory.
 *    - all work is enqueued on the context's stream (ghicp_ctx_set_stream); functions that
 *      return counts synchronise that stream.
 *    - point clouds are `float` with a stride in floats (3 = packed xyz, 4 = float4,
 *      8 = pcl::PointXYZI's 32-byte layout).
 *    - no cwd side effects, no stdout, re-entrant per context (unlike km.cpp:147 / bfe:96).
 * ===================================================================================== */
#ifndef GHICP_C_H_
#define GHICP_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ghicp_ctx ghicp_ctx;

enum {
  GHICP_OK = 0,
  GHICP_ERR_ARG = 1,      /* bad argument */
  GHICP_ERR_HIP = 2,      /* a HIP runtime call failed */
  GHICP_ERR_NO_GPU = 3,   /* no usable gfx950 device: the library has NO CPU fallback */
  GHICP_ERR_CAPACITY = 4, /* an output buffer / internal limit is too small */
  GHICP_ERR_INTERNAL = 5
};

/* include/utility.h:51-57 and :59-64 (same numeric values as the reference enums) */
enum { GHICP_FEATURE_BSC = 0, GHICP_FEATURE_ROPS = 1, GHICP_FEATURE_FPFH = 2, GHICP_FEATURE_NONE = 3 };
enum { GHICP_CORR_NN = 0, GHICP_CORR_NNR = 1, GHICP_CORR_KM = 2 };

/* GHRegistration ctor arguments (include/ghicp_reg.h:77-117) + Energyfunction::init constants
 * (include/ghicp_reg.h:26-41).  ghicp_params_default() fills the reference defaults. */
typedef struct ghicp_params {
  int32_t feature;  /* GHICP_FEATURE_* */
  int32_t corr;     /* GHICP_CORR_*    */
  int32_t dof;      /* 4 or 6 (only selects the number of source BSC variants, ghicp_reg.cpp:178-182) */
  int32_t max_iter; /* guard; the reference loops `while(!converge)` with none (ghicp_reg.cpp:49) */
  float radius_nonmax, adjust_ratio, adjust_step, est_iou;
  float converge_t, converge_r; /* 0.02 m / 0.02 deg (ghicp_reg.h:80) */
  float bbx_magnitude;          /* test/ghicp_main.cpp:91-93 */
  float pad_;
  double penalty_initial, para1, para2, km_eps;
  int32_t min_cor, weight_changing_rate;
} ghicp_params;

/* One iteration of GHRegistration::ghicp_reg (src/ghicp_reg.cpp:49-103): what the reference
 * prints / keeps in `energy, rmse, rmseafter, cor` (include/ghicp_reg.h:147-148). */
typedef struct ghicp_iter {
  int32_t cor, converged;
  double penalty, cdmean, cdstd, rmse, rmse_after, fdm, fdstd, iou, para1, para2, energy;
  double Rt[16]; /* Rt_temp of this iteration, row-major */
} ghicp_iter;

/* ------------------------------------------------------------------ context */
int ghicp_ctx_create(int device, ghicp_ctx** ctx);
int ghicp_ctx_destroy(ghicp_ctx* ctx);
/* The stream every call on this context is issued on.  The persistent pair loop of the Kuhn-Munkres batches forks onto auxiliary streams
 * of its own (one per LDS class of a batch) and joins them back into this stream; those streams follow ghicp_ctx_set_cu_mask, NOT a
 * compute-unit mask the caller may have put on a stream handed in here (the mask of a foreign stream is not inspected). */
int ghicp_ctx_set_stream(ghicp_ctx* ctx, void* hip_stream);
/* Gives the context its own stream restricted to the compute units whose bit is set (32 CUs per word); replaces any stream
 * set before.  Lets a pipeline keep some CUs free of the LDS-filling solve waves for another context's small kernels. */
int ghicp_ctx_set_cu_mask(ghicp_ctx* ctx, const uint32_t* mask, int32_t n_words);
int ghicp_ctx_set_host_pointers(ghicp_ctx* ctx, int on);
/* Host-pointer mode keeps the staged device copy of every LARGE POINT-CLOUD input (xyz rows, >= 256 KB; no other kind of argument)
 * between calls, keyed on host address, size and a fingerprint of the content, so that the reference's call sequence on one cloud -- CFilter::voxelfilter, CKeypointDetect, BSCEncoder /
 * FPFHfeature, transformPointCloud (test/ghicp_main.cpp:86-153) -- uploads it once (at most 1 GiB / 16 arrays, least recently used out
 * first).  Counters since the context was created: inputs served from a kept copy, inputs uploaded, bytes currently kept.  [host] outputs,
 * any may be NULL. */
int ghicp_ctx_stage_stats(const ghicp_ctx* ctx, int64_t* hits, int64_t* misses, int64_t* bytes_kept);
/* Frees every kept copy now (ghicp_ctx_set_host_pointers(ctx, 0) and ghicp_ctx_destroy do the same). */
int ghicp_ctx_stage_clear(ghicp_ctx* ctx);
int ghicp_ctx_synchronize(ghicp_ctx* ctx);
const char* ghicp_last_error(const ghicp_ctx* ctx);
/* Optional per-kernel timing (hipEvent brackets on the context's stream around the named kernels:
 * "pca_cells", "bsc", "km_solve", "cd_rowmin", "km_weights", "fd_bsc", "nms_round", "voxel_sort").
 * ghicp_ctx_kernel_time synchronises the stream; returns total ms and launch count since enabling. */
int ghicp_ctx_kernel_timing(ghicp_ctx* ctx, int on);
int ghicp_ctx_kernel_time(ghicp_ctx* ctx, const char* name, double* total_ms, int64_t* launches);
/* Launch records of the Kuhn-Munkres solve kernel collected while kernel timing is on: out8 = { launches, solves, mean solve ms,
 * mean over launches of the longest solve (ms), mean launch span (ms), solve slots resident on the chip, idle-slot fraction,
 * worst longest/mean ratio of a launch }.  (Diagnostics for the batched loop; the reference has no counterpart.) */
int ghicp_ctx_km_launch_stats(ghicp_ctx* ctx, double* out8);
/* Records of the persistent pair loop (Kuhn-Munkres configurations of ghicp_register_pairs / ghicp_register_clouds: one workgroup =
 * one solve slot runs a pair's whole ghicp_reg loop, src/ghicp_reg.cpp:49-103, and then pops the next pair), collected while kernel
 * timing is on: out8 = { batches, workgroups that ran, solves, mean solve ms, longest solve ms, mean batch span ms, idle-slot fraction
 * (1 - slot lifetimes / (resident slots x span): the tail of a batch), share of the slot lifetimes spent inside Kuhn-Munkres solves }. */
int ghicp_ctx_pair_loop_stats(ghicp_ctx* ctx, double* out8);
/* Diagnostics: Kuhn-Munkres solves of this context's batched loops (ghicp_register_pairs / ghicp_register_clouds, Ct = KM) that went through the
 * solver's literal single-lane fallback since the context was created (rule R4's hazard check, km4_dev.h: correct, slow by design;
 * expected 0 on finite inputs -- a non-zero count explains multi-second solves). */
int ghicp_ctx_loop_hazards(ghicp_ctx* ctx, int64_t* solves /*[host]*/);
/* Scheduling hint for the NEXT batched Kuhn-Munkres registration on this context (ghicp_register_clouds / ghicp_register_pairs with
 * exactly n_pairs pairs; consumed by that call, ignored otherwise): cost[i] = expected relative TIME of pair i in a solve slot, e.g.
 * iterations x keypoints of the same pair last time, or any prior.  The solve slots of the persistent pair loop take the costliest pairs
 * of each class first (graphs that fit four slots per CU form ONE class, i.e. one queue: longest processing time first over the batch),
 * so that the slowest registrations (112 iterations against a mean of 35 on the TLS bench scenes) do not start in the middle of a batch
 * and leave the chip to a handful of stragglers; the share of the CUs the three-slots-per-CU class is confined to follows the hints'
 * sums.  Results do not depend on the hints (pairs share nothing); [host] array. */
int ghicp_ctx_set_loop_cost_hints(ghicp_ctx* ctx, int32_t n_pairs, const float* cost);
/* Diagnostics: the slot timeline of the LAST persistent batch that ran on this context while kernel timing was on -- per pair of the
 * batch (in the order of the call) three int64: when a solve slot took the pair, when it let go (device real-time clock, 100 MHz ticks,
 * bits [51:0]), and the pair's iterations (bits [15:0]).  The spare bits carry what the stragglers of a batch are looked up by:
 * out3[0] [55:52] compute unit, [56] shader array, [59:57] shader engine, [63:60] die (XCC) the slot ran on; out3[2] [31:16] the iteration of
 * the pair's longest Kuhn-Munkres solve, [63:32] that solve's duration in units of 160 ns.
 * out3 may be NULL (only *n_pairs is written); at most cap_pairs triples are copied. */
int ghicp_ctx_loop_timeline(ghicp_ctx* ctx, int64_t* out3, int64_t cap_pairs, int64_t* n_pairs);
/* Progress of the batched loop (ghicp_register_pairs / ghicp_register_clouds) currently running on this context: pairs that are still
 * iterating and pairs of the batch.  No device work; may be called from another thread while the loop runs (a scheduler can start the
 * next batch's front ends when only the slowly converging pairs are left). */
int ghicp_ctx_loop_progress(const ghicp_ctx* ctx, int64_t* active, int64_t* total);
/* Declares a batch of `total` pairs as about to start on this context (active = total) -- for a scheduler that publishes "the batch has
 * started" to other threads BEFORE it calls ghicp_register_clouds, so that they never read the previous batch's finished state. */
int ghicp_ctx_loop_progress_reset(ghicp_ctx* ctx, int64_t total);
const char* ghicp_version(void);
void ghicp_params_default(ghicp_params* p);

/* ------------------------------------------------------------------ front end (per cloud) */
/* CFilter::voxelfilter (include/filter.hpp:28-88, incl. its phantom-entry quirk: output row 0 is
 * a copy of input point 0).  keep_idx: capacity n+1 int32; *m [host]. */
int ghicp_voxel_filter(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, float voxel, int32_t* keep_idx, int64_t* m);
/* The stable sort behind the voxel filter, the grids and the NMS order (the reference's std::sort calls, include/filter.hpp:66 and
 * include/keypoint_detect.hpp:119-130, made deterministic: equal keys keep their input order).  Ascending on the key bits
 * [bit_begin, bit_end) of n keys of key_bytes (4 or 8) bytes each; vals_in / vals_out (u32) may both be NULL (keys only).  The inputs are
 * left untouched; out-of-place (keys_out != keys_in). */
int ghicp_sort_pairs(ghicp_ctx* ctx, int key_bytes, const void* keys_in, void* keys_out, const uint32_t* vals_in, uint32_t* vals_out, int64_t n,
                     int bit_begin, int bit_end);
/* gather rows: out[i] = xyz[idx[i]] as packed float4 (x,y,z,0). */
int ghicp_gather_points(ghicp_ctx* ctx, const float* xyz, int stride, const int32_t* idx, int64_t m, float* out_xyz4);
/* bbx_magnitude of test/ghicp_main.cpp:91-93 (CloudUtility::getCloudBound, utility.h:153-183). [host] out */
int ghicp_bbx_magnitude(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, float* bbx);
/* CloudUtility::getCloudBound (include/utility.h:153-183): min_x, min_y, min_z, max_x, max_y, max_z as doubles holding the float
 * extremes. [host] out6; n == 0 is an argument error (the reference reads cloud[0]). */
int ghicp_cloud_bounds(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, double* out6);

/* PrincipleComponentAnalysis::CalculatePcaFeaturesOfPointCloud(cloud, features, float radius)
 * (include/pca.h:133-165, 202-250).  lambda: m x 3 f32 (descending), curvature: m f64, count: m i32. */
int ghicp_pca_curvature(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, float radius, float* lambda, double* curvature,
                        int32_t* count);
/* CKeypointDetect::pruneUnstablePoints (include/keypoint_detect.hpp:132-147). cand: capacity m; *c [host]. */
int ghicp_prune(ghicp_ctx* ctx, const float* lambda, const int32_t* count, int64_t m, float ratio_max, int min_n, int32_t* cand,
                int64_t* c);
/* CKeypointDetect::nonMaximaSuppression (include/keypoint_detect.hpp:149-191).  kp: capacity c,
 * output in descending-curvature order; *k [host]. */
int ghicp_nms(ghicp_ctx* ctx, const float* xyz, int stride, const double* curvature, const int32_t* cand, int64_t c, float radius,
              int32_t* kp, int64_t* k);
/* CKeypointDetect::keypointDetectionBasedOnCurvature (include/keypoint_detect.hpp:27-51):
 * PCA -> prune -> NMS.  kp_idx: capacity m; *k [host]. */
int ghicp_keypoints(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, float radius, float ratio_max, int min_n, float nms_radius,
                    int32_t* kp_idx, int64_t* k);
/* CKeypointDetect::keypointDetectionBasedOnCurvature_adaptive (include/keypoint_detect.hpp:53-111).  The reference's
 * constants are upper = 50000, lower = 5000; ratio_used / rounds [host] (may be NULL) report the final threshold and the
 * number of extra prune + NMS rounds. */
int ghicp_keypoints_adaptive(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, float radius, float ratio_max, int min_n,
                             float nms_radius, int64_t upper, int64_t lower, int32_t* kp_idx, int64_t* k, float* ratio_used,
                             int32_t* rounds);

/* BSCEncoder::extractBinaryFeatures (include/binary_feature_extraction.hpp:603-676).
 * pattern: 49 x 2 int32 [host] (the sample_pattern.txt content, bfe:63-117).
 * feat: 4 x k x 56 bytes (variants beyond the dof's count are zero), lcs: k x 12 f32 (x,y,z axes, origin). */
int ghicp_bsc_encode(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, const int32_t* kp_idx, int64_t k, float radius, int dof,
                     const int32_t* pattern, uint8_t* feat, float* lcs);

/* FPFHfeature::compute_fpfh_feature (include/fpfh.hpp:36-58): pcl::NormalEstimation k=20 + pcl::FPFHEstimationOMP k=20
 * over the whole cloud.  normals: m x 3 f32 or NULL; hist: m x 33 f32 (pcl::FPFHSignature33 rows). */
int ghicp_fpfh(ghicp_ctx* ctx, const float* xyz, int64_t m, int stride, int k_normal, int k_feature, float* normals, float* hist);
/* FPFHfeature::keyfpfh (include/fpfh.hpp:93-115): out[i] = hist[kp_idx[i]] (k x 33). Device pointers. */
int ghicp_fpfh_keypoints(ghicp_ctx* ctx, const float* hist, const int32_t* kp_idx, int64_t k, float* out);

/* ------------------------------------------------------------------ per pair */
/* GHRegistration::calFD_BSC (src/ghicp_reg.cpp:143-200) + StereoBinaryFeature::hammingDistance
 * (src/stereo_binary_feature.cpp:87-104).  featS: V x ks x 56, featT: kt x 56, FD: ks x kt u16 row-major. */
int ghicp_fd_bsc(ghicp_ctx* ctx, const uint8_t* featS, int64_t ks, int V, const uint8_t* featT, int64_t kt, uint16_t* FD);
/* GHRegistration::calFD_FPFH (src/ghicp_reg.cpp:202-214) + FPFHfeature::compute_fpfh_distance
 * (include/fpfh.hpp:135-165).  hist: k x 33 f32, FD: ks x kt f32 row-major. */
int ghicp_fd_fpfh(ghicp_ctx* ctx, const float* histS, int64_t ks, const float* histT, int64_t kt, float* FD);

/* Km::kmsolve (src/km.cpp:40-126): w is n x n f64 row-major, match[y] = x (int32, n). */
int ghicp_km_solve(ghicp_ctx* ctx, const double* w, int64_t n, double eps, int32_t* match);

/* pcl::registration::TransformationEstimationSVD as used by GHRegistration::transformestimation
 * (src/ghicp_reg.cpp:839-866): c x 3 f64 correspondences (cast to f32 like the reference), Rt 4x4 [host]. */
int ghicp_rigid_svd(ghicp_ctx* ctx, const double* src, const double* tgt, int64_t c, double* Rt16);
/* The same solve on the host (same numerics contract, no context, no GPU): for the few-point closed-form solvers of the
 * reference's API (CRegistration::SVD_6DOF, src/common_reg.cpp:774-888).  src/tgt: c x 3 f64 host arrays. */
int ghicp_rigid_svd_host(const double* src, const double* tgt, int64_t c, double* Rt16);

/* GHRegistration::ghicp_reg (src/ghicp_reg.cpp:24-112): the whole iteration loop
 * (calED, calCD_*, findcorrespondence{NN,NNR,KM}, transformestimation, adjustweight).
 *   kpS: ks x 3 f64, kpT: kt x 3 f64 (Keypoints::setCoordinate, ghicp_reg.h:53-59); kpS is not modified.
 *   FD : ks x kt row-major; u16 for BSC, f32 for FPFH, NULL for None.
 *   Rt16 [host] final 4x4 (Source -> Target); trace [host] capacity max_iter or NULL;
 *   n_iter [host]; matchlist: max_iter x ks int32 (T index or -1) or NULL. */
int ghicp_register(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS, int64_t ks, const double* kpT, int64_t kt, const void* FD,
                   double* Rt16, ghicp_iter* trace, int32_t* n_iter, int32_t* matchlist);

/* One iteration at a time: the body of the while loop of GHRegistration::ghicp_reg (src/ghicp_reg.cpp:49-103) -- calED, calCD_*, the
 * correspondence search, transformestimation (which moves the source keypoints and accumulates Rt_tillnow), adjustweight -- as one call.
 * ghicp_loop_create copies the inputs (same meaning as ghicp_register's) into a state object; every ghicp_iterate runs exactly one
 * iteration on it and returns that iteration's record (`out`, [host]; out->converged is the loop's exit test, ghicp_reg.cpp:796-803,
 * 909-914) and, when match_row is not NULL, its correspondences (ks int32: T index or -1).  A sequence of ghicp_iterate calls until
 * out->converged reproduces ghicp_register's trace bit for bit; calling it again after convergence is an argument error.
 * ghicp_loop_result: the accumulated 4x4 (Rt_tillnow), iterations done, the converged flag, RMSE-after of the last iteration. */
typedef struct ghicp_loop ghicp_loop;
int ghicp_loop_create(ghicp_ctx* ctx, const ghicp_params* p, const double* kpS, int64_t ks, const double* kpT, int64_t kt, const void* FD,
                      ghicp_loop** out);
int ghicp_iterate(ghicp_ctx* ctx, ghicp_loop* loop, ghicp_iter* out, int32_t* match_row);
int ghicp_loop_result(const ghicp_loop* loop, double* Rt16, int32_t* n_iter, int32_t* converged, double* rmse_after);
void ghicp_loop_destroy(ghicp_loop* loop);

/* pcl::transformPointCloud(cloud, out, Rt.cast<float>()) (test/ghicp_main.cpp:153). out: n x 3 packed f32. */
int ghicp_transform_cloud(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, const double* Rt16_host, float* out_xyz);
/* The same for the raw sources of a whole batch of registered pairs in one launch: cloud i (xyz[i], n[i] rows of `stride` floats) under
 * Rt16_host[16 i .. 16 i + 15] into out[i] (n[i] x 3 packed f32).  [host] pointer tables and matrices, device clouds; bit-identical to
 * ghicp_transform_cloud cloud by cloud.  Device-pointer mode only. */
int ghicp_transform_clouds(ghicp_ctx* ctx, int32_t n_clouds, const float* const* xyz, const int64_t* n, int stride, const double* Rt16_host,
                           float* const* out_xyz);

/* ------------------------------------------------------------------ whole pair (test/ghicp_main.cpp:86-153) */
typedef struct ghicp_pair_config {
  ghicp_params reg;         /* bbx_magnitude is computed from the down-sampled source and overwritten */
  float voxel;              /* argv[6]; <= 0 skips the voxel filter */
  float neighborhood_radius;/* argv[7] */
  float ratio_max;          /* 0.65 (main:96) */
  int32_t min_neighbors;    /* 20   (main:97) */
  int32_t pattern[98];      /* BSC sample pattern */
} ghicp_pair_config;

typedef struct ghicp_pair_stats {
  int64_t n_s, n_t, m_s, m_t, k_s, k_t; /* raw, down-sampled, keypoints */
  int32_t iterations, converged;
  double Rt[16];
  float bbx_magnitude;
  float ms_voxel, ms_keypoints, ms_feature, ms_fd, ms_loop, ms_total; /* hipEvent timings */
  float pad_;
  /* src/ghicp_reg.cpp:918-924: at convergence the reference prints "Registration Succeed." iff RMSEafter < 1.5 * nonmax
   * (RMSE of the last iteration's correspondences after its transform), else "Registration Failed.".  registered_ok is that
   * verdict (0 when the loop stopped at the max_iter guard); rmse_after is the value it was taken from. */
  double rmse_after;
  int32_t registered_ok, pad2_;
} ghicp_pair_stats;

/* voxel -> keypoints -> feature -> FD -> loop for one (S,T); raw clouds are device (or host) xyz. */
int ghicp_register_pair(ghicp_ctx* ctx, const ghicp_pair_config* cfg, const float* xyzS, int64_t nS, const float* xyzT, int64_t nT,
                        int stride, ghicp_pair_stats* stats /*[host]*/, ghicp_iter* trace /*[host] or NULL*/);

/* A batch of independent pairs (BASELINE configs[3]: many fragment pairs per GPU): the per-pair front ends run
 * back to back, then one batched GH-ICP loop advances every pair concurrently (one KM solve per wave).
 * Device pointers only.  xyzS/xyzT/nS/nT: host arrays of n_pairs device pointers / sizes; stats: n_pairs [host]. */
int ghicp_register_pairs(ghicp_ctx* ctx, const ghicp_pair_config* cfg, int32_t n_pairs, const float* const* xyzS, const int64_t* nS,
                         const float* const* xyzT, const int64_t* nT, int stride, ghicp_pair_stats* stats);


/* ------------------------------------------------------------------------------------------------
 * Fine registration: CRegistration<PointT> (include/common_reg.h:26-110, src/common_reg.cpp).
 * The reference wraps PCL's IterativeClosestPoint / IterativeClosestPointWithNormals; these entry
 * points run the same loop (1-NN correspondences, optional reciprocal test and trimmed rejector,
 * closed-form solve, PCL's default convergence criteria) on the device.
 * ------------------------------------------------------------------------------------------------ */
enum { GHICP_ICP_POINT_TO_POINT = 0, GHICP_ICP_POINT_TO_PLANE = 1 };
/* pcl::registration::DefaultConvergenceCriteria::ConvergenceState */
enum { GHICP_ICP_NOT_CONVERGED = 0, GHICP_ICP_ITERATIONS = 1, GHICP_ICP_TRANSFORM = 2, GHICP_ICP_ABS_MSE = 3, GHICP_ICP_REL_MSE = 4,
       GHICP_ICP_NO_CORRESPONDENCES = 5 };

typedef struct ghicp_icp_params {
  int32_t max_iter;        /* icp_reg / ptplicp_reg argument (common_reg.cpp:51,130) */
  int32_t use_reciprocal;  /* use_reciprocal_correspondence */
  int32_t use_trimmed;     /* use_trimmed_rejector: overlap = calOverlap(S, T, thre_dis) */
  int32_t metric;          /* GHICP_ICP_POINT_TO_POINT (icp_reg) or GHICP_ICP_POINT_TO_PLANE (ptplicp_reg) */
  float thre_dis;          /* search radius of the overlap estimate */
  float min_overlap;       /* min_overlap_for_reg: below it the registration is refused (reference returns false) */
  int32_t covariance_k;    /* ptplicp_reg: k of the target normals (CalculatePointCloudWithNormal_KNN), <= 20 */
  int32_t pad_;
  double transformation_epsilon;     /* 1e-8 (common_reg.cpp:82,156) */
  double euclidean_fitness_epsilon;  /* 1e-5 (common_reg.cpp:84,158) */
} ghicp_icp_params;

typedef struct ghicp_icp_stats {
  int32_t done;        /* 0: refused because overlap < min_overlap (T untouched), 1: ran */
  int32_t iterations;
  int32_t converged;
  int32_t reason;      /* GHICP_ICP_* convergence state */
  int64_t correspondences; /* used by the last solve */
  float overlap;       /* calOverlap result when use_trimmed, else 0 */
  float pad_;
  double mse;          /* mean squared correspondence distance of the last iteration */
  double fitness;      /* getFitnessScore(): mean squared 1-NN distance of the transformed source */
} ghicp_icp_stats;

void ghicp_icp_params_default(ghicp_icp_params* p);

/* CRegistration::calOverlap (common_reg.cpp:294-317): share of cloud1 points with a cloud2 point at d^2 < thre_dis^2,
 * (0.01 + count) / n1. */
int ghicp_cal_overlap(ghicp_ctx* ctx, const float* xyz1, int64_t n1, int stride1, const float* xyz2, int64_t n2, int stride2, float thre_dis,
                      float* ratio /*[host]*/);

/* CRegistration::icp_reg (common_reg.cpp:45-107) and ptplicp_reg (122-199), selected by params->metric.
 * T16: row-major float 4x4 source->target [host]; transformed: ns x 3 (device/host per mode) or NULL. */
int ghicp_icp(ghicp_ctx* ctx, const float* xyzS, int64_t ns, int strideS, const float* xyzT, int64_t nt, int strideT,
              const ghicp_icp_params* params, float* T16 /*[host]*/, float* transformed, ghicp_icp_stats* stats /*[host]*/);

/* PrincipleComponentAnalysis::CalculateNormalVector_KNN (include/pca.h:92-109): k-NN normals + CheckNormals. */
int ghicp_knn_normals(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, int k, float* normals /*n x 3*/);

/* exact 1-NN of every query in the target (the correspondence step of the ICP loop): ties -> lower target index. */
int ghicp_nn_search(ghicp_ctx* ctx, const float* query, int64_t nq, int strideQ, const float* xyzT, int64_t nt, int strideT,
                    int32_t* idx /*nq*/, float* d2 /*nq*/);

/* CRegistration::invTransform (common_reg.cpp:357-370): R^T with the NEGATED translation, as the reference does. */
void ghicp_inv_transform(const float* T16, float* inv16);

/* CRegistration::transformcloud (common_reg.cpp:325-349): float 4x4 * (x, y, z, 1). */
int ghicp_transform_cloud_f32(ghicp_ctx* ctx, const float* xyz, int64_t n, int stride, const float* T16 /*[host]*/, float* out /*n x 3*/);

/* ------------------------------------------------------------------------------------------------
 * Per-cloud front-end cache (multi-view / all-pairs registration): down-sampling, keypoints and
 * descriptors of one cloud (test/ghicp_main.cpp:86-127) computed once and kept in HBM; a pair then costs
 * only calFD_* and the loop.  Results are identical to ghicp_register_pairs on the same clouds.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ghicp_cloud ghicp_cloud;

typedef struct ghicp_cloud_info {
  int64_t n, m, k;        /* raw points, down-sampled points, keypoints */
  int32_t variants;       /* V: BSC strings per keypoint (1, 2 or 4; binary_feature_extraction.hpp:648-660) */
  int32_t feature;        /* GHICP_FEATURE_* */
  float bbx_magnitude;    /* of the down-sampled cloud (main:91-93) */
  int32_t candidates;     /* points that passed pruneUnstablePoints (keypoint_detect.hpp:132-147) = input of the NMS; 0 when not recorded */
  int64_t feature_bytes;  /* BSC: V*k*56, FPFH: k*33*4, None: 0 */
} ghicp_cloud_info;

int ghicp_cloud_create(ghicp_ctx* ctx, const ghicp_pair_config* cfg, const float* xyz, int64_t n, int stride, ghicp_cloud** out);
/* rebuilds a handle from stored results: kp_xyz k x 3 f64, feat = V*k*56 BSC bytes (variant-major) or k x 33 f32 FPFH rows */
int ghicp_cloud_from_features(ghicp_ctx* ctx, const ghicp_pair_config* cfg, const double* kp_xyz, int64_t k, const void* feat,
                              float bbx_magnitude, ghicp_cloud** out);
/* recomputes a handle for another raw cloud with the configuration it was created with, reusing its buffers (no
 * allocation in a steady-state pipeline).  Handles are synchronised when these calls return and may then be passed to
 * ghicp_register_clouds of ANY context on the same device. */
int ghicp_cloud_recompute(ghicp_cloud* cloud, const float* xyz, int64_t n, int stride);
/* The front ends of n_clouds raw clouds [device pointers xyz[i], n[i] points of `stride` floats] into existing handles of ONE
 * front-end configuration, in one sequence of launches for the whole batch (one radix sort, one cell table, one select ... for all the
 * clouds).  Same results as ghicp_cloud_recompute(clouds[i], xyz[i], n[i], stride) for every i, bit for bit.  (The reference has no
 * counterpart: test/ghicp_main.cpp:86-127 runs the front end once per cloud.) */
int ghicp_clouds_recompute(ghicp_ctx* ctx, int32_t n_clouds, ghicp_cloud* const* clouds, const float* const* xyz /*[host] array of device pointers*/,
                           const int64_t* n /*[host]*/, int stride);
int ghicp_cloud_destroy(ghicp_cloud* cloud);
int ghicp_cloud_get_info(const ghicp_cloud* cloud, ghicp_cloud_info* info /*[host]*/);
/* any destination may be NULL: ds_xyz m x 3 f32, kp_idx k, kp_xyz k x 3 f64, feat feature_bytes */
int ghicp_cloud_download(const ghicp_cloud* cloud, float* ds_xyz, int32_t* kp_idx, double* kp_xyz, void* feat);
/* S[i] -> T[i] for n_pairs pairs of cached clouds; cfg must describe the front end the handles were built with
 * (cfg->reg.corr and the loop parameters may differ from call to call).  stats: n_pairs [host]. */
int ghicp_register_clouds(ghicp_ctx* ctx, const ghicp_pair_config* cfg, int32_t n_pairs, const ghicp_cloud* const* S,
                          const ghicp_cloud* const* T, ghicp_pair_stats* stats);

/* ------------------------------------------------------------------------------------------------
 * Pair queue: independent scan pairs sharded over the GPUs of ONE node, one process per GPU (BASELINE configs[3]).
 * The reference registers one pair per process run (test/ghicp_main.cpp:56-160) and has no counterpart; a caller of the drop-in
 * headers that holds MANY pairs shards them with these calls.  Pairs share nothing, so there is NO collective on the data path --
 * what the ranks exchange is
 *   - the pair manifest (scene ids / paths / parameters, a few KB): ghicp_pairqueue_broadcast  = one ncclBroadcast,
 *   - per step, every rank's block of result records:               ghicp_pairqueue_gather_records = one ncclAllGather,
 *   - for the dynamic split, claims on one shared counter:          ghicp_pairqueue_claim (an atomic add in host shared memory).
 * Transports: GHICP_PQ_RCCL -- an RCCL communicator over xGMI (librccl is opened when the first such queue is created; the
 * ncclUniqueId travels through the rendezvous segment); GHICP_PQ_HOST -- the same three operations through the rendezvous segment
 * alone (ranks that share a GPU, which RCCL refuses, and the world-size-2 tests on a machine without one).
 * `rendezvous`: path of a file every rank of the job can map (e.g. under /dev/shm), new for every queue; rank 0 creates it, the
 * others wait for it (up to timeout_s seconds).  All buffers of these calls are HOST memory.  Collective calls (create, broadcast,
 * gather_records, barrier, counter_reset, destroy) must be made by every rank, in the same order. */
enum { GHICP_PQ_HOST = 0, GHICP_PQ_RCCL = 1 };
enum { GHICP_PQ_RECORD_WIDTH = 19 }; /* pair id, iterations, converged, the 16 entries of the 4x4 (row-major); pair id -1: unused row */
typedef struct ghicp_pairqueue ghicp_pairqueue;
/* ctx: the rank's context (its device and stream carry the RCCL transfers); may be NULL for GHICP_PQ_HOST. */
int ghicp_pairqueue_create(ghicp_ctx* ctx, const char* rendezvous, int32_t rank, int32_t world, int32_t transport, double timeout_s,
                           ghicp_pairqueue** queue);
int ghicp_pairqueue_destroy(ghicp_pairqueue* queue);
const char* ghicp_pairqueue_last_error(const ghicp_pairqueue* queue);
int ghicp_pairqueue_info(const ghicp_pairqueue* queue, int32_t* rank, int32_t* world, int32_t* transport);
/* `bytes` bytes of root's `buf` into every rank's `buf`. */
int ghicp_pairqueue_broadcast(ghicp_pairqueue* queue, void* buf, int64_t bytes, int32_t root);
int ghicp_pairqueue_barrier(ghicp_pairqueue* queue);
/* Static split: pair p belongs to rank p mod world.  ids: capacity `cap`; *n = number of pairs of this rank (ascending). */
int ghicp_pairqueue_static_share(const ghicp_pairqueue* queue, int64_t n_pairs, int64_t* ids, int64_t cap, int64_t* n);
/* Dynamic split: the next `count` pair ids below `limit` from the queue's shared counter: [*first, *first + *n), *n == 0 when the job
 * is drained.  Not collective.  ghicp_pairqueue_counter_reset (collective) starts the next job at 0. */
int ghicp_pairqueue_claim(ghicp_pairqueue* queue, int64_t count, int64_t limit, int64_t* first, int64_t* n);
int ghicp_pairqueue_counter_reset(ghicp_pairqueue* queue);
/* Every rank's `rows` x GHICP_PQ_RECORD_WIDTH block (f64) into all[world x rows x GHICP_PQ_RECORD_WIDTH] on every rank, rank order. */
int ghicp_pairqueue_gather_records(ghicp_pairqueue* queue, const double* block, int64_t rows, double* all);
/* ghicp_pair_stats of this rank's pairs -> record rows (pair ids in `ids`); rows beyond n_mine get pair id -1. */
int ghicp_pairqueue_pack_records(const int64_t* ids, const ghicp_pair_stats* stats, int64_t n_mine, int64_t rows, double* block);
/* ghicp_register_pairs of THIS RANK'S SHARE of n_pairs pairs (static split when chunk <= 0, else claims of `chunk` pairs from the
 * shared counter), then one gather: records[n_pairs x GHICP_PQ_RECORD_WIDTH] of ALL pairs, in pair order, on every rank.  xyzS / xyzT /
 * nS / nT are indexed by GLOBAL pair id; a rank only touches the entries of the pairs it registers (device pointers, like
 * ghicp_register_pairs).  stats (n_pairs, [host], may be NULL): filled for this rank's pairs only.  Collective. */
int ghicp_pairqueue_register_pairs(ghicp_pairqueue* queue, ghicp_ctx* ctx, const ghicp_pair_config* cfg, int64_t n_pairs, const float* const* xyzS,
                                   const int64_t* nS, const float* const* xyzT, const int64_t* nT, int stride, int64_t chunk,
                                   ghicp_pair_stats* stats, double* records);

/* StereoBinaryFeature::writeFeatures / readFeatures (src/stereo_binary_feature.cpp:107-148): the reference's dump
 * format for one vector of 441-bit strings.  Host memory.  ghicp_sbf_read with feat == NULL only reports *k. */
int ghicp_sbf_write(const char* path, const uint8_t* feat /*k x 56*/, int64_t k);
int ghicp_sbf_read(const char* path, uint8_t* feat /*capacity x 56 or NULL*/, int64_t capacity, int64_t* k);

#ifdef __cplusplus
}
#endif
#endif /* GHICP_C_H_ */
