/*
 * pcr_hip.h -- C ABI of the MI355X-native scan-to-map registration library
 * (libpcr_hip.so).  Plain pointers and sizes only; no C++/torch types.
 *
 * It is the drop-in boundary for the reference's registration plugin
 *   PCR::PointCloudRegister::scan2Map(const PC_cPtr& src, const PC_cPtr& dst, pose_t& res)
 *   (reference PCR/include/PCR/PointCloudRegister.hpp:12-38)
 * and its three implementations selected by the config key frontend.pcr
 *   "loam"  -> PCR::LoamRegister   (PCR/src/LoamRegister.cpp:99-223)
 *   "ndt"   -> PCR::NdtRegister    (PCR/src/NdtRegister.cpp:21-31)
 *   "vgicp" -> PCR::VgicpRegister  (PCR/src/VgicpRegister.cpp:30-45)
 *   (factory: frontend/src/LidarOdometry.cpp:44-54).
 * INTEGRATION.md shows the reference-side adapter class that binds these entry
 * points; simpleslam_amd/host/PCR mirrors the reference classes without PCL/Eigen.
 *
 * Conventions
 *   - A point is `stride_bytes` bytes whose first 12 are float x,y,z.
 *     stride 32 = pcl::PointXYZI (reference common/types/basic.hpp:16), 16 = float4.
 *   - A pose is 16 doubles, column-major 4x4 (Eigen::Isometry3d::matrix().data(),
 *     reference common/types/basic.hpp:19); in = initial guess, out = refined.
 *   - Every call returns 0 on success, nonzero on error; pcr_last_error() then
 *     describes it.  Nothing throws across this boundary.  A handle serves one
 *     caller at a time; distinct handles are independent (own stream and buffers).
 */
#ifndef PCR_HIP_H
#define PCR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcr_handle pcr_handle;

/* Defaults (pcr_default_params) are the reference's constants; see SURVEY.md App. A. */
typedef struct pcr_params {
    uint32_t struct_size;      /* sizeof(pcr_params), set by pcr_default_params */
    int32_t device;            /* HIP device ordinal; -1 = current device */

    /* LOAM -- PCR/include/PCR/LoamRegister.hpp:30-40 */
    int32_t loam_iters;        /* 8   iteration cap (LoamRegister.hpp:40) */
    int32_t loam_early_exit;   /* 1   stop when the step is small (LoamRegister.cpp:202-206) */
    double loam_knn_max_sq;    /* 1.0 gate on the SQUARED 5th-neighbour distance (LoamRegister.cpp:59) */
    double loam_plane_thresh;  /* 0.2 plane validity, times |x| (LoamRegister.cpp:39) */
    double loam_point_thresh;  /* 0.1 weight gate (LoamRegister.cpp:151) */
    double loam_pos_conv;      /* 5e-3 (LoamRegister.hpp:37) */
    double loam_rot_conv;      /* 5e-3 (LoamRegister.hpp:38) */

    /* NDT -- PCR/src/NdtRegister.cpp:12-13, third_parties/pclomp/src/ndt_omp_impl.hpp:50-51,71-72 */
    double ndt_resolution;     /* 1.0 */
    double ndt_step_size;      /* 0.1 */
    double ndt_outlier_ratio;  /* 0.55 */
    double ndt_trans_eps;      /* 0.1 */
    int32_t ndt_max_iters;     /* 35 */
    int32_t ndt_min_points;    /* 6 points per voxel (pclomp/voxel_grid_covariance_omp.h:210) */

    /* VGICP -- PCR/src/VgicpRegister.cpp:13, third_parties/pclomp/src/fast_vgicp_impl.hpp:22-24,
     * fast_gicp_impl.hpp:16-20, lsq_registration_impl.hpp:11-18 */
    double vgicp_resolution;   /* 1.0 */
    int32_t vgicp_k_corr;      /* 20 neighbours for the covariances */
    int32_t vgicp_max_iters;   /* 64 */
    int32_t vgicp_lm_inner;    /* 10 */
    double vgicp_rot_eps;      /* 2e-3 */
    double vgicp_trans_eps;    /* 5e-4 */
    double vgicp_lm_init_scale;/* 1e-9 */

    int32_t record_trace;      /* 1: keep per-iteration normal equations for pcr_get_trace */

    /* ---- switches (all 0 by default; none of them changes a result except where noted).  The library reads NO environment
     * variable: what used to be PCR_* variables and pcr_params.reserved[] are these fields, or exist only in a development build
     * (make DEV=1, csrc/pcr_internal.h: dev_env). ---- */
    int32_t index_no_hints;    /* 1: every target index is built from scratch -- fresh bounding box, no tile layout taken over from the
                                *    previous build of this handle (three build launches instead of two; no state crosses calls) */
    int32_t ndt_evaluate_repeats; /* 1: NDT: every request of the line search becomes an evaluation pass, also one at the point just
                                *    evaluated (what the reference does; by default such a request is answered from the sums already
                                *    held -- same numbers, csrc/ndt_opt.h) */
    int32_t loam_disable_cache;/* 1: LOAM: no temporal-coherence neighbour cache, every iteration searches (same result bit for bit) */
    int32_t record_timeline;   /* 1: LOAM: record the in-kernel timeline of every launch (pcr_get_timeline) */
    int32_t loam_coresident;   /* 1: LOAM: the two-waves-per-SIMD variant of the iterate kernel -- ~3 % slower for one handle, ~25 % more
                                *    scans/s when several handles register scans concurrently on one GPU (their blocks share the CUs) */
    int32_t loam_clamp_margin_mm; /* != 0: LOAM: first margin, in millimetres, of the region a target too sparse for the dense index is
                                *    cut to around the scan (default 10 m; it grows whenever a query reaches a cut face).  Negative values cut
                                *    into the scan's own box: a test hook that makes the widening path reachable with ordinary clouds */
    int32_t full_target;       /* 1: NDT, VGICP: pcr_scan2map prepares the WHOLE target (every covariance, every voxel Gaussian), as the
                                *    reference does, instead of only the region the scan can reach (same result: a call whose pose leaves
                                *    the region is repeated on the whole target anyway) */
    int32_t host_optimiser;    /* 1: NDT, VGICP: drive the optimiser from the host (one round trip per evaluation pass) instead of on the
                                *    device; the path handles sharded over a host-supplied collective always take.  Same state machines
                                *    (csrc/ndt_opt.h, csrc/vgicp_opt.h), same result to rounding */
    int32_t host_copy_xyz;     /* 1: of HOST clouds whose records are wider than 16 bytes (pcl::PointXYZI: 32) only the first 16 bytes of every record are
                                *    uploaded (a pitched hipMemcpy2DAsync into a staging area of the same stride; registration reads x, y, z and nothing
                                *    else).  Off by default: measured on MI355X / ROCm 7.2 the pitched copy of a 1 M-point map takes 3.5 ms where the
                                *    verbatim copy of twice the bytes takes 0.6 ms (profiles/r04_notes.md) */
} pcr_params;

/* Per-call device timings, from HIP events on the handle's stream. */
typedef struct pcr_stats {
    double total_ms;        /* whole call on the device timeline */
    double index_ms;        /* target index build (0 when the cached index was used) */
    double solve_ms;        /* all optimisation launches */
    double kernel_ms;       /* sum over the dominant kernel's launches (only with pcr_set_profile(h,2)) */
    int32_t kernel_launches;/* launches summed in kernel_ms */
    int32_t iterations;     /* linearisations performed */
    int64_t n_src, n_dst;
    int32_t attempts;       /* LOAM: passes over the iteration loop (> 1: the cell table grew, or a cut index was widened);
                             * NDT: evaluation passes the device loop launched -- fewer than kernel_launches (= the evaluations the
                             * reference makes) when repeated line-search evaluations were answered without a pass */
    int32_t target_builds;  /* pcr_scan2map_submap: times this handle has (re)built its target structures (one per sub-map generation) */
    int32_t region_repeats; /* NDT, VGICP pcr_scan2map: calls of this handle so far whose pose left the region the target had been prepared for and
                             * that were therefore repeated on the whole target (pcr_params.full_target) */
    int32_t region_index;   /* NDT pcr_scan2map: 1 when the last call's target index itself held only the points of the scan's region (possible once
                             * an earlier call of the handle has left its lattice and tile layout as hints), 0 when it held the whole cloud */
    /* Filled by a call made under pcr_set_profile(h, 2) only (events at the kernels' own begin and end, counters in the kernels): what the
     * last NDT / VGICP pcr_scan2map really processed, so that a roofline figure can be computed from the work done.
     *   VGICP: kernel_ms = the target's covariance kernel (vgicp_cov_kernel<false>), aux_kernel_ms = the three kernels of the scan's own
     *          covariance search (csrc/cov_search.hip), region_points / region_voxels = target points whose covariance was computed / voxels built
     *   NDT:   kernel_ms / kernel_launches = the evaluation launches of the device loop (ndt_pass_pro_kernel), pairs_grad / pairs_hess =
     *          (point, voxel) pairs evaluated by the gradient-only passes / by the passes that also accumulate the float Hessian */
    double aux_kernel_ms;
    int64_t region_points, region_voxels, pairs_grad, pairs_hess;
    /* the handle's last build of its target index: did it reuse the previous target's bounding box (header), and did it place its points by the
     * previous build's tile layout (both are hints that are checked on the device; a hint that failed shows as 0: the build was redone without it) */
    int32_t index_box_hint, index_layout_hint;
} pcr_stats;

void pcr_default_params(pcr_params* p);

/* method = "loam" | "ndt" | "vgicp" (frontend.pcr).  NULL on failure; pcr_last_error(NULL)
 * then holds the reason (unknown method: the reference factory throws,
 * LidarOdometry.cpp:50-54).  p == NULL uses the defaults. */
pcr_handle* pcr_create(const char* method, const pcr_params* p);
void pcr_destroy(pcr_handle* h);
const char* pcr_last_error(const pcr_handle* h);

/* scan2Map with HOST buffers (what PCR::PointCloudRegister::scan2Map hands over:
 * src->points.data(), dst->points.data()).  The target index is rebuilt on every
 * call, as the reference does (LoamRegister.cpp:110).  *converged = isConverge. */
int pcr_scan2map(pcr_handle* h, const void* src, size_t n_src, const void* dst, size_t n_dst,
                 size_t stride_bytes, double pose_inout[16], int* converged);

/* Page-lock a host buffer that will be handed to pcr_scan2map / pcr_set_target / pcr_align (and the other entry points taking host
 * clouds) repeatedly -- e.g. the sub-map's point array between two MapManager updates: the upload then reads the caller's pages
 * directly (~55 GB/s over PCIe 5 x16) instead of going through the runtime's staging of pageable memory.  Explicit on purpose:
 * the registration holds for exactly the range [ptr, ptr + bytes) until pcr_host_unpin(ptr), which MUST precede freeing or
 * reallocating the buffer -- the library never keys anything on a pointer it merely remembers.  Process-wide, any
 * thread; the CONTENT is still copied on every call.  Errors: nonzero, message in pcr_last_error(NULL). */
int pcr_host_pin(const void* ptr, size_t bytes);
int pcr_host_unpin(const void* ptr);

/* Same with DEVICE-resident buffers (HBM pointers valid on the handle's device). */
int pcr_scan2map_device(pcr_handle* h, const void* d_src, size_t n_src, const void* d_dst, size_t n_dst,
                        size_t stride_bytes, double pose_inout[16], int* converged);

/* Static-map localisation (reference test/loc.cpp: one PCD map, many scans): build
 * the target index once, then align scans against it.  on_device != 0 marks the
 * buffers as HBM pointers.  The library copies what it keeps. */
int pcr_set_target(pcr_handle* h, const void* dst, size_t n_dst, size_t stride_bytes, int on_device);
int pcr_align(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device,
              double pose_inout[16], int* converged);
/* Drop the cached target (the reference caches by pointer identity and goes stale,
 * SURVEY.md F10; this library never keys on the pointer). */
int pcr_invalidate_target(pcr_handle* h);

/* PointCloudRegister::getFitnessScore (PointCloudRegister.hpp:34; VgicpRegister.cpp:42-45):
 * mean squared 1-NN distance of the last aligned source.  Negative when unavailable.
 * As in the reference the score is not part of scan2Map: the handle keeps a copy of the last aligned scan and its final pose, and
 * this call evaluates the score (once; later calls return the cached value) against the target the handle holds at that moment --
 * what PCL's getFitnessScore() does with input_, final_transformation_ and the current target tree.  1.797e308 (DBL_MAX) when no
 * point has a neighbour or no target is prepared, like PCL.  (A handle of a sharded target evaluates it with the alignment instead:
 * every rank takes part in the sum.) */
double pcr_fitness(pcr_handle* h);

/* ---- introspection used by tests and bench.py ---- */

/* One LOAM linearisation at `pose` against the current target (pcr_set_target):
 * JtJ (36, row-major), JtE (6), accepted count.  Optional per-point outputs (host,
 * may be NULL): status[n] (0 accepted, 1 k-NN gate, 2 plane gate, 3 weight gate),
 * rows[n*7] (s*[n ; p x n], s*d), nn[n*5] (original target indices, ascending distance). */
int pcr_loam_linearize(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device,
                       const double pose[16], double JtJ[36], double JtE[6], int64_t* n_accepted,
                       int8_t* status, double* rows, int32_t* nn);

/* Per-iteration trace of the last LOAM call (params.record_trace = 1):
 * for it < *n_iters: JtJ[it*36..], JtE[it*6..], n[it], x[it*6..]. Arrays sized for loam_iters. */
int pcr_get_trace(pcr_handle* h, int32_t* n_iters, double* JtJ, double* JtE, int64_t* n, double* x);
/* Per linearisation of the last traced LOAM call: scan points whose neighbours came from the temporal
 * neighbour cache (proven exact) and points that needed a full grid search. */
int pcr_get_trace_counts(pcr_handle* h, int64_t* cache_hits, int64_t* searches);

/* VGICP introspection.  Per-point covariances as fast_gicp::FastGICP::calculate_covariances forms
 * them (fast_gicp_impl.hpp:241-297; 20-NN, PLANE regularisation): cov_out[n*6] = xx xy xz yy yz zz. */
int pcr_vgicp_covariances(pcr_handle* h, const void* pts, size_t n, size_t stride_bytes, int on_device, double* cov_out);
/* The neighbour lists behind the covariances of the LAST pcr_vgicp_covariances call on a scan-sized cloud (n <= 300 000):
 * nbr_out[n*20] = original indices of each point's 20 nearest neighbours (FLANN's float distances, ties on the lower index; the
 * point itself first), 0xffffffff where the cloud holds fewer; *queued_out = the queries the lane-per-query search handed to the
 * wave-per-query one (csrc/cov_search.hip).  fast_gicp_impl.hpp:250-253 (kdtree.nearestKSearch). */
int pcr_vgicp_neighbours(pcr_handle* h, size_t n, uint32_t* nbr_out, uint32_t* queued_out);
/* One FastVGICP::linearize (fast_vgicp_impl.hpp:119-180) at `pose` against the current target
 * (pcr_set_target): H (36, row-major, twist = [rotation; translation]), b (6), sum of errors,
 * number of source points with a voxel correspondence. */
int pcr_vgicp_linearize(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device,
                        const double pose[16], double H[36], double b[6], double* error, int64_t* n_corr);

/* pcl::VoxelGrid<PointXYZI>::filter (leaf, leaf, leaf) -- the step before the path: every scan at
 * frontend/src/LidarOdometry.cpp:36,170-171, every rebuilt sub-map at frontend/src/MapManager.cpp:78,192 through
 * common/pcp/pcp.hpp:14-28.  One centroid per occupied voxel of PCL's lattice (idx = floor(p / leaf) - min_b, float
 * arithmetic as in PCL), all fields averaged, output in ascending voxel index.  Points keep the input stride; with the
 * 32-byte pcl::PointXYZI layout data[3] = 1 and the intensity (float 4) is averaged, with a 16-byte layout float 3 is.
 * Non-finite points are skipped.  *n_out = number of voxels (also when out_capacity was too small, so that the caller
 * can retry).  A leaf so small that PCL's integer voxel index would overflow returns the input unfiltered, like PCL.
 * Works on any handle (the method is irrelevant). */
int pcr_voxel_filter(pcr_handle* h, const void* pts, size_t n, size_t stride_bytes, int on_device, double leaf, void* out,
                     size_t out_capacity, int out_on_device, size_t* n_out);

/* The same in two halves, device memory in and out: _begin queues the filter on the handle's stream and returns, _end waits for it and returns the voxel
 * count (the filter's one synchronisation).  For a caller that has the NEXT scan while this one is being registered -- replaying a recording, or a
 * front end that lags its sensor: filter it on a second handle (a stream of its own) meanwhile.  out_capacity >= n; input and output stay untouched until _end.
 * One filter per handle at a time (pcr_voxel_filter on that handle in between is refused). */
int pcr_voxel_filter_begin(pcr_handle* h, const void* d_pts, size_t n, size_t stride_bytes, double leaf, void* d_out, size_t out_capacity);
int pcr_voxel_filter_end(pcr_handle* h, size_t* n_out);

/* ---- the producer of `dst`: MapManager's key-frame store and sub-map assembly, on the device ----
 * MapManager::updateMap (frontend/src/MapManager.cpp:151-201): key frames whose position lies within `radius`
 * (mSurroundingKeyframeSearchRadius = 8 m, MapManager.hpp:68; squared L2 in double, strict '<' like
 * KeyFramesKdtree::radiusSearch, kfs_adaptor.hpp:57-75) are transformed by their pose cast to float
 * (pcp::transformPointCloud, common/pcp/pcp.hpp:38-62), concatenated and voxel-filtered with grid_size
 * (pcp::voxelDownSample, pcp.hpp:14-20).  Key frames are copied into HBM when added; the assembled sub-map stays in
 * HBM and is handed to pcr_scan2map_device / pcr_set_target as `dst` (pcr_map_submap: pointer valid until the next
 * pcr_map_update or pcr_map_destroy).  pcr_map_submap_indices = mSubmapIdx (ascending key-frame index). */
typedef struct pcr_map pcr_map;
pcr_map* pcr_map_create(int device);                    /* device ordinal, -1 = current; NULL on failure (pcr_map_last_error(NULL)) */
void pcr_map_destroy(pcr_map* m);
const char* pcr_map_last_error(const pcr_map* m);
/* KeyFrame{pc, pose} (common/types/basic.hpp:33-40); pose = 16 doubles column-major.  Index of the new key frame = count - 1. */
int pcr_map_add_keyframe(pcr_map* m, const void* pts, size_t n, size_t stride_bytes, int on_device, const double pose[16]);
int pcr_map_keyframes(const pcr_map* m, size_t* n_keyframes);
/* Forget every key frame and the sub-map (a new session: MapManager::reset), keep the device memory the store has grown to; starts a new generation. */
int pcr_map_clear(pcr_map* m);
int pcr_map_update(pcr_map* m, const double position[3], double radius, double grid_size, size_t* n_submap);
/* The same in two halves, for a caller that has something else to do meanwhile -- the reference assembles its sub-map on a map thread of its own
 * (frontend/src/MapManager.cpp:109-119 notifies it and LidarOdometry goes on with the next scan): pcr_map_update_begin selects the key frames, starts a
 * new generation and QUEUES the assembly on the map's stream; pcr_map_wait collects it (the update's one synchronisation; errors of the assembly are
 * reported here).  Every call that needs the sub-map (pcr_map_submap, pcr_scan2map_submap) or changes the store (pcr_map_add_keyframe, another update)
 * collects a queued assembly first, so the result never depends on when the wait happens: pcr_map_update = begin + wait.
 * (A pcr_map, like a pcr_handle, is used by one thread at a time: the overlap is between the device's streams, not between host threads.) */
int pcr_map_update_begin(pcr_map* m, const double position[3], double radius, double grid_size);
int pcr_map_wait(pcr_map* m, size_t* n_submap);
/* LoopClosureManager::loopFindNearKeyframes (backend/src/LoopClosureManager.cpp:40-60): key frames key - search_num .. key + search_num
 * (clipped to the store), transformed, concatenated, voxel-filtered: the target of the loop-closure registration (:96-99). */
int pcr_map_update_window(pcr_map* m, long long key, int search_num, double grid_size, size_t* n_submap);
const void* pcr_map_submap(const pcr_map* m, size_t* n, size_t* stride_bytes);
int pcr_map_submap_indices(const pcr_map* m, int64_t* idx, size_t capacity, size_t* n);
/* Identity of the store and of the sub-map it currently holds: every pcr_map_update / pcr_map_update_window starts a new
 * generation.  This is the explicit version of what the reference's registrars approximate by comparing cloud POINTERS
 * (fast_gicp_impl.hpp:83-90): a structure built for generation g is valid exactly while the map reports g. */
int pcr_map_generation(const pcr_map* m, uint64_t* id, uint64_t* generation);
/* scan2Map against the sub-map `m` currently holds (stride = the map's).  The handle keeps the target structures it builds
 * -- LOAM's grid index, NDT's voxel Gaussians, VGICP's covariances and voxel map -- together with (id, generation) of the
 * sub-map they were built from and rebuilds them only when the map has moved on: LidarOdometry registers several scans
 * against one sub-map between two MapManager updates (frontend/src/LidarOdometry.cpp:184, MapManager.cpp:151-201).  Same
 * result as pcr_scan2map_device on pcr_map_submap(m), bit for bit. */
int pcr_scan2map_submap(pcr_handle* h, const void* src, size_t n_src, int src_on_device, const pcr_map* m,
                        double pose_inout[16], int* converged);

/* ---- the loop-closure descriptor: backend/src/ScanContext.cpp (20 rings x 60 sectors over 80 m) ----
 * pcr_sc_add = addContext (:56-66): the polar binning of the (down-sampled, lidar-frame) scan runs on the device, the ring
 * and sector keys follow on the host.  pcr_sc_query = query (:231-279): the 10 nearest ring keys among the contexts
 * older than num_exclude_recent (a snapshot refreshed every build_tree_gap contexts, as the reference's lazily rebuilt
 * tree), each compared by distanceBtnScanContext (:116-150); *match = -1 when the best distance exceeds dist_thres. */
typedef struct pcr_sc pcr_sc;
typedef struct pcr_sc_params {
    double lidar_height;          /* 2.0  config/params.json tf.lidar_height */
    int32_t num_exclude_recent;   /* 40   backend.context.scancontext.numExcludeRecent */
    int32_t build_tree_gap;       /* 10 */
    int32_t num_candidates;       /* 10   numCandidatesFromTree */
    int32_t pad;
    double search_ratio;          /* 0.1 */
    double dist_thres;            /* 0.4  scDistThres */
} pcr_sc_params;
void pcr_sc_default_params(pcr_sc_params* p);
pcr_sc* pcr_sc_create(int device, const pcr_sc_params* p);
void pcr_sc_destroy(pcr_sc* sc);
const char* pcr_sc_last_error(const pcr_sc* sc);
int pcr_sc_size(const pcr_sc* sc, size_t* n);
int pcr_sc_add(pcr_sc* sc, const void* pts, size_t n, size_t stride_bytes, int on_device);
int pcr_sc_descriptor(const pcr_sc* sc, size_t id, double* desc_row_major_20x60, double* ring_key_20, double* sector_key_60);
int pcr_sc_distance(const pcr_sc* sc, size_t id1, size_t id2, double* dist, int* shift);
int pcr_sc_query(pcr_sc* sc, long long id, long long* match, float* yaw_rad, double* min_dist);

/* The NDT optimiser on its own (host only, no GPU): pclomp's computeTransformation + computeStepLengthMT (ndt_omp_impl.hpp:81-171,
 * 735-932) as the state machine that pcr_scan2map runs on the device (csrc/ndt_opt.h), driven from outside -- the caller evaluates what
 * it asks for and feeds the 43 sums back.  An introspection entry point like pcr_ndt_derivatives: the CPU test suite drives it with the
 * oracle's derivatives and must arrive where the oracle's own loop arrives.
 *   request: kind 0 = score + gradient + Hessian at p6 (computeDerivatives), 1 = score + gradient only, 2 = computeHessian (double) at
 *            the same p6 as the previous request, 3 = finished.  (Score + gradient are never asked for twice running at one p6: the
 *            reference's clamped trial steps do repeat, and the state machine feeds itself the sums it already has.)  pose16 (optional) = the float transform of p6, column-major.
 *   feed:    sums = score, gradient[6], Hessian[36] (row-major; ignored entries may be anything finite).
 *   result:  the pose computeTransformation would return (Matrix4f values), converged flag, iterations. */
typedef struct pcr_ndt_opt pcr_ndt_opt;
pcr_ndt_opt* pcr_ndt_opt_create(const double pose_guess[16], double step_size, double trans_eps, int max_iters);
void pcr_ndt_opt_destroy(pcr_ndt_opt* o);
int pcr_ndt_opt_request(const pcr_ndt_opt* o, int* kind, double p6[6], double pose16[16]);
int pcr_ndt_opt_feed(pcr_ndt_opt* o, const double sums[43]);
int pcr_ndt_opt_result(const pcr_ndt_opt* o, double pose16[16], int* converged, int* iterations, int* done);
/* evaluations = computeDerivatives calls of the reference's loop so far, hessians = its computeHessian calls, replayed = how many of the
 * evaluations were requests for score + gradient at the point they had just been evaluated at, answered without asking (ndt_opt.h) */
int pcr_ndt_opt_counts(const pcr_ndt_opt* o, int* evaluations, int* hessians, int* replayed);

/* The VGICP optimiser on its own (host only, no GPU): fast_gicp's LsqRegistration::computeTransformation + step_lm
 * (lsq_registration_impl.hpp:53-79, 125-171) as the state machine that pcr_scan2map runs on the device (csrc/vgicp_opt.h), driven
 * from outside like pcr_ndt_opt_*: the caller evaluates what it asks for and feeds the 29 sums back.
 *   request: kind 0 = linearize at pose_eval (fast_vgicp_impl.hpp:119-180); 1 = an LM trial: compute_error (:183-204) at pose_eval on
 *            the correspondences of the linearisation at pose_lin, AND the linearisation at pose_eval; 2 = finished.  Poses column-major.
 *   feed:    sums = H upper triangle row by row (21), b (6), error of the linearisation at pose_eval (1), the trial's error (1).
 *   result:  the pose computeTransformation would return (before the Matrix4f cast of VgicpRegister.cpp:37), converged, outer iterations. */
typedef struct pcr_vgicp_opt pcr_vgicp_opt;
pcr_vgicp_opt* pcr_vgicp_opt_create(const double pose_guess[16], int max_iters, int lm_inner, double lm_init_scale, double rot_eps, double trans_eps);
void pcr_vgicp_opt_destroy(pcr_vgicp_opt* o);
int pcr_vgicp_opt_request(const pcr_vgicp_opt* o, int* kind, double pose_eval[16], double pose_lin[16]);
int pcr_vgicp_opt_feed(pcr_vgicp_opt* o, const double sums[29]);
int pcr_vgicp_opt_result(const pcr_vgicp_opt* o, double pose16[16], int* converged, int* outer_iterations, int* done);

/* Profiling aid: with pcr_params.record_timeline = 1 thread 0 of every linearisation block records seven
 * s_memrealtime stamps (100 MHz ticks): entry, prologue done, misses posted, search done, plane+cache done,
 * accumulation done, partial sums stored (+ the fold inside the prologue and three stamps of the dense search).  out receives
 * [launches][blocks][16] u64; call with out = NULL to size it. */
int pcr_get_timeline(pcr_handle* h, uint64_t* out, size_t capacity, int* launches, int* blocks);

/* NDT introspection: one computeDerivatives pass (ndt_omp_impl.hpp:180-285) at the parameter vector
 * p = [tx ty tz, roll pitch yaw] (Translation * Rx * Ry * Rz, ndt_omp_impl.hpp:146-149) against the current
 * target (pcr_set_target): score, gradient (6), Hessian (36, row-major) and, when hess_d != NULL, the
 * double-precision Hessian of computeHessian (ndt_omp_impl.hpp:541-645) at the same pose. */
int pcr_ndt_derivatives(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device,
                        const double p[6], double* score, double grad[6], double hess[36], double* hess_d);

int pcr_get_stats(pcr_handle* h, pcr_stats* out);
/* 0: no timing events; 1: phase events (default); 2: also an event pair around every
 * launch of the dominant kernel (adds host work; for roofline measurement only). */
int pcr_set_profile(pcr_handle* h, int level);
/* Use an existing HIP stream (hipStream_t) instead of the handle's own. */
int pcr_set_stream(pcr_handle* h, void* hip_stream);

/* ---- multi-GPU (one process per GPU; map tiles + 1-cell halo per rank) ---- */

/* Restrict the handle to queries whose transformed position lies in [lo, hi) (metres,
 * map frame): each scan point is then processed by exactly one rank.  lo > hi clears it. */
int pcr_set_query_tile(pcr_handle* h, const double lo[3], const double hi[3]);
/* RCCL communicator for the per-linearisation all-reduce of the normal equations
 * (28 doubles).  unique_id = 128 bytes from pcr_comm_unique_id on rank 0, shared by the
 * caller (e.g. torch.distributed broadcast).  librccl is dlopen'ed here, never before. */
int pcr_comm_unique_id(void* out128);
int pcr_comm_init(pcr_handle* h, const void* unique_id128, int rank, int nranks);
/* Who takes part in this handle's exchange: *transport = 0 none (unsharded), 1 RCCL, 2 the caller's collective, 3 the peer exchange.  With RCCL
 * rank and nranks are read back from the communicator (ncclCommUserRank / ncclCommCount), not from the arguments it was
 * created with -- bench.py reports them as "rccl_ranks". */
int pcr_comm_info(const pcr_handle* h, int* rank, int* nranks, int* transport);
/* The same exchange through a caller-supplied collective (MPI, gloo, threads of one process ...), used when no RCCL
 * communicator is set: fn must combine `count` doubles IN PLACE over all ranks -- op 0 = sum, 1 = max -- return 0 on
 * success, and leave bitwise-identical results on every rank (the ranks then solve redundantly and must agree).  It is
 * called on the thread that called the registration entry point, between device launches (one host round trip per
 * linearisation: the fallback and test path; RCCL keeps the exchange on the device).  fn == NULL clears it. */
typedef int (*pcr_allreduce_fn)(double* inout, size_t count, int op, void* user);
int pcr_comm_init_host(pcr_handle* h, pcr_allreduce_fn fn, void* user, int rank, int nranks);
/* Peer exchange (PROTOTYPE, loam handles): the ranks' 32 sums without a collective library -- every rank pushes its values into a slot of
 * every peer's receive buffer (stores over xGMI into memory mapped through hipIpc), waits for its own buffer to fill and folds the slots in
 * rank order; one small launch per linearisation instead of a reduce kernel + ncclAllReduce (csrc/loam.hip: peer_exchange_block).
 *   1. every rank: pcr_comm_peer_export(h, handle)  -> 64 bytes (a hipIpcMemHandle_t) naming its receive buffer
 *   2. the caller shares the handles (any channel: MPI, gloo, a file), in rank order
 *   3. every rank: pcr_comm_init_peer(h, handles (nranks x 64 bytes), rank, nranks)      (nranks <= 8: one node)
 * A rank that does not arrive within 2 s fails the exchange on every rank that waited for it (no device hang).  pcr_comm_info: transport 3. */
int pcr_comm_peer_export(pcr_handle* h, void* ipc_handle64);
int pcr_comm_init_peer(pcr_handle* h, const void* ipc_handles, int rank, int nranks);
/* Tile of a sharded target, all three methods: the handle processes the scan points whose transformed position lies in
 * [lo, hi) (as pcr_set_query_tile) and is promised that the target cloud it is given holds EVERY map point inside
 * [lo - halo, hi + halo) (faces at +-1e30 are open).  What the halo must cover:
 *   loam   the k-NN gate radius (sqrt(loam_knn_max_sq) = 1 m, LoamRegister.cpp:59);
 *   ndt    lo, hi multiples of ndt_resolution and halo >= one voxel (two unless the resolution is a power of two): the
 *          DIRECT7 neighbourhood (ndt_omp_impl.hpp:242) reads the six face voxels, which must be complete;
 *   vgicp  lo, hi on the voxel lattice ((k + 0.5) * vgicp_resolution, fast_vgicp_voxel.hpp:158-160) and a halo that holds
 *          the 20 nearest neighbours of every point of the tile (fast_gicp_impl.hpp:253): checked on the device for every
 *          point within one voxel of the tile -- a neighbourhood that reaches past the halo fails the call on all ranks.
 * Misaligned bounds are refused.  lo[0] > hi[0] clears the tile. */
int pcr_set_shard(pcr_handle* h, const double lo[3], const double hi[3], double halo);

/* Replace the parameters of a live handle (host-side state; the prepared target is dropped when a parameter that shaped
 * it changed).  VgicpRegister::initForLC() (PCR/src/VgicpRegister.cpp:21-28, called on a constructed object at
 * backend/src/LoopClosureManager.cpp:21-22) = pcr_set_params with vgicp_max_iters 100, vgicp_trans_eps 1e-6.  device and
 * struct_size must match the handle's. */
int pcr_set_params(pcr_handle* h, const pcr_params* p);
int pcr_get_params(const pcr_handle* h, pcr_params* out);

/* The fitness score of the reference's test/align.cpp:29-61: the source transformed by `pose` (float, as
 * pcl::transformPointCloud), 1-NN in the handle's current target, mean of the squared distances that are <= max_sq
 * (align.cpp uses 1.0); *n_in = points counted.  score = -1 when none is (align.cpp:56-59).  Any method's handle with a target.
 * An NDT handle whose last pcr_scan2map indexed the scan's region only (pcr_stats.region_index) indexes the target again, in full, when
 * that target came in as a HOST buffer (it still lies in the handle's staging copy); a DEVICE target is the caller's and may be gone: the
 * call fails and says so (pcr_set_target, or pcr_params.full_target = 1, avoid it). */
int pcr_fitness_gated(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device, const double pose[16],
                      double max_sq, double* score, int64_t* n_in);

#ifdef __cplusplus
}
#endif
#endif /* PCR_HIP_H */
