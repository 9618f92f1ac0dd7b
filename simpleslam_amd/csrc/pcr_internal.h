// pcr_internal.h -- shared host/device declarations of libpcr_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include <stdlib.h>

#include "../../include/pcr_hip.h"

// Development switches (sweeps, A/B runs, debug prints: PCR_TILE_SHIFT, PCR_BIN_PER, PCR_INDEX_ATOMIC, PCR_INDEX_NO_LAYOUT,
// PCR_INDEX_DEBUG, PCR_COV_LEVELS, PCR_COV_RATIO, PCR_COV_NO_AHEAD, PCR_NDT_TWO_LAUNCHES, PCR_NDT_TICKS, PCR_NDT_NO_REPLAY, PCR_ABLATE)
// exist only in a library built with `make DEV=1` (-DPCR_DEV_SWITCHES): the product reads no environment variable, what a caller
// may select is a named field of pcr_params.
#ifdef PCR_DEV_SWITCHES
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

namespace pcr {

// ---------------------------------------------------------------------------
// Uniform-grid spatial index over the target cloud ("voxel tiles").
//
// Replaces the reference's per-call nanoflann kd-tree (LoamRegister.cpp:110;
// nanoflann.hpp:1542-1564).  Cell edge = the k-NN gate radius (1 m for the
// reference constants), so every neighbour that can pass the gate
// lies in the 3x3x3 block around the query's cell.  Points are counting-sorted by
// linear cell key (x fastest), so the three x-adjacent cells of one (y,z) row
// are ONE contiguous run of float4s: a query reads 9 runs.
// ---------------------------------------------------------------------------
struct GridHeader {
    double origin[3];        // lower corner of cell (0,0,0) = org * cell
    double org[3];           // integer-valued: cell (i,j,k) spans [(org+i)*cell, (org+i+1)*cell)
    double cell;             // edge length (a power of two >= gate radius: x/cell is exact)
    double inv_cell;
    int32_t dims[3];         // cells per axis, including 2 pad cells on every side
    uint32_t n_points;       // points kept (finite coordinates)
    uint64_t n_cells;        // dims[0]*dims[1]*dims[2]
    int32_t overflow;        // n_cells + 1 exceeds the allocated cell table
    int32_t empty;           // no finite point
    double shift;            // lattice offset in cells: cell index = floor(x / cell - shift) - org (0.5 for the VGICP voxel lattice)
    // pcl::VoxelGrid lattice (voxel_filter.hip): cell index = (int)(floorf(x * inv_leaf_f) - (float)min_b), all in float,
    // dims = max_b - min_b + 1 without pad cells.  pcl_mode = 0 everywhere else.
    int32_t pcl_mode;
    float inv_leaf_f;
    int32_t min_b[3];
    int32_t too_fine;        // pcl_mode: more than INT_MAX voxels ("Leaf size is too small", voxel_grid.hpp)
    float sum_sq;            // sum over the cells of count^2 (written by the scan of the atomic build path): sum_sq / points =
                             // occupancy of the cell a point lives in, averaged over the points -- the density estimate behind the
                             // choice of a search cell.  The tiled build path accumulates the same sum exactly in sum_sq_u.
    int32_t clamped;         // the box was cut to a region of interest (ClampBox): points outside it are not indexed
    int32_t cut_mask;        // clamped: bit d = the lower face of axis d was cut (target points lie beyond it), bit 3 + d = the upper
    unsigned long long sum_sq_u;
    int32_t stale;           // the header was reused from the previous build (bounding-box hint) and a point of the new cloud lies
    int32_t pad2_;           // outside its box: the index is incomplete, the caller rebuilds with a fresh box
};
__host__ __device__ inline double grid_sum_sq(const GridHeader& h) { return h.sum_sq_u ? (double)h.sum_sq_u : (double)h.sum_sq; }

// Region of interest for an index whose full bounding box cannot be tabulated (a far outlier in the cloud): see capi.hip
struct ClampBox { double lo[3], hi[3]; int32_t use, pad_; };

struct HeaderTwin { GridHeader* hdr; GridHeader* mirror; uint64_t capacity; double cell; };      // grid_bbox_header_kernel: a second header from the same box

struct GridView {            // what kernels need to query the index
    const GridHeader* hdr;
    const float4* pts;       // sorted by cell; .w = original index (uint bits)
    const uint32_t* cell_start;  // n_cells + 1 entries
};

static constexpr int kPad = 2;           // pad cells per side (see grid_index.hip)
// progress word of the device-resident optimisers (NdtOut / VgOut): call number * kProgressWindow + passes consumed.  The pass budget
// of a call ((ndt_max_iters + 3) * 13 + 5, vgicp_max_iters * lm_inner + 3) must stay inside the window: checked where it is computed.
static constexpr double kProgressWindow = 1048576.0;
static constexpr int kBBoxBlocks = 256;  // partial bounding boxes

struct Pose16 { double m[16]; };

// ---------------------------------------------------------------------------
// LOAM Gauss-Newton state carried between launches (device memory).
// ---------------------------------------------------------------------------
struct LoamState {
    double pose[16];     // column-major map<-lidar
    int32_t done;        // 1: loop finished (converged, too few points, or error)
    int32_t converged;   // isConverge (PointCloudRegister.hpp:15)
    int32_t iters_run;   // linearisations whose result was consumed
    int32_t fail;        // 1: fewer than 6 accepted points (LoamRegister.cpp:173-176)
                         // 2: sharded call, some rank could not index its tile (kSlotRankFail)
                         // 3: a query came within one cell of a cut face of a clamped index (kSlotEscapes): the call is redone
};

static constexpr int kAccum = 32;        // 21 JtJ (upper) + 6 JtE + 1 count, padded; [28], [29] search statistics and
static constexpr int kSlotRankFail = 30; //   sharded: > 0 when a rank's index is unusable (summed over the ranks by the exchange)
static constexpr int kSlotEscapes = 31;  //   queries that left the region a clamped index covers
static constexpr int kMaxPartials = 512; // linearisation blocks
static constexpr int kTimelineSlots = 16; // s_memrealtime stamps per block and launch (pcr_get_timeline)

struct LoamConsts {
    double knn_max_sq, plane_thresh, point_thresh, pos_conv, rot_conv;
    int32_t iters, early_exit;
};

struct LoamResult {          // written by the finalize launch into host-mapped memory
    double pose[16];         // after T2SE3
    int32_t converged, iters_run, fail, grid_overflow, grid_empty, pad;
    uint64_t grid_cells;     // cells the target needs (to grow the table on overflow)
    int32_t grid_stale;
    int32_t progress;        // (launch whose prologue has run) << 1 | the loop has ended: written by every launch's prologue, read by a host that queues launches a few ahead (run_loam, early exit)
};

struct LoamTrace {           // per consumed linearisation
    double JtJ[36];
    double JtE[6];
    double x[6];
    int64_t n;
    int64_t cache_hits, searches;   // queries served by the neighbour cache / by a full search in that linearisation
};

struct NnCacheEntry;

struct LoamArgs {
    const float* src;        // scan points, stride in floats
    uint32_t n_src;
    uint32_t src_stride;
    GridView grid;
    LoamConsts c;
    double init_pose[16];    // initial guess, consumed by launch 0
    LoamState* state;        // [2], ping-pong by launch parity
    double* partials;        // [2][kMaxPartials][kAccum]
    const double* reduced;   // non-null: partial sums already reduced (and all-reduced) into kAccum doubles
    uint32_t n_partials;     // blocks of the linearisation grid
    uint32_t n_prev;         // rows the PREVIOUS launch wrote (0: n_partials) -- a launch of half blocks writes twice as many
    uint32_t half;           // this launch: 128 queries per block (the upper half of its threads only helps searching), twice the blocks
    LoamTrace* trace;        // [iters] or null
    LoamResult* result;      // final pose (after T2SE3) and flags
    // optional per-point outputs (tests): null in production
    int8_t* dbg_status;
    double* dbg_rows;
    int32_t* dbg_nn;
    // optional query tile (multi-GPU): process only queries inside [lo,hi)
    struct NnCacheEntry* nn_cache;   // [n_src] neighbours of the previous iteration (loam.hip), or null
    int32_t use_tile;
    double tile_lo[3], tile_hi[3];
    // profiling aid (development builds: PCR_ABLATE): skip phases to price them.  bit0: candidate loop,
    // bit1: plane fit and everything after it, bit2: prologue solve.  Results are then meaningless.
    int32_t ablate;
    int32_t coresident;      // pcr_params.loam_coresident = 1: the two-waves-per-SIMD variant of the iterate kernel (loam.hip)
    int32_t rank_fail;       // sharded: this rank has no usable index (its grid view is a dummy marked overflow); told to the others
    // profiling aid (pcr_params.record_timeline = 1): [launch][block][kTimelineSlots] s_memrealtime stamps (100 MHz) taken by thread 0
    unsigned long long* timeline;
};

// host-side launchers (grid_index.hip / loam.hip)
struct DeviceBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return static_cast<T*>(p); }
};

static constexpr int kMaxBins = 8192;    // tiles of the tiled build path (grid_index.hip): one 32 KB LDS histogram
static constexpr int kMaxTileShift = 13; // a tile's cells are histogrammed in LDS too

// A build that indexes only the points inside a region (pcr_scan2map of NDT: voxel Gaussians depend on a voxel's own points alone, so
// a lattice that holds nothing outside the region the scan can reach serves that scan as the full one would).  Possible only when the
// build reuses the header and the tile layout of a previous FULL build (the lattice the mask is laid over must be known before the
// points are binned, and the room of every tile is the full cloud's): build() then calls enqueue_mask -- which queues the marking of the
// region on the build's stream and fills mask / mshift -- and sets `applied`; otherwise it builds in full and leaves `applied` false.
// What the tile pass of a FILTERED build does for NDT while a tile's cell counts are still in LDS (grid_index.hip: grid_tile_kernel<.., kTail>):
// the cells inside the mask with at least min_points points are appended to `list` (one atomic on `count` per tile), the slot of every cell is
// written -- 0 inside the mask, kNdtUnprepared outside -- and `count_next` is cleared: what ndt_candidates_kernel does in a launch of its own
// by reading the finished table again (ndt.hip).
static constexpr uint32_t kNdtUnprepared = 0xffffffffu;      // slot of a cell the target was not prepared for (ndt.hip)
struct TileTail {
    uint32_t* vox_slot; uint32_t* list; uint32_t* count; uint32_t* count_next;
    int32_t min_points; uint32_t capacity;
    const uint8_t* mask; int32_t mshift;      // (filled by build() from the filter's mask)
};
struct BuildFilter {
    std::function<hipError_t()> enqueue_mask;
    const uint8_t* mask = nullptr;      // one byte per macro cell of the lattice (RoiView::mask)
    int mshift = 0;
    bool applied = false;
    bool want_tail = false;             // `tail` is filled in: the tile pass may list NDT's voxel cells
    TileTail tail = {};
    bool tail_applied = false;          // ... and did (only together with `applied`, on the dense tile kernel)
};

struct GridIndex {
    DeviceBuf sorted, cell_count, cell_start, block_sums, bbox_partials, header, keys, ranks, ticket;
    DeviceBuf kept;                             // region-only builds: the points of the region as (x, y, z, original index), listed by grid_keep_kernel for the bin pass
    DeviceBuf tiled, bin_count, bin_start, tile_sq;      // tiled build path: points grouped by tile, points per tile, first point of every tile, sum of count^2 per tile
    int tiled_shift = -1;                       // log2(cells per tile) of the last build when it took the tiled path
    // Bounding-box hint: a build whose header the host has seen to be good (confirm()) lets the NEXT build of the same kind skip
    // the bounding-box pass and reuse that header -- a sub-map changes by a key frame at a time.  The bin kernel checks every
    // point against the box; one outside sets header.stale and the caller rebuilds with a fresh (and from then on padded) box.
    bool hint_ok = false;
    double hint_cell = 0.0, hint_shift = 0.0;
    int hint_pcl = 0;                           // lattice kind of the build the hint comes from (a hint serves only a build of the same kind)
    int hint_margin = 0;                        // cells added around a fresh box in x and y (0 until a hint has failed once)
    int hint_margin_z_pcl = 0;                  // ... and in z, for a pcl::VoxelGrid lattice (the voxel filter; the other lattices: see grid_bbox_header_kernel)
    bool used_hint = false;                     // the last build() reused the header
    bool no_hints = false;                      // pcr_params.index_no_hints: never reuse a header or a tile layout
    bool coherent_input = false;                // the clouds are stored in a spatially coherent order (scans, ring by ring): the tile pass counts by runs of equal cells (grid_tile_kernel: kRuns)
    bool cut_sparse = false;                    // sparse grids too have their heavy tiles cut into slabs by the layout hint (grid_index.hip: BINS); the voxel filter's index
    bool split_sparse_tiles = true;             // sparse grids: light and heavy tiles by an instantiation of the tile kernel each (grid_index.hip)
    bool prefer_one_level = false;              // build by the one-level path (histogram with ranks -> scan -> scatter): a small cloud on a COARSE grid puts
                                                // a third of its points into one tile, which one block of the tiled path then sorts alone
    // Layout hint: where each BIN's points may go in `tiled` (an eighth more room than the bin held + 32) and which bins the tiles are cut into
    // (grid_index.hip: BINS), planned by one block of every tiled build's tile pass for the next one; two buffers, alternating.  Used only
    // together with a reused header.
    DeviceBuf layout[2];
    int lay_idx = 0, lay_shift = -1;
    uint32_t lay_nb_max = 0;                    // bins the layout in hand was planned for (the next build must have room for as many)
    bool lay_cuts = false;                      // ... and it may hold tiles that are cut (the bin pass then reads the tiles' words)
    size_t lay_n = 0;
    uint32_t lay_room_add = 32;                 // ... + lay_room_add
    int lay_room_shift = 3;                     // a bin's room in the layout: what it held + that >> lay_room_shift + 32 (the voxel filter's clouds change more from call to call: 1)
    bool lay_ok = false, used_layout = false;
    bool filtered = false;                      // the last build() indexed the points of a region only (BuildFilter)
    // A build WITHOUT hints computes the bounding box itself; then (grid_bbox_header_kernel)
    GridHeader* header_mirror = nullptr;        //   the header is also written to this host-mapped address (no copy queued behind the build): `mirrored` says so
    bool mirrored = false;
    GridIndex* twin = nullptr;                  //   and `twin`, an index the caller builds NEXT over the same cloud with cell twin_cell, gets its header from the
    double twin_cell = 0.0;                     //   same pass (its build() then launches no box kernel)
    bool header_preset = false;
    const float* preset_pts = nullptr; size_t preset_n = 0; double preset_cell = 0.0;
    hipError_t ensure_tables(hipStream_t s, std::string* err);
    size_t effective_capacity() const;
    void confirm() { hint_ok = valid && tiled_shift >= 0; }
    size_t cell_capacity = 0;   // entries available in cell_count / cell_start
    uint64_t cells_hint = 0;    // cells of the last header of this index the host has seen (0: none): bounds the next build's cell count and sets its tile size
    void note_cells(uint64_t n_cells) { if (n_cells) cells_hint = n_cells; }
    size_t n_points = 0;
    size_t reserve_points = 0;  // build() sizes its point-sized buffers for at least that many points (a caller whose clouds grow: every growth is an allocation and a free, device-wide stops)
    bool valid = false;
    GridView view() const {
        return GridView{header.as<GridHeader>(), sorted.as<float4>(), cell_start.as<uint32_t>()};
    }
    // Enqueue the build of the index over n points (device pointer, stride in floats).
    // No host synchronisation unless the cell table must grow.  cell = grid edge.
    hipError_t build(const float* d_pts, size_t n, size_t stride_floats, double cell, hipStream_t s,
                     std::string* err, double shift = 0.0, int pcl_mode = 0, const ClampBox* clamp = nullptr, bool allow_hint = false,
                     BuildFilter* filter = nullptr);
    // Make room for `need_cells` cells (+1 start) after the device reported overflow.
    hipError_t grow_cells(uint64_t need_cells, std::string* err);
    // header.sum_sq_u <- sum over the cells of count^2 (enqueued; VGICP reads it back with the header)
    hipError_t enqueue_density(hipStream_t s);
    void release();
};

// ---------------------------------------------------------------------------
// VGICP (vgicp.hip)
// ---------------------------------------------------------------------------
struct VgicpVoxel {          // fast_vgicp_voxel.hpp:59-82 (GaussianVoxel, ADDITIVE, finalized)
    double mean[3];
    double cov[6];           // xx xy xz yy yz zz
    double w;                // sqrt(num_points)
    uint32_t n, pad;
};

// Region of interest of a target that is prepared for ONE scan (pcr_scan2map: the reference rebuilds its target structures for every
// call -- fast_vgicp_impl.hpp:66-67, ndt_omp.h:276-283 -- but a scan can only ever look up the voxels near where its points land).
// The lattice of the target index is grouped into macro cells of 2^mshift cells per axis (about 2 m); the macro cells that hold a scan
// point at the initial pose are marked, every mark is spread over the macro cells within base + 0.05 x (distance from the sensor) metres
// -- a translation of `base`, a rotation of 0.05 rad (grid_index.hip: roi_*) -- and only voxels inside marked macro cells are prepared (per-point covariances + voxel Gaussians for VGICP, voxel Gaussians for NDT).  Every lookup of the
// optimiser checks the cell it hits: an occupied cell OUTSIDE the region is counted in *escapes, and the host then prepares the whole
// target and repeats the call -- so the result is the full preparation's whenever the call succeeds (a pose that moves the scan by
// more than the margin is rare and merely costs the repeat).
struct RoiView {
    const GridHeader* lat;       // the lattice (header of the target index); macro dims = ceil(dims / 2^mshift)
    const uint8_t* mask;         // one byte per macro cell, != 0: prepared.  nullptr: no region -- everything is prepared
    uint32_t* escapes;
    int32_t mshift;
    int32_t filtered;            // the index itself holds the region's points only (BuildFilter): NDT then treats EVERY cell outside the mask as unprepared
    uint32_t* count;             // profiling passes only (pcr_set_profile >= 2), else nullptr: [0] += target points whose covariance was computed, [16] += voxels built
};
__host__ __device__ inline uint32_t roi_macro(const GridHeader& h, int mshift, int cx, int cy, int cz) {      // cell (cx, cy, cz) inside the lattice
    const uint32_t m0 = ((uint32_t)h.dims[0] + (1u << mshift) - 1u) >> mshift, m1 = ((uint32_t)h.dims[1] + (1u << mshift) - 1u) >> mshift;
    return (((uint32_t)cz >> mshift) * m1 + ((uint32_t)cy >> mshift)) * m0 + ((uint32_t)cx >> mshift);
}
__host__ __device__ inline bool roi_mask_holds_cell(const GridHeader& h, const uint8_t* mask, int mshift, uint32_t key) {      // linear cell key
    const uint32_t d0 = (uint32_t)h.dims[0], d1 = (uint32_t)h.dims[1], row = key / d0, cz = row / d1;
    return mask[roi_macro(h, mshift, (int)(key - row * d0), (int)(row - cz * d1), (int)cz)] != 0;
}
#ifdef __HIPCC__
// is the voxel that the point (x, y, z) belongs to inside the prepared region?  (lat: the voxel lattice's header)
__device__ __forceinline__ bool roi_holds_point(const RoiView& roi, const GridHeader& lat, double x, double y, double z) {
    const double fx = floor(x / lat.cell - lat.shift) - lat.org[0], fy = floor(y / lat.cell - lat.shift) - lat.org[1], fz = floor(z / lat.cell - lat.shift) - lat.org[2];
    if (!(fx >= 0.0 && fx < (double)lat.dims[0] && fy >= 0.0 && fy < (double)lat.dims[1] && fz >= 0.0 && fz < (double)lat.dims[2])) return false;
    return roi.mask[roi_macro(lat, roi.mshift, (int)fx, (int)fy, (int)fz)] != 0;
}
#endif
// A few hundred words a kernel stores on the side for a later launch on its stream: the initial state of NDT's device-resident optimiser rides
// on the first launch of its call (the region's mark pass).  A launch of its own cost ~5 us -- a kernel reads its argument block over PCIe --
// and ~2 us of gap, on a stream where nothing else could run meanwhile.
struct BlobStore { uint32_t* dst; uint32_t* zero; uint32_t n; uint32_t pad; uint32_t w[416]; };
hipError_t roi_launch(const GridIndex& lattice, const float* d_src, size_t n_src, size_t stride_floats, const Pose16& T, int mshift,
                      uint8_t* d_mark, uint8_t* d_mark_next, uint8_t* d_tmp, uint8_t* d_mask, double base_m, double per_m, hipStream_t s,
                      const BlobStore* blob = nullptr);

struct VgicpArgs {
    const float* src; uint32_t n_src, src_stride;
    const double* src_cov6;      // per source point, original order
    const GridHeader* hdr;       // target index header (voxel lattice = index grid shifted by half a cell)
    const uint32_t* cell_start;  // of the target index: voxel = cell, stored at the position of the cell's first point
    const VgicpVoxel* vox;
    uint32_t* corr_slot;         // [n_src] voxel slot + 1 of the correspondence, 0 = none
    double* corr_M;              // [n_src][6] Mahalanobis matrix of the correspondence
    uint32_t* corr_slot_next;    // the same two for the linearisation an LM trial pass computes ahead (vgicp_launch_error)
    double* corr_M_next;
    double* partials;            // [blocks][32]
    int32_t use_tile, pad_;      // sharded target: only source points whose transformed position lies in [tile_lo, tile_hi)
    double tile_lo[3], tile_hi[3];
    uint32_t* escapes;           // target index cut to the bulk of the cloud (header.clamped): count of source points that land within
    int32_t guard_cells, pad2_;  //   guard_cells voxels of a face with target points beyond it (their voxels' covariances may lack neighbours); else NULL
    RoiView roi;                 // mask == nullptr: the whole target is prepared
};

// Halo check of a sharded target (pcr_set_shard): for every point inside [chk_lo, chk_hi) the 20th neighbour must be nearer
// than the faces of [ext_lo, ext_hi) -- the region the rank's cloud is complete in; *violations counts the others.
struct CovCheck { double chk_lo[3], chk_hi[3], ext_lo[3], ext_hi[3]; uint32_t* violations; };
// Work memory of the two-class covariance search of a scan-sized cloud (cov_search.hip): neighbour lists [20][n], the queue of the queries
// the lane-per-query kernel hands to the wave-per-query one (+ the bound each brings along), its counter.  One per stream that runs it.
struct CovScratch {
    DeviceBuf nbr, queue, seed, count;
    // a map-sized target prepared for one scan (vgicp.hip: vgicp_region_list_kernel): the sorted positions of the region's points, and two counters
    // used alternately (each call leaves the other one cleared for the next)
    DeviceBuf region_list, region_count;
    int region_idx = 0;
    hipError_t reserve(size_t n);
    hipError_t reserve_region(size_t n, hipStream_t s);
    void release();
};
// ev (optional, profiling passes): 6 events = begin / end of the three kernels
hipError_t cov_search_launch(const GridIndex& grid, const GridIndex* coarse1, const GridIndex* coarse2, const float* d_orig, size_t stride_floats,
                             size_t n, double* d_cov6, hipStream_t s, const CovCheck* check, const RoiView* roi, CovScratch& sc, hipEvent_t* ev = nullptr);
// scratch: scan-sized clouds (n <= 300 000) go through cov_search.hip when given one
hipError_t vgicp_launch_cov(const GridIndex& grid, const GridIndex* coarse1, const GridIndex* coarse2, const float* d_orig, size_t stride_floats,
                            size_t n, double* d_cov6, hipStream_t s, const CovCheck* check = nullptr, const RoiView* roi = nullptr, CovScratch* scratch = nullptr,
                            hipEvent_t* ev = nullptr);
hipError_t vgicp_launch_voxels(const GridIndex& grid, const double* d_cov6, VgicpVoxel* d_vox, hipStream_t s, const RoiView* roi = nullptr);
// (seq: written last into d_out32[31] / d_out48[47], host-mapped: the completion word the host spins on)
hipError_t vgicp_launch_linearize(const VgicpArgs& a, const Pose16& T, double* d_out32, hipStream_t s, double seq = 0.0);
// device-resident Levenberg-Marquardt loop (vgicp_opt.h): state in HBM, result in host-mapped memory
struct VgCtl;
struct VgOut;
hipError_t vgicp_launch_ctl_init(VgCtl* d_ctl2, const Pose16& guess, int max_iters, int lm_inner, double lm_init_scale, double rot_eps, double trans_eps, hipStream_t s,
                                 uint32_t* d_roi_escapes = nullptr);
struct PeerComm;
hipError_t vgicp_launch_pass_pro(const VgicpArgs& a, VgCtl* d_ctl2, double* d_rows2, VgOut* d_out, hipStream_t s, double seq, int index,
                                 const PeerComm* pc = nullptr, double xseq = 0.0, double* d_reduced = nullptr);
// out32[28] = compute_error(T); out32[0..27] = the linearisation at T (correspondences into a.corr_*_next)
hipError_t vgicp_launch_error(const VgicpArgs& a, const Pose16& T, double* d_out32, hipStream_t s, double seq = 0.0);
// out32[0] = sum of the squared 1-NN distances <= max_range, [1] = their number, [2] = (tile given) source points whose nearest
// neighbour could lie outside [ext_lo, ext_hi)
struct FitTile { int32_t use, pad_; double lo[3], hi[3], ext_lo[3], ext_hi[3]; };
hipError_t fitness_launch(const GridIndex& grid, const float* d_src, size_t n_src, size_t stride_floats, const double pose[16], double max_range,
                          double* d_partials, double* d_out32, hipStream_t s, double seq = 0.0, const FitTile* tile = nullptr);
uint32_t vgicp_blocks(uint32_t n_src);

// pcl::VoxelGrid on the device (voxel_filter.hip); grid must have been built with pcl_mode = 1
// (two launches queued behind the build; they read the header themselves and do nothing when it says overflow, stale or empty.  d_inten: n floats,
//  d_sums: n / 2048 + 2 words, d_wave: voxel_filter_wave_bytes(n); result_mapped: page-locked host memory the last block writes -- valid once the stream has been synchronised)
struct VfResult { uint32_t count; int32_t overflow, too_fine, stale, empty; uint32_t pad_; unsigned long long n_cells; };
size_t voxel_filter_wave_bytes(size_t n);
hipError_t voxel_filter_launch(const GridIndex& grid, const float* d_orig, size_t stride_floats, size_t n, uint32_t* d_inten, uint32_t* d_sums,
                               void* d_wave, float* d_out, size_t out_capacity, void* result_mapped, hipStream_t s);

// ---------------------------------------------------------------------------
// NDT (ndt.hip)
// ---------------------------------------------------------------------------
struct NdtVoxel {            // VoxelGridCovariance::Leaf (pclomp/voxel_grid_covariance_omp.h:98-193): mean_, icov_
    double mean[3];
    double icov[9];          // row-major
    int32_t n, pad;
};
struct NdtPose { float R[9]; float t[3]; };              // final_transformation_ (Matrix4f), row-major R
struct NdtAngles {           // computeAngleDerivatives (ndt_omp_impl.hpp:289-395)
    float j[8][3], h[15][3];         // float tables j_ang / h_ang
    double jd[8][3], hd[15][3];      // double vectors j_ang_a_.. / h_ang_a2_..
};
struct NdtArgs {
    const float* src; uint32_t n_src, src_stride;
    const GridHeader* hdr;
    const uint32_t* vox_slot;
    const NdtVoxel* vox;
    double d1, d2;           // gauss_d1_, gauss_d2_ (ndt_omp_impl.hpp:86-93)
    double* partials;        // [blocks][48]
    int32_t use_tile, pad_;  // sharded target: only source points whose transformed position lies in [tile_lo, tile_hi)
    double tile_lo[3], tile_hi[3];
    uint32_t* roi_escapes;   // target prepared for one scan (RoiView): count of lookups that hit a voxel it was not prepared for; else NULL
    uint32_t* pair_count;    // profiling passes only (pcr_set_profile >= 2), else NULL: [32] += (point, voxel) pairs of gradient-only passes, [48] += of passes with a Hessian
};
hipError_t ndt_launch_voxels(const GridIndex& grid, uint32_t* d_slot, NdtVoxel* d_vox, uint32_t* d_count, uint32_t* d_count_next, uint32_t* d_list, size_t list_capacity,
                             int min_points, double eig_mult, hipStream_t s, const RoiView* roi = nullptr, bool listed_by_tile_pass = false);
hipError_t ndt_launch_derivatives(const NdtArgs& a, const NdtPose& T, const NdtAngles& ang, int compute_hessian, double* d_out48, hipStream_t s, double seq = 0.0);
hipError_t ndt_launch_hessian(const NdtArgs& a, const NdtPose& T, const NdtAngles& ang, double* d_out48, hipStream_t s, double seq = 0.0);
// device-resident optimiser (ndt_opt.h): controller state in HBM, result in host-mapped memory
struct NdtCtl;
struct NdtOut;
hipError_t ndt_launch_ctl_init(NdtCtl* d_ctl, const NdtPose& T0, const double p[6], double step_size, double trans_eps, int max_iters, hipStream_t s, int no_replay = 0,
                               uint32_t* d_roi_escapes = nullptr);
void ndt_ctl_init_blob(BlobStore* b, NdtCtl* d_ctl, const NdtPose& T0, const double p[6], double step_size, double trans_eps, int max_iters, int no_replay_arg);
hipError_t ndt_launch_pass(const NdtArgs& a, NdtCtl* d_ctl, NdtOut* d_out, hipStream_t s, double seq);
hipError_t ndt_launch_pass_pro(const NdtArgs& a, NdtCtl* d_ctl2, double* d_rows2, NdtOut* d_out, hipStream_t s, double seq, int index,
                               hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
hipError_t ndt_launch_pass_fold(const NdtArgs& a, NdtCtl* d_ctl, double* d_sums48, hipStream_t s);
hipError_t ndt_launch_ctl(const NdtArgs& a, NdtCtl* d_ctl, const double* d_sums48, NdtOut* d_out, hipStream_t s, double seq, int batch_mark);
struct PeerComm;
hipError_t ndt_launch_pass_peer(const NdtArgs& a, NdtCtl* d_ctl, const PeerComm& pc, double xseq, NdtOut* d_out, hipStream_t s, double seq, int batch_mark);
uint32_t ndt_blocks(uint32_t n_src);

// peer exchange (loam.hip): receive buffers of all ranks as THIS process maps them
static constexpr int kMaxPeers = 8;
static constexpr int kPeerSlot = 72;                       // doubles per (parity, writer) slot: 64 values + the sequence word + padding
static constexpr int kPeerFlag = 64;
static constexpr unsigned long long kPeerTimeoutTicks = 200000000ull;      // 2 s of the 100 MHz clock
struct PeerComm { double* buf[kMaxPeers]; int32_t rank, nranks; int32_t* status; };      // status: host-mapped word, set to 1 by an exchange that timed out (out of band: the sums may hold any value, NaN included)
hipError_t loam_launch_peer_exchange(const LoamArgs& a, int k, const PeerComm& pc, double seq, double* d_out, hipStream_t s);
hipError_t peer_launch_allreduce(double* d_inout, int n, int op, const PeerComm& pc, double seq, hipStream_t s);
hipError_t loam_launch_iteration(const LoamArgs& a, int k, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr, bool allow_half = false);
hipError_t loam_launch_finalize(const LoamArgs& a, int k, hipStream_t s);
hipError_t loam_launch_reduce(const LoamArgs& a, int k, double* d_out, hipStream_t s);
uint32_t loam_grid_blocks(uint32_t n_src);

}  // namespace pcr

// (library-internal, not part of the C ABI) the stream a handle queues its work on: the sub-map assembly queues its transform pass in front of the
// voxel filter it runs through a handle, on that handle's stream, so that one synchronisation serves both
struct pcr_handle;
hipStream_t pcr_internal_stream(const pcr_handle* h);
// ... and the voxel filter in two halves, device memory into device memory (out_capacity >= n): queue it; synchronise and collect it (redone there if the index's hints did not hold)
int pcr_internal_vf_begin(pcr_handle* h, const void* d_pts, size_t n, size_t stride_bytes, double leaf, void* d_out, size_t out_capacity);
int pcr_internal_vf_end(pcr_handle* h, size_t* n_out);
// ... and a hint: clouds of up to `points` points will come (the filter's and its index's buffers are sized for that at the next call)
void pcr_internal_vf_reserve(pcr_handle* h, size_t points);
