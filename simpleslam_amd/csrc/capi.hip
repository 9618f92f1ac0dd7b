// capi.hip -- the C ABI of libpcr_hip.so (include/pcr_hip.h): handle management,
// host<->HBM staging and the launch sequence of each registration method.
// Host C++ only talks to the kernels through the launchers in pcr_internal.h.
#include <dlfcn.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <vector>

#include "pcr_internal.h"
#include "ndt_opt.h"
#include "vgicp_opt.h"
#include "small_math.h"

using namespace pcr;

namespace {

thread_local std::string g_create_error;

enum Method { kLoam = 0, kNdt = 1, kVgicp = 2 };

// --- RCCL, loaded lazily (multi-GPU sharded mode only) ---------------------------
struct NcclId { char internal[128]; };
typedef int (*nccl_get_id_fn)(NcclId*);
typedef int (*nccl_init_rank_fn)(void**, int, NcclId, int);
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_destroy_fn)(void*);
typedef int (*nccl_comm_int_fn)(void*, int*);
struct Rccl {
    void* lib = nullptr;
    nccl_get_id_fn get_id = nullptr;
    nccl_init_rank_fn init_rank = nullptr;
    nccl_allreduce_fn allreduce = nullptr;
    nccl_destroy_fn destroy = nullptr;
    nccl_comm_int_fn comm_count = nullptr, comm_user_rank = nullptr;
    bool load(std::string* err) {
        if (lib) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { if (err) *err = std::string("dlopen(librccl) failed: ") + dlerror(); return false; }
        get_id = (nccl_get_id_fn)dlsym(lib, "ncclGetUniqueId");
        init_rank = (nccl_init_rank_fn)dlsym(lib, "ncclCommInitRank");
        allreduce = (nccl_allreduce_fn)dlsym(lib, "ncclAllReduce");
        destroy = (nccl_destroy_fn)dlsym(lib, "ncclCommDestroy");
        comm_count = (nccl_comm_int_fn)dlsym(lib, "ncclCommCount");
        comm_user_rank = (nccl_comm_int_fn)dlsym(lib, "ncclCommUserRank");
        if (!get_id || !init_rank || !allreduce || !destroy) { if (err) *err = "librccl lacks ncclGetUniqueId/CommInitRank/AllReduce"; return false; }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

}  // namespace

#ifdef PCR_DEV_SWITCHES
namespace pcr { int dev_stamps(unsigned long long* out, size_t count); }
#endif

struct pcr_handle {
    Method method = kLoam;
    pcr_params prm;
    std::string err;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int profile = 1;

    // target
    GridIndex grid;
    DeviceBuf tgt_stage;         // host targets are staged here
    const float* tgt_ptr = nullptr;   // device pointer the index was built from (for rebuild on overflow)
    size_t tgt_n = 0, tgt_stride = 0;
    bool have_target = false;
    ClampBox clamp{};                // LOAM scan2map fallback: region of interest of a target whose full box cannot be tabulated
    bool clamp_allowed = false;      // set while a scan2map call (target rebuilt for this very scan) is running
    uint32_t last_blocks = 0;        // linearisation blocks of the last LOAM call (timeline readout)
    // source
    DeviceBuf src_stage;
    GridIndex vf_grid;               // pcl::VoxelGrid lattice of the cloud being down-sampled (pcr_voxel_filter)
    DeviceBuf vf_in, vf_out, vf_head, vf_sums, vf_count;

    // LOAM work memory
    DeviceBuf loam_state, loam_partials, loam_trace, loam_reduced, dbg_status, dbg_rows, dbg_nn, nn_cache, timeline;
    LoamResult* result_host = nullptr;   // host-mapped, written by the finalize kernel
    LoamResult* result_dev = nullptr;
    std::vector<LoamTrace> trace_host;
    int trace_iters = 0;

    // VGICP work memory
    GridIndex src_grid;
    GridIndex cov_l1, cov_l2;        // the TARGET cloud while its covariances are computed, indexed at 4x and 16x the cell
    GridIndex src_l1, src_l2;        // the same for the source (own buffers: its side runs on side_stream next to the target's)
    hipStream_t side_stream = nullptr;     // source index + covariances of a scan2map call, concurrent with the target preparation
    hipEvent_t ev_side_in = nullptr, ev_side_done = nullptr, ev_hdr = nullptr, ev_aux_in = nullptr, ev_aux_done = nullptr;
    hipStream_t aux_stream = nullptr; // the target's covariance grid, built beside its voxel lattice (settle_cov_levels)
    GridHeader* side_hdr = nullptr;  // pinned: headers of the three source levels [0..2] and of the target's grids [3..6], read back without blocking the host
    bool side_pending = false;       // source work of (side_src, side_n, side_stride) is in flight on side_stream
    const float* side_src = nullptr; size_t side_n = 0, side_stride = 0;
    GridHeader cov_hdr0;             // header of the fine level of the last settle_cov_levels (density estimate)
    double cov_scale_hint = 0.0;     // cell scale of the last map-sized target's covariance grid: built ahead of the density it is derived from
    bool cov_l1_ahead = false;       // ... and whether that build is the one in h->cov_l1 now
    DeviceBuf tgt_cov6, src_cov6, vox, corr_slot, corr_M, corr_slot2, corr_M2, vg_partials;
    CovScratch src_scratch, tgt_scratch;   // neighbour lists + queue of the covariance search of a scan-sized cloud: the source's runs on the side stream beside the target's
    double seq = 0.0;                    // completion numbers of the host-mapped result blocks below
    double* out32_host = nullptr;        // host-mapped: 32 doubles written by sum_partials_kernel
    double* out32_dev = nullptr;
    bool vg_target_ready = false;
    int vg_outer = 0, vg_lin = 0, vg_err = 0;
    DeviceBuf vg_ctl;                    // two VgCtl: the device-resident LM loop's state, by launch parity
    // getFitnessScore() is a call of its own in the reference (VgicpRegister.cpp:42-45: PCL evaluates it when asked, from the source it
    // still holds): an unsharded alignment keeps a copy of the scan and the final pose, and pcr_fitness() evaluates the score on demand
    DeviceBuf fit_src;
    const float* fit_copied_from = nullptr;   // the scan fit_src holds (copied on the side stream by vgicp_source_enqueue)
    size_t fit_n = 0, fit_stride = 0;
    double fit_pose[16];
    bool fit_pending = false;
    VgOut* vg_out_host = nullptr;        // host-mapped: its result and progress word
    VgOut* vg_out_dev = nullptr;

    // NDT work memory
    DeviceBuf nd_slot, nd_vox, nd_count, nd_list, nd_partials;
    double* out48_host = nullptr;        // host-mapped: 48 doubles written by ndt_sum_partials_kernel
    double* out48_dev = nullptr;
    DeviceBuf nd_ctl;                    // NdtCtl: the device-resident optimiser's state
    struct VfJob { const float* d_pts; size_t n, sf; double leaf; float* d_out; size_t cap; } vf_job = {};      // the filter that is queued (vf_enqueue / vf_settle)
    bool vf_inflight = false; size_t vf_inflight_n = 0;      // pcr_voxel_filter_begin has queued a filter that pcr_voxel_filter_end has not collected
    unsigned long long vf_builds = 0, vf_stale = 0;      // index builds of the voxel filter, and how many found the reused box / layout too small
    char* vf_ret = nullptr;              // page-locked: what the voxel filter's last block reports (VfResult: the voxel count + the index header's verdict)
    DeviceBuf vg_reduced;                // sharded VGICP over the peer exchange: a pass's 32 sums folded over the rows and the ranks
    DeviceBuf nd_sums;                   // sharded device loop: the 48 sums of a pass, all-reduced in place
    NdtOut* nd_out_host = nullptr;       // host-mapped: its result and progress word
    NdtOut* nd_out_dev = nullptr;
    bool clamp_from_bulk = false;        // pcr_set_target in progress: an untabulatable box may be cut to the bulk of the target
    uint64_t map_id = 0, map_gen = 0;    // pcr_scan2map_submap: the sub-map the target structures were built from
    long long target_builds = 0;         // ... and how often it had to build them
    bool nd_grid_checked = false, nd_grid_bad = false, nd_grid_empty = false;      // the device loop reported the state of the index header with its result
    uint64_t nd_grid_cells = 0;
    int nd_count_idx = 0;
    int nd_last_passes = 8;              // passes the previous alignment took: how many are enqueued up front
    bool nd_target_ready = false;
    int nd_iters = 0, nd_deriv = 0, nd_hess = 0;
    double nd_score = 0;

    // multi-GPU
    int use_tile = 0;
    double tile_lo[3] = {0, 0, 0}, tile_hi[3] = {0, 0, 0};
    bool have_halo = false;          // pcr_set_shard: the target holds every map point inside [tile_lo - halo, tile_hi + halo)
    double halo = 0.0;
    void* comm = nullptr;            // RCCL communicator (pcr_comm_init)
    // peer exchange (pcr_comm_init_peer): this rank's receive buffer (fine-grained HBM, exported over IPC), every rank's as mapped here
    double* peer_own = nullptr;
    bool peer_on = false;
    bool peer_exported = false;      // pcr_comm_peer_export has cleared the receive buffer for a session that pcr_comm_init_peer has not opened yet
    bool peer_broken = false;        // an exchange of the session timed out: the ranks' sequence numbers no longer agree, every further exchange is refused
    int32_t* peer_status_host = nullptr;      // host-mapped: set by a kernel whose exchange timed out
    int32_t* peer_status_dev = nullptr;
    PeerComm peer{};
    double peer_seq = 0.0;
    pcr_allreduce_fn host_ar = nullptr;     // or the caller's collective (pcr_comm_init_host)
    void* host_ar_user = nullptr;
    int nranks = 1, rank = 0;
    double* red_host = nullptr;      // host-mapped, kAccum doubles: the LOAM sums on their way through the caller's collective
    double* red_dev = nullptr;
    DeviceBuf ar_stage;              // RCCL: staging of the 48 doubles the host-driven optimisers exchange
    DeviceBuf dummy_grid;            // a GridHeader marked overflow + empty: the grid view of a rank whose index failed
    DeviceBuf cov_viol;              // VGICP halo check: number of neighbourhoods that reach past the halo
    double clamp_margin = 10.0;      // LOAM ClampBox: room around the scan, doubled when a query reached a cut face
    // region of interest of a target prepared for one scan (RoiView): two marking buffers used alternately, the dilated mask, the escape counter
    DeviceBuf roi_mark[2], roi_tmp, roi_mask, roi_esc;
    BlobStore blob;                  // the optimiser's initial state as a rider of the region's mark pass (pcr_internal.h: BlobStore)
    bool blob_pending = false;       //   filled in by this call and not launched yet
    bool blob_stored = false;        //   launched by this call: run_ndt needs no launch of its own for it
    int roi_idx = 0, roi_mshift = 0;
    uint64_t roi_cells_seen = 0;     // cell count of the lattice the mark buffers were last used with (a change clears them in full)
    bool roi_on = false;             // the target structures the handle holds cover only the region of the scan they were prepared for
    long long roi_repeats = 0;       // calls that left the region and were repeated on the whole target

    // timing
    hipEvent_t ev_start = nullptr, ev_index = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_kernel;
    // profiling passes (pcr_set_profile 2) of NDT / VGICP: counters the kernels add to ([0] target points with a covariance, [16] voxels,
    // [32] / [48] (point, voxel) pairs of gradient-only / Hessian passes), events of the covariance kernels ([0..1] target, [2..7] scan's search)
    DeviceBuf prof_count;
    hipEvent_t ev_cov[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_cov_tgt_used = false, ev_cov_src_used = false;
    int nd_prof_launches = 0;
    pcr_stats stats;
    double fitness = -1.0;
};

namespace {

#define H_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { h->err = std::string(#x) + ": " + hipGetErrorString(_e); return 1; } } while (0)

// inside a loop that has launches reading the caller's buffers queued ahead of the device: an error must not return before they drained
#define H_TRY_DRAIN(x) do { hipError_t _e = (x); if (_e != hipSuccess) { h->err = std::string(#x) + ": " + hipGetErrorString(_e); (void)hipStreamSynchronize(h->stream); return 1; } } while (0)

int fail(pcr_handle* h, const std::string& msg) { h->err = msg; return 1; }

bool sharded(const pcr_handle* h) { return h->comm != nullptr || h->host_ar != nullptr || h->peer_on; }
// the peer session, if one is open: the peers' buffers unmapped, the transport off (pcr_comm_init / pcr_comm_init_host / a new pcr_comm_init_peer / pcr_destroy)
void peer_close(pcr_handle* h) {
    if (h->peer_on) for (int p = 0; p < h->peer.nranks; ++p) if (p != h->peer.rank && h->peer.buf[p]) (void)hipIpcCloseMemHandle(h->peer.buf[p]);
    memset(&h->peer, 0, sizeof h->peer);
    h->peer_on = false; h->peer_broken = false; h->peer_seq = 0.0;
}
// before an exchange is queued / after its results have arrived: a session in which an exchange timed out is over
int peer_check(pcr_handle* h) {
    if (!h->peer_on) return 0;
    if (h->peer_status_host && __atomic_load_n(h->peer_status_host, __ATOMIC_ACQUIRE) != 0) h->peer_broken = true;
    if (h->peer_broken)
        return fail(h, "peer exchange: a rank did not arrive within 2 s; the session is over (the ranks' sequence numbers no longer agree): "
                       "pcr_comm_peer_export + pcr_comm_init_peer on every rank start a new one");
    return 0;
}

// profiling passes: the counters cleared (on the handle's stream, ahead of everything the call queues), the covariance events made
uint32_t* prof_counters(const pcr_handle* h) { return (h->profile >= 2 && h->prof_count.p) ? h->prof_count.as<uint32_t>() : nullptr; }
int prof_begin(pcr_handle* h) {
    h->ev_cov_tgt_used = h->ev_cov_src_used = false; h->nd_prof_launches = 0;
    h->stats.aux_kernel_ms = 0; h->stats.region_points = h->stats.region_voxels = h->stats.pairs_grad = h->stats.pairs_hess = 0;
    if (h->profile < 2) return 0;
    H_TRY(h->prof_count.reserve(64 * sizeof(uint32_t)));
    H_TRY(hipMemsetAsync(h->prof_count.p, 0, 64 * sizeof(uint32_t), h->stream));
    for (hipEvent_t& e : h->ev_cov) if (!e) H_TRY(hipEventCreate(&e));
    return 0;
}
// ... read back once the call's work has drained
int prof_end(pcr_handle* h) {
    if (h->profile < 2 || !h->prof_count.p) return 0;
    uint32_t c[64];
    H_TRY(hipStreamSynchronize(h->stream));
    if (h->side_stream) H_TRY(hipStreamSynchronize(h->side_stream));
    H_TRY(hipMemcpy(c, h->prof_count.p, sizeof c, hipMemcpyDeviceToHost));
    h->stats.region_points = c[0]; h->stats.region_voxels = c[16]; h->stats.pairs_grad = c[32]; h->stats.pairs_hess = c[48];
    float ms = 0;
    if (h->ev_cov_tgt_used) { H_TRY(hipEventElapsedTime(&ms, h->ev_cov[0], h->ev_cov[1])); h->stats.kernel_ms = ms; h->stats.kernel_launches = 1; }
    if (h->ev_cov_src_used) {
        h->stats.aux_kernel_ms = 0;
        for (int k = 0; k < 3; ++k) { H_TRY(hipEventElapsedTime(&ms, h->ev_cov[2 + 2 * k], h->ev_cov[3 + 2 * k])); h->stats.aux_kernel_ms += ms; }
    }
    if (h->nd_prof_launches > 0) {
        h->stats.kernel_ms = 0;
        for (int k = 0; k < h->nd_prof_launches; ++k) { H_TRY(hipEventElapsedTime(&ms, h->ev_kernel[2 * k], h->ev_kernel[2 * k + 1])); h->stats.kernel_ms += ms; }
        h->stats.kernel_launches = h->nd_prof_launches;
    }
    return 0;
}

// Combine n (<= 64) doubles held in host memory over the ranks of a sharded handle, in place: the caller's collective, or RCCL
// through a device staging buffer.  The stream is idle when this is called (the values were just waited for).
int ranks_allreduce(pcr_handle* h, double* v, int n, int op = 0) {
    if (h->host_ar) {
        const int rc = h->host_ar(v, (size_t)n, op, h->host_ar_user);
        if (rc != 0) return fail(h, "the caller's all-reduce failed with code " + std::to_string(rc));
        return 0;
    }
    if (h->peer_on) {
        if (peer_check(h)) return 1;
        H_TRY(h->ar_stage.reserve(64 * sizeof(double)));
        H_TRY(hipMemcpyAsync(h->ar_stage.p, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        h->peer_seq += 1.0;
        H_TRY(peer_launch_allreduce(h->ar_stage.as<double>(), n, op, h->peer, h->peer_seq, h->stream));
        H_TRY(hipMemcpyAsync(v, h->ar_stage.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        return peer_check(h);      // (the status word, not the values: a sum may be NaN in its own right)
    }
    if (h->comm) {
        H_TRY(h->ar_stage.reserve(64 * sizeof(double)));
        H_TRY(hipMemcpyAsync(h->ar_stage.p, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        const int rc = g_rccl.allreduce(h->ar_stage.p, h->ar_stage.p, (size_t)n, /*ncclFloat64*/ 8, op == 1 ? /*ncclMax*/ 2 : /*ncclSum*/ 0, h->comm, h->stream);
        if (rc != 0) return fail(h, "ncclAllReduce failed with code " + std::to_string(rc));
        H_TRY(hipMemcpyAsync(v, h->ar_stage.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
    }
    return 0;
}

// pcr_set_shard: faces farther out than this are open (the outer tiles reach to +-1e30, shard.py)
inline bool open_face(double v) { return !(fabs(v) < 1e29); }
void shard_extent(const pcr_handle* h, double ext_lo[3], double ext_hi[3]) {
    for (int d = 0; d < 3; ++d) {
        ext_lo[d] = open_face(h->tile_lo[d]) ? -1e300 : h->tile_lo[d] - h->halo;
        ext_hi[d] = open_face(h->tile_hi[d]) ? 1e300 : h->tile_hi[d] + h->halo;
    }
}

double grid_cell_for(double max_sq) {
    // smallest power of two >= the gate radius, so that x / cell is exact in f64
    double r = sqrt(max_sq > 0 ? max_sq : 1.0), c = 1.0;
    while (c < r) c *= 2.0;
    while (c * 0.5 >= r) c *= 0.5;
    return c;
}

int ensure_loam_buffers(pcr_handle* h) {
    H_TRY(h->loam_state.reserve(2 * sizeof(LoamState)));
    H_TRY(h->loam_partials.reserve((size_t)2 * kMaxPartials * kAccum * sizeof(double)));
    H_TRY(h->loam_reduced.reserve(kAccum * sizeof(double)));
    const int iters = std::max(1, h->prm.loam_iters);
    if (h->prm.record_trace) H_TRY(h->loam_trace.reserve((size_t)iters * sizeof(LoamTrace)));
    if (!h->result_host) {
        H_TRY(hipHostMalloc((void**)&h->result_host, sizeof(LoamResult), hipHostMallocMapped));
        H_TRY(hipHostGetDevicePointer((void**)&h->result_dev, h->result_host, 0));
    }
    if (h->host_ar && !h->red_host) {
        H_TRY(hipHostMalloc((void**)&h->red_host, kAccum * sizeof(double), hipHostMallocMapped));
        H_TRY(hipHostGetDevicePointer((void**)&h->red_dev, h->red_host, 0));
    }
    return 0;
}

void fill_loam_args(pcr_handle* h, LoamArgs* a, const float* d_src, size_t n_src, size_t stride_floats, const double pose[16]) {
    memset(a, 0, sizeof(*a));
    a->src = d_src; a->n_src = (uint32_t)n_src; a->src_stride = (uint32_t)stride_floats;
    a->grid = h->grid.view();
    a->c.knn_max_sq = h->prm.loam_knn_max_sq; a->c.plane_thresh = h->prm.loam_plane_thresh;
    a->c.point_thresh = h->prm.loam_point_thresh; a->c.pos_conv = h->prm.loam_pos_conv; a->c.rot_conv = h->prm.loam_rot_conv;
    a->c.iters = h->prm.loam_iters; a->c.early_exit = h->prm.loam_early_exit;
    memcpy(a->init_pose, pose, 16 * sizeof(double));
    a->state = h->loam_state.as<LoamState>();
    a->partials = h->loam_partials.as<double>();
    a->reduced = nullptr;
    a->n_partials = loam_grid_blocks((uint32_t)n_src);
    h->last_blocks = a->n_partials;
    a->trace = h->prm.record_trace ? h->loam_trace.as<LoamTrace>() : nullptr;
    a->result = h->result_dev;
    a->ablate = dev_env("PCR_ABLATE") ? atoi(dev_env("PCR_ABLATE")) : 0;      // (only a -DPCR_ABLATION development build looks at it)
    a->coresident = h->prm.loam_coresident == 1;
    if (h->prm.loam_disable_cache == 0 && h->nn_cache.reserve((n_src + 1) * 192) == hipSuccess) a->nn_cache = (NnCacheEntry*)h->nn_cache.p;
    if (h->prm.record_timeline == 1 && h->timeline.reserve((size_t)(std::max(1, h->prm.loam_iters) + 1) * kMaxPartials * kTimelineSlots * sizeof(unsigned long long)) == hipSuccess)
        a->timeline = h->timeline.as<unsigned long long>();
    a->use_tile = h->use_tile;
    for (int d = 0; d < 3; ++d) { a->tile_lo[d] = h->tile_lo[d]; a->tile_hi[d] = h->tile_hi[d]; }
}

int build_target(pcr_handle* h, const float* d_dst, size_t n_dst, size_t stride_floats) {
    double cell = 1.0;
    if (h->method == kLoam) cell = grid_cell_for(h->prm.loam_knn_max_sq);
    // (the bounding box of the previous target is tried first: GridIndex::hint_ok -- unless pcr_params.index_no_hints)
    h->grid.no_hints = h->prm.index_no_hints != 0;
    hipError_t e = h->grid.build(d_dst, n_dst, stride_floats, cell, h->stream, &h->err, 0.0, 0, h->clamp.use ? &h->clamp : nullptr, h->method == kLoam);
    if (e != hipSuccess) return 1;
    h->tgt_ptr = d_dst; h->tgt_n = n_dst; h->tgt_stride = stride_floats; h->have_target = true;
    return 0;
}

// After a synchronisation: did the device-side build overflow the cell table?  Then grow and rebuild.
// Returns 0 ok (no overflow), 2 rebuilt (caller must rerun), 1 error.
// Dense tables stop at 4e9 cells.  A cloud that needs more -- a stray point kilometres away from the map -- is refused,
// except where the caller's scan tells which part of it can matter: LOAM scan2map then indexes only the target points
// within clamp_margin of the scan as the initial pose places it (a query only ever looks one gate radius around itself;
// the margin is the room the pose has to move during the iterations).  The kernels count the queries that come within a
// cell of a cut face (LoamState.fail == 3): such a call is redone on a region twice as wide, so the result is the full
// index's whenever the call succeeds.
static constexpr double kClampMargin = 10.0;
static constexpr int kClampRetries = 6;      // 10 m ... 320 m

int set_clamp_from_scan(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats, const double pose[16]) {
    if (!n_src) return fail(h, "target bounding box too large for the dense index and the scan is empty");
    // bounding box of the scan: one index build over it (rare path), read back from its header
    if (h->src_grid.build(d_src, n_src, stride_floats, 1.0, h->stream, &h->err) != hipSuccess) return 1;
    GridHeader sh;
    H_TRY(hipMemcpyAsync(&sh, h->src_grid.header.p, sizeof sh, hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));
    if (sh.empty) return fail(h, "target bounding box too large for the dense index and the scan has no finite point");
    double lo[3], hi[3], mlo[3] = {1e300, 1e300, 1e300}, mhi[3] = {-1e300, -1e300, -1e300};
    for (int d = 0; d < 3; ++d) { lo[d] = sh.origin[d]; hi[d] = sh.origin[d] + sh.dims[d] * sh.cell; }      // a superset of the scan's box
    for (int c = 0; c < 8; ++c) {
        const double p[3] = {c & 1 ? hi[0] : lo[0], c & 2 ? hi[1] : lo[1], c & 4 ? hi[2] : lo[2]};
        for (int r = 0; r < 3; ++r) {
            const double v = pose[r] * p[0] + pose[4 + r] * p[1] + pose[8 + r] * p[2] + pose[12 + r];
            mlo[r] = std::min(mlo[r], v); mhi[r] = std::max(mhi[r], v);
        }
    }
    for (int d = 0; d < 3; ++d) {
        if (!(mlo[d] == mlo[d] && mhi[d] == mhi[d])) return fail(h, "target bounding box too large for the dense index and the initial pose is not finite");
        h->clamp.lo[d] = mlo[d] - h->clamp_margin; h->clamp.hi[d] = mhi[d] + h->clamp_margin;
    }
    h->clamp.use = 1;
    return 0;
}

// pcr_set_target on a cloud whose bounding box cannot be tabulated (a stray point kilometres away) and no scan to cut the box around:
// the box of the BULK of the cloud instead.  A strided sample of <= 4096 points comes to the host; per axis the 2nd and 98th
// percentile of the finite samples, widened by half their span + 20 m, is a region that holds every point of an ordinary map and
// leaves a stray one out.  What lies outside is not indexed -- and, as with the scan-centred cut, not silently: a face with target
// points beyond it is marked (header.cut_mask), a query whose 3x3x3 block touches such a face is counted, and the registration is
// then redone on a region cut around the scan (run_loam), so a scan that really visits the far part of the cloud is still served.
int set_clamp_from_target_sample(pcr_handle* h) {
    const size_t n = h->tgt_n, stride_b = h->tgt_stride * 4;
    if (!n || !h->tgt_ptr) return fail(h, "target bounding box too large for the dense index");
    const size_t step = std::max<size_t>(1, n / 4096), m = (n + step - 1) / step;
    std::vector<float> xyz(m * 3);
    H_TRY(hipMemcpy2DAsync(xyz.data(), 12, h->tgt_ptr, stride_b * step, 12, m, hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));
    for (int d = 0; d < 3; ++d) {
        std::vector<double> v;
        v.reserve(m);
        for (size_t i = 0; i < m; ++i) { const float a = xyz[i * 3], b = xyz[i * 3 + 1], c = xyz[i * 3 + 2]; if (std::isfinite(a) && std::isfinite(b) && std::isfinite(c)) v.push_back((double)xyz[i * 3 + d]); }
        if (v.empty()) return fail(h, "target bounding box too large for the dense index and no finite point in its sample");
        std::sort(v.begin(), v.end());
        const double q_lo = v[(size_t)(0.02 * (double)(v.size() - 1))], q_hi = v[(size_t)(0.98 * (double)(v.size() - 1) + 0.5)];
        const double pad = 0.5 * (q_hi - q_lo) + 20.0;
        h->clamp.lo[d] = q_lo - pad; h->clamp.hi[d] = q_hi + pad;
    }
    h->clamp.use = 1;
    return 0;
}

int check_grid_overflow(pcr_handle* h, int overflow, uint64_t need_cells, const float* d_src = nullptr, size_t n_src = 0, size_t stride_floats = 0,
                        const double* pose = nullptr) {
    if (!overflow) return 0;
    if (need_cells > 4000000000ull && h->method == kLoam && d_src && pose && !h->clamp.use) {
        if (set_clamp_from_scan(h, d_src, n_src, stride_floats, pose)) return 1;
        if (build_target(h, h->tgt_ptr, h->tgt_n, h->tgt_stride)) return 1;
        return 2;
    }
    if (need_cells > 4000000000ull && h->method == kLoam && h->clamp_from_bulk && !h->clamp.use) {      // pcr_set_target: no scan to go by
        if (set_clamp_from_target_sample(h)) return 1;
        if (build_target(h, h->tgt_ptr, h->tgt_n, h->tgt_stride)) return 1;
        return 2;
    }
    if (h->grid.grow_cells(need_cells, &h->err) != hipSuccess) return 1;
    if (build_target(h, h->tgt_ptr, h->tgt_n, h->tgt_stride)) return 1;
    return 2;
}

// Bring the enqueued index build to a usable state before anything else is launched: read the header back and grow the
// cell table (or cut the box around the scan) until it fits.  No collective in here -- a sharded call settles its tile
// on every rank independently and only then enters the exchange loop (a rank that retried on its own after the loop, as the
// unsharded path does, would leave the other ranks' all-reduces without a peer).
int settle_loam_index(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats, const double* pose) {
    for (int attempt = 0; attempt < 6; ++attempt) {
        GridHeader hdr;
        H_TRY(hipMemcpyAsync(&hdr, h->grid.header.p, sizeof(hdr), hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        if (hdr.stale) {      // the box taken over from the previous target does not hold this one: fresh box, padded from now on
            h->grid.hint_margin = 8; h->grid.cells_hint = 0;      // (a cloud that left the old box: its cell count is anybody's guess too)
            if (build_target(h, h->tgt_ptr, h->tgt_n, h->tgt_stride)) return 1;
            continue;
        }
        const int ov = check_grid_overflow(h, hdr.overflow, hdr.n_cells, h->clamp_allowed ? d_src : nullptr, n_src, stride_floats, pose);
        if (ov == 0) { if (!hdr.empty) { h->grid.confirm(); h->grid.note_cells(hdr.n_cells); } return 0; }      // (the header of an EMPTY target is no hint: the kernels of a build that reused it would leave at once)
        if (ov == 1) return 1;
    }
    return fail(h, "target index could not be sized");
}

int run_loam(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats, double pose[16], int* converged,
             bool index_timed) {
    if (n_src > 0xfffffff0ull) return fail(h, "source cloud too large");
    if (peer_check(h)) return 1;
    if (ensure_loam_buffers(h)) return 1;
    const int iters = std::max(0, h->prm.loam_iters);
    const bool shard = sharded(h);
    // (pcr_params.loam_clamp_margin_mm: first margin in millimetres, a test hook that makes the widening path reachable with ordinary clouds)
    // (may be negative: the scan's box is padded by two index cells already, so only a region cut INTO the scan makes queries reach its edge)
    h->clamp_margin = h->prm.loam_clamp_margin_mm != 0 ? 1e-3 * h->prm.loam_clamp_margin_mm : kClampMargin;
    const double margin_cap = kClampMargin * (1 << kClampRetries);
    for (int attempt = 0; attempt < 24; ++attempt) {
        bool rank_fail = false;
        std::string rank_err;
        if (shard) {
            // The exchange loop below must run the same number of collectives on every rank whatever happens to this rank's
            // index: settle it first, and if that fails take part with empty sums and a flag that stops all ranks together.
            if (!h->grid.valid || settle_loam_index(h, d_src, n_src, stride_floats, pose)) {
                rank_fail = true; rank_err = h->err.empty() ? "target index not built" : h->err;
                if (!h->dummy_grid.p) {
                    GridHeader dh;
                    memset(&dh, 0, sizeof dh);
                    dh.overflow = 1; dh.empty = 1; dh.cell = dh.inv_cell = 1.0; dh.n_cells = 1;
                    // (the header is followed by a few zero words that stand in for the cell table and the point array)
                    if (h->dummy_grid.reserve(sizeof dh + 1024) != hipSuccess || hipMemset(h->dummy_grid.p, 0, sizeof dh + 1024) != hipSuccess ||
                        hipMemcpy(h->dummy_grid.p, &dh, sizeof dh, hipMemcpyHostToDevice) != hipSuccess)
                        return fail(h, rank_err + " (and no memory for the stand-in header: the other ranks of this call will hang)");
                }
            }
        }
        LoamArgs a;
        fill_loam_args(h, &a, d_src, n_src, stride_floats, pose);
        if (rank_fail) {
            a.grid.hdr = h->dummy_grid.as<GridHeader>();
            a.grid.pts = reinterpret_cast<const float4*>(h->dummy_grid.as<char>() + 512);
            a.grid.cell_start = reinterpret_cast<const uint32_t*>(h->dummy_grid.as<char>() + 512);
            a.rank_fail = 1; a.nn_cache = nullptr;
        }
        if (shard) a.reduced = h->host_ar ? h->red_dev : h->loam_reduced.as<double>();
        h->result_host->pad = 0;
        if (h->profile >= 1 && !index_timed) { H_TRY(hipEventRecord(h->ev_start, h->stream)); H_TRY(hipEventRecord(h->ev_index, h->stream)); }
        const bool per_kernel = h->profile >= 2;
        if (per_kernel) {
            while ((int)h->ev_kernel.size() < 2 * iters) { hipEvent_t e; H_TRY(hipEventCreate(&e)); h->ev_kernel.push_back(e); }
        }
        // Early exit (the reference's default: LoamRegister.cpp:198-220 leaves the loop once a step is small): the device ends the loop in a launch's
        // prologue and the launches behind it leave at once -- but each still costs ~5 us of the stream, and the caller's loop converges after two
        // iterations of eight (extra.sequence).  So with early exit on, the host stays TWO launches ahead of the prologue's progress word (host-mapped)
        // instead of queueing all of them: a loop that ends after launch 2 costs four launches, not eight.  Without early exit (BASELINE's ten fixed
        // iterations), sharded, or timed per kernel: everything is queued at once, as before.
        const bool paced = a.c.early_exit != 0 && !shard && !per_kernel && iters > 3;
        volatile int32_t* const progress = &h->result_host->progress;
        *progress = -1;
        int launched = 0;
        for (int k = 0; k < iters; ++k) {
            if (paced && k >= 3) {
                const int32_t want = (int32_t)(k - 2) << 1;
                int32_t v = *progress;
                for (int spin = 0; spin < 400000 && v < want; ++spin) { __builtin_ia32_pause(); v = *progress; }      // (bounded: a stalled device gets the launch anyway)
                if (v >= 0 && (v & 1)) break;
            }
            ++launched;
            if (per_kernel) H_TRY(loam_launch_iteration(a, k, h->stream, h->ev_kernel[2 * k], h->ev_kernel[2 * k + 1]));
            else H_TRY(loam_launch_iteration(a, k, h->stream, nullptr, nullptr, !shard));
            if (h->host_ar) {
                // the caller's collective: sums to the host, through fn, back (one host round trip per linearisation)
                H_TRY(loam_launch_reduce(a, k, h->red_dev, h->stream));
                H_TRY(hipStreamSynchronize(h->stream));
                if (ranks_allreduce(h, h->red_host, kAccum)) return 1;
            } else if (h->peer_on) {
                // fold + push to every peer + fold what arrived, one launch (the sums never leave the device)
                h->peer_seq += 1.0;
                H_TRY(loam_launch_peer_exchange(a, k, h->peer, h->peer_seq, h->loam_reduced.as<double>(), h->stream));
            } else if (h->comm) {
                H_TRY(loam_launch_reduce(a, k, h->loam_reduced.as<double>(), h->stream));
                int rc = g_rccl.allreduce(h->loam_reduced.p, h->loam_reduced.p, kAccum, /*ncclFloat64*/ 8, /*ncclSum*/ 0, h->comm, h->stream);
                if (rc != 0) return fail(h, "ncclAllReduce failed with code " + std::to_string(rc));
            }
        }
        H_TRY(loam_launch_finalize(a, launched, h->stream));
        if (h->profile >= 1) H_TRY(hipEventRecord(h->ev_end, h->stream));
        // The result arrives in host-mapped memory, its completion word written last with a system-scope release: a short
        // spin on that word returns a few microseconds before the stream's completion signal wakes a sleeping thread
        // (a frontend thread is waiting for this pose anyway).  Timing with events, or a slow call, falls back to the sync.
        if (h->profile == 0) {
            volatile int32_t* flag = &h->result_host->pad;
            for (int spin = 0; spin < 200000 && *flag != 1; ++spin) __builtin_ia32_pause();
            std::atomic_thread_fence(std::memory_order_acquire);
            if (*flag != 1) H_TRY(hipStreamSynchronize(h->stream));
        } else {
            H_TRY(hipStreamSynchronize(h->stream));
        }
        const LoamResult r = *h->result_host;
        if (r.pad != 1) return fail(h, "LOAM finalize kernel did not complete");
        if (peer_check(h)) return 1;      // (an exchange of this call timed out: the ranks that waited stopped their loops -- slot 30 -- and the session is over)
        if (r.fail == 2 || rank_fail)      // (every rank sees the flag in the sums of the first linearisation: all return here together)
            return fail(h, rank_fail ? "this rank could not index its map tile: " + rank_err : "sharded scan2map: another rank could not index its map tile");
        if (r.fail == 3) {
            // some query reached a face the index was cut at: the same call again, twice the room around the scan.  Sharded: all
            // ranks see the count in the same sums and come back here together, whether or not their own tile was cut.
            h->clamp_margin = std::max(2.0 * h->clamp_margin, h->clamp_margin + 1.0);
            if (h->clamp_margin > margin_cap) return fail(h, "the pose left the region of a target too sparse for the dense index (a stray point far from the map?)");
            if (h->clamp.use) {
                if (set_clamp_from_scan(h, d_src, n_src, stride_floats, pose)) { if (!shard) return 1; h->grid.valid = false; }
                else if (build_target(h, h->tgt_ptr, h->tgt_n, h->tgt_stride)) { if (!shard) return 1; }
            }
            index_timed = false;
            continue;
        }
        if (!shard) {
            if (r.grid_stale) {      // the box taken over from the previous target does not hold this one: fresh box, padded from now on
                h->grid.hint_margin = 8; h->grid.cells_hint = 0;      // (a cloud that left the old box: its cell count is anybody's guess too)
                if (build_target(h, h->tgt_ptr, h->tgt_n, h->tgt_stride)) return 1;
                index_timed = false;
                continue;
            }
            int ov = check_grid_overflow(h, r.grid_overflow, r.grid_cells, h->clamp_allowed ? d_src : nullptr, n_src, stride_floats, pose);
            if (ov == 1) return 1;
            if (ov == 2) { index_timed = false; continue; }
            if (!h->clamp.use && !r.grid_empty) { h->grid.confirm(); h->grid.note_cells(r.grid_cells); }      // (an empty target's header is no hint)
        }
        memcpy(pose, r.pose, 16 * sizeof(double));
        if (converged) *converged = r.converged;
        h->stats.iterations = r.iters_run; h->stats.attempts = attempt + 1;
        h->stats.n_src = (int64_t)n_src; h->stats.n_dst = (int64_t)h->tgt_n;
        h->stats.kernel_ms = 0; h->stats.kernel_launches = 0;
        if (h->profile >= 1) {
            float ms = 0;
            H_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_end)); h->stats.total_ms = ms;
            H_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_index)); h->stats.index_ms = ms;
            H_TRY(hipEventElapsedTime(&ms, h->ev_index, h->ev_end)); h->stats.solve_ms = ms;
        }
        if (per_kernel) {
            // only launches that linearised count (the loop may have ended early)
            const int used = std::min(iters, r.converged || r.fail ? r.iters_run : iters);
            for (int k = 0; k < used; ++k) {
                float ms = 0;
                H_TRY(hipEventElapsedTime(&ms, h->ev_kernel[2 * k], h->ev_kernel[2 * k + 1]));
                h->stats.kernel_ms += ms; h->stats.kernel_launches++;
            }
        }
        if (h->prm.record_trace && iters > 0) {
            h->trace_host.resize(iters);
            H_TRY(hipMemcpy(h->trace_host.data(), h->loam_trace.p, (size_t)iters * sizeof(LoamTrace), hipMemcpyDeviceToHost));
            h->trace_iters = r.iters_run;
        }
        return 0;
    }
    return fail(h, "target index could not be sized");
}

// ---- host ranges page-locked by pcr_host_pin, and who copies out of them ----
// pcr_host_unpin must not unregister a range while a copy out of it is in flight.  Every upload that reads a pinned range notes (device, stream)
// with the range; unpinning waits for exactly those streams -- not for every device of the node (which created a context on each of them, waited
// for unrelated work, and switched the calling thread's device: ADVICE r4).  A handle that goes away, or is given another stream, takes its
// entries with it (after waiting for its own stream).
namespace {
struct PinUse { int device; hipStream_t stream; };
struct PinnedRange { const void* p; size_t bytes; std::vector<PinUse> uses; };
std::mutex g_pin_mu;
std::vector<PinnedRange> g_pinned;
void pin_note_copy(const void* src, size_t bytes, int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pinned.empty()) return;
    const char* a = (const char*)src;
    for (PinnedRange& r : g_pinned) {
        const char* b = (const char*)r.p;
        if (a < b + r.bytes && b < a + bytes) {      // the copy reads from the range
            bool known = false;
            for (const PinUse& u : r.uses) known = known || (u.device == device && u.stream == stream);
            if (!known) r.uses.push_back(PinUse{device, stream});
        }
    }
}
void pin_forget_stream(hipStream_t stream) {      // (the caller has waited for the stream)
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (PinnedRange& r : g_pinned)
        for (size_t i = r.uses.size(); i-- > 0;) if (r.uses[i].stream == stream) r.uses.erase(r.uses.begin() + (long)i);
}
}  // namespace

// Host clouds are uploaded verbatim.  pcr_params.host_copy_xyz = 1: of records wider than 16 bytes (pcl::PointXYZI is 32: basic.hpp:16 -- what
// the plugin adapter hands over) only the first 16 bytes cross PCIe, by a pitched copy into a staging area of the SAME stride
// (nothing downstream changes; whole_records: every field is needed -- pcr_voxel_filter).  Measured and NOT the default: the pitched copy of a
// 1 M-point map takes 3.5 ms, pageable or page-locked, where the verbatim copy of twice the bytes takes 0.6 ms.
int stage_host(pcr_handle* h, DeviceBuf* buf, const void* src, size_t n, size_t stride_bytes, const float** out, bool whole_records = false) {
    const size_t bytes = n * stride_bytes;
    H_TRY(buf->reserve(bytes ? bytes : 16));
    if (bytes) {
        if (stride_bytes > 16 && !whole_records && h->prm.host_copy_xyz == 1)
            H_TRY(hipMemcpy2DAsync(buf->p, stride_bytes, src, stride_bytes, 16, n, hipMemcpyHostToDevice, h->stream));
        else
            H_TRY(hipMemcpyAsync(buf->p, src, bytes, hipMemcpyHostToDevice, h->stream));
        pin_note_copy(src, bytes, h->device, h->stream);
    }
    *out = buf->as<float>();
    return 0;
}

int check_stride(pcr_handle* h, size_t stride_bytes) {
    if (stride_bytes < 12 || stride_bytes % 4 != 0) return fail(h, "stride_bytes must be a multiple of 4 and >= 12");
    return 0;
}

int set_device(pcr_handle* h) {
    H_TRY(hipSetDevice(h->device));
    return 0;
}

int ensure_out32(pcr_handle* h) {
    if (!h->out32_host) {
        H_TRY(hipHostMalloc((void**)&h->out32_host, 32 * sizeof(double), hipHostMallocMapped));
        memset(h->out32_host, 0, 32 * sizeof(double));
        H_TRY(hipHostGetDevicePointer((void**)&h->out32_dev, h->out32_host, 0));
    }
    H_TRY(h->vg_partials.reserve((size_t)512 * 32 * sizeof(double)));
    return 0;
}

// Wait until the kernel that was given `seq` has written it into the last word of a host-mapped result block (it does so
// after everything else, with a system-scope release).  A short spin returns ~15 us before a sleeping thread would be
// woken by the stream's completion signal, and the LM / line-search drivers wait like this a dozen times per call.
int wait_result(pcr_handle* h, const double* flag_word, double seq) {
    const volatile double* f = flag_word;
    if (h->profile == 0) {
        for (int spin = 0; spin < 200000 && *f != seq; ++spin) __builtin_ia32_pause();
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*f == seq) return 0;
    }
    H_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---------------------------------------------------------------------------------
// VGICP host driver: PCL align() + LsqRegistration (lsq_registration_impl.hpp:53-171)
// ---------------------------------------------------------------------------------
int settle_grid(pcr_handle* h, GridIndex& g, const float* d_pts, size_t n, size_t stride_floats, double cell, int pcl_mode = 0, const ClampBox* clamp = nullptr) {
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (g.build(d_pts, n, stride_floats, cell, h->stream, &h->err, 0.0, pcl_mode, clamp) != hipSuccess) return 1;
        GridHeader hdr;
        H_TRY(hipMemcpyAsync(&hdr, g.header.p, sizeof(hdr), hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        if (!hdr.overflow) { g.note_cells(hdr.n_cells); return 0; }
        if (g.grow_cells(hdr.n_cells, &h->err) != hipSuccess) return 1;
    }
    return fail(h, "index could not be sized");
}

// The coarse level pays for clouds with a long sparse tail -- raw or lightly filtered lidar scans, whose far points need
// dozens of rings on the fine grid -- and costs a little (another index build, de-duplication) on a voxel-filtered map of
// uniform density: measured 0.74 ms on one level vs 0.94 ms on three for the 1 M-point map, 7.7 ms vs 0.36 ms for the
// 65 k-point scan.  Scan-sized clouds get a second level, map-sized ones stay on one.  Two levels six cells apart (0.5 m and 3 m)
// beat the three levels four apart (0.5 / 2 / 8 m) this started with: the 8 m grid of a scan has a handful of cells holding most
// of its points -- three blocks sorted the whole scan, 120 us -- and each level is an index build on the side stream; A/B of the
// whole scan2map on one box: 3 x 4: 0.906 ms, 2 x 4: 0.917, 2 x 5: 0.890, 2 x 6: 0.875, 2 x 7: 0.900, 2 x 8: 0.880, 3 x 6: 0.975, 2 x 12: 1.25.
// (PCR_COV_LEVELS / PCR_COV_RATIO override for such runs.)
static int cov_levels_small() { static const int v = dev_env("PCR_COV_LEVELS") ? atoi(dev_env("PCR_COV_LEVELS")) : 2; return v < 1 ? 1 : (v > 3 ? 3 : v); }
static double src_cell0() { static const double v = dev_env("PCR_COV_CELL0") ? atof(dev_env("PCR_COV_CELL0")) : 1.0; return v > 0 ? v : 1.0; }      // (development: cell of a scan's own search levels, in voxels)
static double cov_ratio() { static const double v = dev_env("PCR_COV_RATIO") ? atof(dev_env("PCR_COV_RATIO")) : 6.0; return v; }
int cov_levels(size_t n) { return n <= 300000 ? cov_levels_small() : 1; }

int vgicp_source_enqueue(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats, bool marked = false, std::string* errp = nullptr);
int vgicp_side_init(pcr_handle* h) {
    if (!h->side_stream) {
        // The three streams of a VGICP call must be three HARDWARE queues.  The runtime spreads the streams of a process over a small pool
        // of queues per priority level (four by default), in the order they were created: in a process that had made a few streams
        // before -- bench.py's LOAM handles, any host application -- this handle's main and side stream came to share a queue, the scan's
        // side ran behind the target's kernels instead of beside them, and a call took 0.76 ms instead of 0.53.  Each priority level
        // has a pool of its own: the side stream asks for the highest, the auxiliary one for the lowest, the main one keeps the default.
        int prio_least = 0, prio_greatest = 0;
        H_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        H_TRY(hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, prio_greatest));
        H_TRY(hipEventCreateWithFlags(&h->ev_side_in, hipEventDisableTiming));
        H_TRY(hipEventCreateWithFlags(&h->ev_side_done, hipEventDisableTiming));
        H_TRY(hipEventCreateWithFlags(&h->ev_hdr, hipEventDisableTiming));
        H_TRY(hipEventCreateWithFlags(&h->ev_aux_in, hipEventDisableTiming));
        H_TRY(hipEventCreateWithFlags(&h->ev_aux_done, hipEventDisableTiming));
        H_TRY(hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, prio_least));
        H_TRY(hipHostMalloc((void**)&h->side_hdr, 7 * sizeof(GridHeader), hipHostMallocDefault));
    }
    return 0;
}

// (inside settle_cov_levels: work may be in flight on the auxiliary stream -- never return before it has drained)
#define H_TRY_AUX(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { if (aux) (void)hipStreamSynchronize(h->aux_stream); return fail(h, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
// the fine index plus the coarse ones of the covariance search, settled with one round trip
// after_ahead (with a grid built ahead only): what the caller would enqueue once the headers have been read, enqueued BEFORE they are -- the
// host waits for the headers alone (an event behind their copies) while the device goes on; *after_clean tells whether what was enqueued
// stands (first attempt, nothing stale or overflowing, the grid built ahead usable).  Kernels queued that way see the flags in the headers
// and leave early; whatever they wrote is written again by the caller.
int settle_cov_levels(pcr_handle* h, GridIndex& g, GridIndex& l1, GridIndex& l2, const float* d_pts, size_t n, size_t stride_floats, double cell,
                      double shift0, GridHeader* hdr0_out, bool may_cut = false, double ahead_cell = 0.0, bool* ahead_ok = nullptr,
                      const std::function<int()>* after_ahead = nullptr, bool* after_clean = nullptr, const std::function<int()>* before_wait = nullptr,
                      bool scan_levels = false, BuildFilter* filter0 = nullptr) {
    // filter0: the fine level may index the points of a region only (BuildFilter: possible when this build reuses the header and the tile layout of
    // an earlier full build of the level -- build() decides and says so in filter0->applied)
    // scan_levels: the levels of a SCAN (the source of an alignment that did not come through vgicp_source_enqueue: pcr_set_target + pcr_align,
    // pcr_vgicp_covariances, the redo path): built like vgicp_source_enqueue builds them -- one-level path, no hints: one scan's box and tile
    // layout do not hold the next (walls at other distances; measured there: every hint failed and the redo cost 0.9 ms)
    GridIndex* lv[3] = {&g, &l1, &l2};
    const double cells[3] = {cell, cov_ratio() * cell, cov_ratio() * cov_ratio() * cell};
    const int levels = cov_levels(n);
    bool todo[3] = {true, levels > 1, levels > 2};
    if (after_clean) *after_clean = false;
    if (after_ahead && vgicp_side_init(h)) return 1;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (attempt > 0 && filter0) filter0->applied = filter0->tail_applied = false;      // (a repeat is a full build)
        GridHeader hdr_stack[4];
        GridHeader* hdr = after_ahead ? h->side_hdr + 3 : hdr_stack;      // (page-locked when the host is not to block in the copies)
        // (a one-level target's covariance grid at last call's cell size, enqueued with the fine level so that one round trip
        //  settles both; whether that size still suits the density is the caller's check)
        GridHeader& hdr_ahead = hdr[3];
        const bool ahead = ahead_cell > 0.0 && levels == 1 && attempt == 0 && !(h->clamp.use && may_cut);
        // ... on a stream of its own when the caller queues its work behind both (after_ahead): the two builds read the same cloud and
        // depend on nothing of each other, and neither fills the device (a chain of two or three launches of a few hundred blocks)
        static const bool no_aux = dev_env("PCR_COV_NO_AUX") != nullptr;      // (A/B runs)
        const bool aux = ahead && after_ahead != nullptr && h->aux_stream != nullptr && !no_aux;
        auto build_ahead = [&](hipStream_t s) -> int {
            l1.no_hints = h->prm.index_no_hints != 0;
            if (l1.build(d_pts, n, stride_floats, ahead_cell, s, &h->err, 0.0, 0, nullptr, true) != hipSuccess) return 1;
            H_TRY(hipMemcpyAsync(&hdr_ahead, l1.header.p, sizeof(GridHeader), hipMemcpyDeviceToHost, s));
            return 0;
        };
        if (aux) {
            H_TRY(hipEventRecord(h->ev_aux_in, h->stream));      // (whatever brought the cloud here is on the main stream)
            H_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_aux_in, 0));
            const int rc = build_ahead(h->aux_stream);
            if (hipEventRecord(h->ev_aux_done, h->aux_stream) != hipSuccess || rc) { (void)hipStreamSynchronize(h->aux_stream); return rc ? 1 : fail(h, "hipEventRecord failed"); }
        }
        for (int l = 0; l < 3; ++l) {
            if (!todo[l]) continue;
            // (the box and the tile layout of this index's previous build serve as hints -- GridIndex::hint_ok: a sub-map changes by a key frame
            //  at a time, a scan's box in the sensor frame hardly at all; a cloud that does not fit raises header.stale and is built afresh)
            lv[l]->no_hints = h->prm.index_no_hints != 0 || scan_levels;
            if (scan_levels) { lv[l]->prefer_one_level = true; lv[l]->header_mirror = nullptr; lv[l]->twin = nullptr; }
            if (lv[l]->build(d_pts, n, stride_floats, cells[l], h->stream, &h->err, l == 0 ? shift0 : 0.0, 0, h->clamp.use && may_cut ? &h->clamp : nullptr, !scan_levels,
                             l == 0 && attempt == 0 ? filter0 : nullptr) != hipSuccess) { if (aux) (void)hipStreamSynchronize(h->aux_stream); return 1; }
            if (l == 0 && hdr0_out) H_TRY_AUX(lv[l]->enqueue_density(h->stream));
            H_TRY_AUX(hipMemcpyAsync(&hdr[l], lv[l]->header.p, sizeof(GridHeader), hipMemcpyDeviceToHost, h->stream));
        }
        if (aux) H_TRY_AUX(hipStreamWaitEvent(h->stream, h->ev_aux_done, 0));
        else if (ahead && build_ahead(h->stream)) return 1;
        const bool early = ahead && after_ahead != nullptr;
        if (early) {
            H_TRY(hipEventRecord(h->ev_hdr, h->stream));
            if ((*after_ahead)()) { (void)hipStreamSynchronize(h->stream); return 1; }
        }
        // (what the caller wants queued on OTHER streams while the host waits here: the scan's side of a scan2map call)
        if (before_wait && attempt == 0 && (*before_wait)()) { (void)hipStreamSynchronize(h->stream); return 1; }
        if (early) H_TRY(hipEventSynchronize(h->ev_hdr));
        else H_TRY(hipStreamSynchronize(h->stream));
        if (ahead_ok) *ahead_ok = false;
        if (ahead && hdr_ahead.stale) { l1.hint_margin = 8; l1.cells_hint = 0; }      // (built afresh by the caller: it checks ahead_ok)
        else if (ahead && !hdr_ahead.overflow) { l1.note_cells(hdr_ahead.n_cells); if (!hdr_ahead.empty) l1.confirm(); if (ahead_ok) *ahead_ok = true; }
        bool again = false;
        for (int l = 0; l < 3; ++l) {
            if (!todo[l]) continue;
            if (hdr[l].stale) {      // the box (or a tile's room) taken over from the previous build does not hold this cloud: fresh box, padded from now on
                lv[l]->hint_margin = 8; lv[l]->cells_hint = 0;
                again = true;
                continue;
            }
            if (hdr[l].overflow) {
                if (hdr[l].n_cells > 4000000000ull && may_cut && !h->clamp.use) {
                    // a box no dense table can hold (a stray point far from the map): index the bulk of the cloud instead, all levels alike
                    if (set_clamp_from_target_sample(h)) return 1;
                    for (int k = 0; k < 3; ++k) todo[k] = k < levels;
                    again = true;
                    break;
                }
                if (lv[l]->grow_cells(hdr[l].n_cells, &h->err) != hipSuccess) return 1;
                again = true;
            }
            else { todo[l] = false; lv[l]->note_cells(hdr[l].n_cells); if (!hdr[l].empty && !h->clamp.use) lv[l]->confirm(); if (l == 0 && hdr0_out) *hdr0_out = hdr[0]; }
        }
        if (!again) { if (after_clean) *after_clean = early && !hdr_ahead.stale && !hdr_ahead.overflow; return 0; }
    }
    return fail(h, "index could not be sized");
}

// Source side of a VGICP scan2map call (its own index levels + covariances, fast_gicp_impl.hpp:103-108) enqueued on the side
// stream BEFORE the target is prepared on the main one: the 65 k-point covariance search is latency-bound and hides under
// the target's kernels.  Speculative about the cell tables: an overflowing level makes its kernels return early, which
// vgicp_source_settle() detects from the headers and redoes in order.
// The point of the main stream the scan's side may start behind (its staging copy, if any, is on the main stream): recorded BEFORE the
// target's work is queued there, so that a source side enqueued later (vgicp_source_enqueue(.., marked)) does not wait for that work.
int vgicp_source_mark(pcr_handle* h) {
    if (vgicp_side_init(h)) return 1;
    H_TRY(hipEventRecord(h->ev_side_in, h->stream));
    return 0;
}
// errp: where messages go (the worker thread's own string while the calling thread may be writing h->err)
int vgicp_source_enqueue(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats, bool marked, std::string* errp) {
    std::string& err = errp ? *errp : h->err;
#define S_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(_e); return 1; } } while (0)
    h->side_pending = false;
    if (!marked && vgicp_side_init(h)) return 1;      // (marked: vgicp_source_mark has made the streams and events)
    if (n_src > 0xfffffff0ull) return 0;                         // run_vgicp reports it
    S_TRY(h->src_cov6.reserve((n_src + 1) * 6 * sizeof(double)));
    if (!marked) S_TRY(hipEventRecord(h->ev_side_in, h->stream));             // the scan's staging copy (if any) is on the main stream
    S_TRY(hipStreamWaitEvent(h->side_stream, h->ev_side_in, 0));
#undef S_TRY
    GridIndex* lv[3] = {&h->src_grid, &h->src_l1, &h->src_l2};
    const double cell = h->prm.vgicp_resolution, cells[3] = {cell, cov_ratio() * cell, cov_ratio() * cov_ratio() * cell};
    const int levels = cov_levels(n_src);
    // from here on kernels reading the caller's d_src may be queued on the side stream: an error must not return before they
    // have drained (the caller is free to release d_src as soon as the call has failed)
    hipError_t e = hipSuccess;
    for (int l = 0; l < levels && e == hipSuccess; ++l) {
        // (no hints here: the box and the tile layout of one scan do not hold the next -- walls at other distances, other tiles crowded;
        //  measured: every call's hint failed and the redo cost 0.9 ms)
        // (a scan's points crowd around the sensor: on the coarse level a few cells hold a third of the scan, and the tiled build leaves
        //  them to ONE block, 94-146 us in the trace; on the fine level the crowded tiles still cost 34 us where the one-level build
        //  -- histogram with ranks, scan of the table, scatter -- takes ~30 us for the whole level.  A/B: 0.571 -> 0.54 ms per scan)
        lv[l]->prefer_one_level = dev_env("PCR_COV_SCAN_TILED") == nullptr;
        // (one pass over the scan finds the box of both levels, and the headers go to the host from the kernel that makes them: a build
        //  without hints changes nothing in its header afterwards)
        lv[l]->header_mirror = &h->side_hdr[l];
        lv[l]->twin = (l == 0 && levels > 1) ? lv[1] : nullptr;
        lv[l]->twin_cell = src_cell0() * cells[1];
        e = lv[l]->build(d_src, n_src, stride_floats, src_cell0() * cells[l], h->side_stream, &err, 0.0);
        lv[l]->twin = nullptr;
        if (e == hipSuccess && !lv[l]->mirrored && (e = hipMemcpyAsync(&h->side_hdr[l], lv[l]->header.p, sizeof(GridHeader), hipMemcpyDeviceToHost, h->side_stream)) != hipSuccess)
            err = std::string("hipMemcpyAsync(side header): ") + hipGetErrorString(e);
    }
    if (e == hipSuccess && (e = vgicp_launch_cov(h->src_grid, levels > 1 ? &h->src_l1 : nullptr, levels > 2 ? &h->src_l2 : nullptr, d_src, stride_floats, n_src,
                                                 h->src_cov6.as<double>(), h->side_stream, nullptr, nullptr, &h->src_scratch,
                                                 (h->profile >= 2 && h->ev_cov[2] && n_src > 0 && n_src <= 300000) ? h->ev_cov + 2 : nullptr)) != hipSuccess)
        err = std::string("vgicp_launch_cov: ") + hipGetErrorString(e);
    if (e == hipSuccess && h->profile >= 2 && h->ev_cov[2] && n_src > 0 && n_src <= 300000 && dev_env("PCR_COV_OLD") == nullptr) h->ev_cov_src_used = true;      // (the development switch takes the old kernel, which records none of the three event pairs)
    h->fit_copied_from = nullptr;
    if (e == hipSuccess && !sharded(h) && n_src > 0) {      // the scan, kept for a later pcr_fitness() (off the critical path here)
        const size_t bytes = n_src * stride_floats * sizeof(float);
        if ((e = h->fit_src.reserve(bytes)) != hipSuccess || (e = hipMemcpyAsync(h->fit_src.p, d_src, bytes, hipMemcpyDeviceToDevice, h->side_stream)) != hipSuccess)
            err = std::string("keeping the scan for the fitness score: ") + hipGetErrorString(e);
        else h->fit_copied_from = d_src;
    }
    if (e == hipSuccess && (e = hipEventRecord(h->ev_side_done, h->side_stream)) != hipSuccess) err = std::string("hipEventRecord: ") + hipGetErrorString(e);
    if (e != hipSuccess) { (void)hipStreamSynchronize(h->side_stream); return 1; }
    h->side_pending = true; h->side_src = d_src; h->side_n = n_src; h->side_stride = stride_floats;
    return 0;
}

// Source covariances ready on return (ordered before whatever the main stream runs next).
int vgicp_source_settle(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats) {
    const int levels = cov_levels(n_src);
    if (h->side_pending && h->side_src == d_src && h->side_n == n_src && h->side_stride == stride_floats) {
        h->side_pending = false;
        H_TRY(hipEventSynchronize(h->ev_side_done));
        bool overflow = false;
        GridIndex* lv[3] = {&h->src_grid, &h->src_l1, &h->src_l2};
        for (int l = 0; l < levels; ++l) {
            overflow = overflow || h->side_hdr[l].overflow != 0 || h->side_hdr[l].stale != 0;
            if (h->side_hdr[l].stale) { lv[l]->hint_margin = 8; lv[l]->cells_hint = 0; }      // (redone below, with a fresh box)
        }
        if (!overflow) {
            for (int l = 0; l < levels; ++l) { lv[l]->note_cells(h->side_hdr[l].n_cells); if (!h->side_hdr[l].empty) lv[l]->confirm(); }
            H_TRY(hipStreamWaitEvent(h->stream, h->ev_side_done, 0));
            return 0;
        }
    } else if (h->side_pending) {
        h->side_pending = false;
        H_TRY(hipEventSynchronize(h->ev_side_done));             // never leave side work in flight behind the caller's back
    }
    if (settle_cov_levels(h, h->src_grid, h->src_l1, h->src_l2, d_src, n_src, stride_floats, h->prm.vgicp_resolution, 0.0, nullptr, false, 0.0, nullptr, nullptr, nullptr, nullptr, true)) return 1;
    H_TRY(h->src_cov6.reserve((n_src + 1) * 6 * sizeof(double)));
    H_TRY(vgicp_launch_cov(h->src_grid, levels > 1 ? &h->src_l1 : nullptr, levels > 2 ? &h->src_l2 : nullptr, d_src, stride_floats, n_src,
                           h->src_cov6.as<double>(), h->stream, nullptr, nullptr, &h->src_scratch));
    return 0;
}

// The scan a target is being prepared for (pcr_scan2map of an unsharded handle), or nullptr: everything is prepared.
struct RoiScan { const float* d_src; size_t n_src, stride_floats; const double* pose; };

// Marks the macro cells of h->grid's lattice that `scan` can reach from its initial pose and fills `view`: base_m metres in any direction
// plus kRoiPerMetre metres per metre of distance from the sensor (a rotation of 0.05 rad = 2.9 degrees moves a point 100 m away by 5 m).
// The pose goes through a Matrix4f first, as VGICP and NDT hand their guess over as one.
static constexpr double kRoiPerMetre = 0.05;
int roi_enqueue(pcr_handle* h, const RoiScan& scan, double cell, double base_m, RoiView* view) {
    int ms = 0;      // macro cell of about 2 m
    while (ms < 5 && cell * (double)(1 << ms) < 2.0 * (1.0 - 1e-9)) ++ms;
    h->roi_mshift = ms;
    const size_t bytes = h->grid.cell_capacity + 4096;      // macro cells <= cells <= capacity (a build that needs more raises header.overflow)
    for (DeviceBuf* b : {&h->roi_mark[0], &h->roi_mark[1]}) {
        const void* before = b->p;
        H_TRY(b->reserve(bytes));
        if (b->p != before) H_TRY(hipMemsetAsync(b->p, 0, b->cap, h->stream));      // the marks start out clear; every call clears the other buffer for the next
    }
    // Every call clears the OTHER mark buffer for the next one -- over the macro cells of ITS lattice only.  When the lattice changes
    // (another box, another cell count) marks of the old one would survive beyond the new one's extent as spurious region: harmless for the
    // result (the mask only grows), wasteful.  An asynchronous memset then.
    if (h->roi_cells_seen != h->grid.cells_hint) {
        for (DeviceBuf* b : {&h->roi_mark[0], &h->roi_mark[1]}) H_TRY(hipMemsetAsync(b->p, 0, b->cap, h->stream));
        h->roi_cells_seen = h->grid.cells_hint;
    }
    for (DeviceBuf* b : {&h->roi_tmp, &h->roi_mask}) H_TRY(b->reserve(bytes));      // (written in full by every call)
    H_TRY(h->roi_esc.reserve(64));
    Pose16 T;
    for (int i = 0; i < 16; ++i) T.m[i] = (double)(float)scan.pose[i];
    const int k = h->roi_idx;
    const bool rider = h->blob_pending;
    if (rider) h->blob.zero = h->roi_esc.as<uint32_t>();      // (the escape counter starts at zero with the state)
    H_TRY(roi_launch(h->grid, scan.d_src, scan.n_src, scan.stride_floats, T, ms, h->roi_mark[k].as<uint8_t>(), h->roi_mark[k ^ 1].as<uint8_t>(),
                     h->roi_tmp.as<uint8_t>(), h->roi_mask.as<uint8_t>(), base_m, kRoiPerMetre, h->stream, rider ? &h->blob : nullptr));
    if (rider) { h->blob_pending = false; h->blob_stored = true; }
    h->roi_idx ^= 1;
    view->lat = h->grid.header.as<GridHeader>(); view->mask = h->roi_mask.as<uint8_t>(); view->escapes = h->roi_esc.as<uint32_t>();
    view->mshift = ms; view->filtered = 0; view->count = prof_counters(h);
    return 0;
}
RoiView roi_view(const pcr_handle* h) {      // the region the handle's target was prepared for (the mask of the LAST roi_enqueue)
    RoiView v;
    memset(&v, 0, sizeof v);
    if (h->roi_on) { v.lat = h->grid.header.as<GridHeader>(); v.mask = h->roi_mask.as<uint8_t>(); v.escapes = h->roi_esc.as<uint32_t>(); v.mshift = h->roi_mshift; }
    return v;
}

// keep_clamp: the caller has set h->clamp (a region cut around a scan, vgicp_align_recut): index that region instead of deciding here
int vgicp_prepare_target(pcr_handle* h, const float* d_dst, size_t n_dst, size_t stride_floats, const RoiScan* roi_scan = nullptr, bool keep_clamp = false,
                         const std::function<int()>* before_wait = nullptr) {
    h->vg_target_ready = false;
    h->roi_on = false;
    const double res = h->prm.vgicp_resolution;
    if (!(res > 0)) return fail(h, "vgicp_resolution must be positive");
    if (h->prm.vgicp_k_corr != 20) return fail(h, "this build supports vgicp_k_corr = 20 (the reference's value) only");
    if (!keep_clamp) h->clamp.use = 0;
    h->tgt_ptr = d_dst; h->tgt_n = n_dst; h->tgt_stride = stride_floats;
    static const bool no_ahead = dev_env("PCR_COV_NO_AHEAD") != nullptr;      // (A/B runs)
    static const bool no_early = dev_env("PCR_COV_NO_EARLY") != nullptr;
    bool ahead_ok = false;
    const double ahead_cell = (!no_ahead && cov_levels(n_dst) == 1 && h->cov_scale_hint >= 1.3) ? res * h->cov_scale_hint : 0.0;
    CovCheck chk;
    const bool check = h->use_tile && h->have_halo;
    // prepared for one scan: covariances and voxels only where that scan can land (a cut index keeps the full preparation: its
    // escape accounting is of another kind)
    RoiView roi;
    memset(&roi, 0, sizeof roi);
    const bool want_roi = roi_scan && roi_scan->n_src > 0 && n_dst > 0 && !check;
    // region, covariances and voxels over (lattice, search grid): enqueued by settle_cov_levels behind the grid it builds ahead, before the
    // host has seen a header (the device used to idle ~45 us between the header read-back and the first of these launches), or below
    // ... and, from a handle's second call on, the voxel LATTICE holds the region's points only (BuildFilter, as NDT's: the region is marked first, on
    // the lattice of the previous full build, and the bin pass drops every point outside it -- a tenth of the map's points go through the two passes and
    // the voxel kernel; A/B on one box, round 5: preparation 0.372 -> 0.344 ms).  Needs the covariance search on a grid of its own (the grid built
    // ahead: it holds every point, and serves the fitness score), the previous full build's header and tile layout as hints (build() checks), and
    // lookups that test the mask first (vgicp.hip: vgicp_lookup, RoiView::filtered).
    static const bool no_filter = dev_env("PCR_VG_NO_FILTER") != nullptr;      // (A/B runs)
    BuildFilter bf;
    bf.enqueue_mask = [&]() -> hipError_t {
        if (roi_enqueue(h, *roi_scan, res, 1.0, &roi)) return hipErrorUnknown;      // 1 m of translation + 0.05 rad
        bf.mask = roi.mask; bf.mshift = roi.mshift;
        return hipSuccess;
    };
    auto enqueue_rest = [&](const GridIndex* cov_grid) -> int {
        h->roi_on = false;
        if (bf.applied && !h->clamp.use) {      // (the region was marked inside the lattice's build)
            roi.filtered = 1;
            h->roi_on = true;
        } else if (want_roi && !h->clamp.use) {
            if (roi_enqueue(h, *roi_scan, res, 1.0, &roi)) return 1;      // 1 m of translation + 0.05 rad
            h->roi_on = true;
        }
        H_TRY(vgicp_launch_cov(*cov_grid, cov_levels(n_dst) > 1 ? &h->cov_l1 : nullptr, cov_levels(n_dst) > 2 ? &h->cov_l2 : nullptr, d_dst, stride_floats,
                               n_dst, h->tgt_cov6.as<double>(), h->stream, check ? &chk : nullptr, h->roi_on ? &roi : nullptr, &h->tgt_scratch,
                               (h->profile >= 2 && h->ev_cov[0] && n_dst > 300000) ? h->ev_cov : nullptr));
        if (h->profile >= 2 && h->ev_cov[0] && n_dst > 300000) h->ev_cov_tgt_used = true;
        H_TRY(vgicp_launch_voxels(h->grid, h->tgt_cov6.as<double>(), h->vox.as<VgicpVoxel>(), h->stream, h->roi_on ? &roi : nullptr));
        return 0;
    };
    H_TRY(h->tgt_cov6.reserve((n_dst + 1) * 6 * sizeof(double)));
    H_TRY(h->vox.reserve((n_dst + 1) * sizeof(VgicpVoxel)));
    const std::function<int()> early = [&]() -> int { return enqueue_rest(&h->cov_l1); };
    bool early_clean = false;
    const bool try_early = want_roi && !no_early && ahead_cell > 0.0 && !h->clamp.use;
    // (may_cut: a cloud too spread out for dense tables -- a stray point kilometres off -- is indexed over its bulk.  A rank of a sharded call
    //  too: its cloud is its own, the cut is its own decision, and a scan that reaches the cut fails the call on EVERY rank, run_vgicp)
    const bool try_filter = try_early && !no_filter && h->prm.index_no_hints == 0;
    if (settle_cov_levels(h, h->grid, h->cov_l1, h->cov_l2, d_dst, n_dst, stride_floats, res, 0.5, &h->cov_hdr0, true, ahead_cell, &ahead_ok,
                          try_early ? &early : nullptr, &early_clean, before_wait, false, try_filter ? &bf : nullptr)) return 1;
    if (h->clamp.use) { ahead_ok = false; early_clean = false; }      // (the target was cut to its bulk in there: the grid built ahead covers the uncut cloud)
    h->have_target = true;
    // A map-sized cloud is searched on ONE level whose cell is sized for the 20-neighbour radius, not for the voxel
    // lattice: sum_sq / n is the occupancy of the cell a point lives in (averaged over the points); on a surface it grows
    // with cell^2, and ~10 points per cell put ~4 K candidates into the 27-cell block (measured optimum).  (0.5 m voxels over a 0.5 m-spaced
    // map: cell 1.25 m, 0.74 -> 0.50 ms for 1 M points, the extra index build included.)
    const GridIndex* cov_grid = &h->grid;
    bool kept_ahead = false;
    if (bf.applied) {      // (the density figure of a region-only lattice is the region's: the search cell of the previous call stays -- it decides how many candidates a search visits, never its result)
        if (!ahead_ok && settle_grid(h, h->cov_l1, d_dst, n_dst, stride_floats, res * h->cov_scale_hint, 0, nullptr)) return 1;      // (the grid built ahead did not stand: built again, in full)
        kept_ahead = ahead_ok; cov_grid = &h->cov_l1;
    } else if (cov_levels(n_dst) == 1 && n_dst > 0 && grid_sum_sq(h->cov_hdr0) > 0.0) {
        const double occ = grid_sum_sq(h->cov_hdr0) / (double)n_dst;
        const double scale = std::min(8.0, sqrt(10.0 / std::max(occ, 1e-3)));
        if (scale >= 1.3) {
            // the grid built ahead serves if its cell is within 15 % of what this cloud's density asks for (the cell only decides how
            // many candidates a search visits, never its result)
            const bool keep = ahead_ok && fabs(h->cov_scale_hint / scale - 1.0) <= 0.15;
            if (!keep) {
                if (settle_grid(h, h->cov_l1, d_dst, n_dst, stride_floats, res * scale, 0, h->clamp.use ? &h->clamp : nullptr)) return 1;
                h->cov_scale_hint = scale;
            }
            kept_ahead = keep;
            cov_grid = &h->cov_l1;
        } else h->cov_scale_hint = 0.0;
    } else h->cov_scale_hint = 0.0;
    if (check) {
        // sharded target (pcr_set_shard): the covariances of the points that can enter a voxel of the tile must be the whole
        // map's -- every neighbourhood of a point within one voxel of the tile has to end inside the halo
        H_TRY(h->cov_viol.reserve(16));
        H_TRY(hipMemsetAsync(h->cov_viol.p, 0, 16, h->stream));
        shard_extent(h, chk.ext_lo, chk.ext_hi);
        for (int d = 0; d < 3; ++d) { chk.chk_lo[d] = h->tile_lo[d] - res; chk.chk_hi[d] = h->tile_hi[d] + res; }
        chk.violations = h->cov_viol.as<uint32_t>();
    }
    if (!(early_clean && kept_ahead) && enqueue_rest(cov_grid)) return 1;
    if (check) {
        uint32_t viol = 0;
        H_TRY(hipMemcpyAsync(&viol, h->cov_viol.p, sizeof viol, hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        if (viol) return fail(h, std::to_string(viol) + " target points near this rank's tile have their 20 nearest neighbours reaching past the halo of " +
                                 std::to_string(h->halo) + " m: shard the map with a wider halo");
    }
    h->vg_target_ready = true;
    return 0;
}

int run_vgicp(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats, double pose[16], int* converged) {
    if (!h->vg_target_ready) return fail(h, "no target prepared");
    if (n_src > 0xfffffff0ull) return fail(h, "source cloud too large");
    if (ensure_out32(h)) return 1;
    // source covariances over the source's own index (fast_gicp_impl.hpp:103-108): already in flight when this is a
    // scan2map call, computed here otherwise
    h->fit_pending = false;
    const bool fit_side = h->fit_copied_from == d_src && d_src != nullptr;      // (the side stream copied this very scan)
    h->fit_copied_from = nullptr;
    if (vgicp_source_settle(h, d_src, n_src, stride_floats)) return 1;
    if (!sharded(h) && n_src > 0 && !fit_side) {
        H_TRY(h->fit_src.reserve(n_src * stride_floats * sizeof(float)));
        H_TRY(hipMemcpyAsync(h->fit_src.p, d_src, n_src * stride_floats * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    }
    H_TRY(h->corr_slot.reserve((n_src + 1) * sizeof(uint32_t)));
    H_TRY(h->corr_M.reserve((n_src + 1) * 6 * sizeof(double)));
    H_TRY(h->corr_slot2.reserve((n_src + 1) * sizeof(uint32_t)));
    H_TRY(h->corr_M2.reserve((n_src + 1) * 6 * sizeof(double)));
    VgicpArgs a;
    memset(&a, 0, sizeof a);      // (roi.mask = nullptr: the whole target is prepared, unless set below)
    a.src = d_src; a.n_src = (uint32_t)n_src; a.src_stride = (uint32_t)stride_floats;
    a.src_cov6 = h->src_cov6.as<double>();
    a.hdr = h->grid.header.as<GridHeader>();
    a.cell_start = h->grid.cell_start.as<uint32_t>();
    a.vox = h->vox.as<VgicpVoxel>();
    a.corr_slot = h->corr_slot.as<uint32_t>(); a.corr_M = h->corr_M.as<double>();
    a.corr_slot_next = h->corr_slot2.as<uint32_t>(); a.corr_M_next = h->corr_M2.as<double>();
    a.partials = h->vg_partials.as<double>();
    a.use_tile = h->use_tile; a.pad_ = 0;
    for (int d = 0; d < 3; ++d) { a.tile_lo[d] = h->tile_lo[d]; a.tile_hi[d] = h->tile_hi[d]; }
    a.escapes = nullptr; a.guard_cells = 0; a.pad2_ = 0;
    a.roi = roi_view(h);
    if (h->clamp.use) {      // the target index was cut to the bulk of the cloud (vgicp_prepare_target): watch where the scan goes
        H_TRY(h->cov_viol.reserve(16));
        H_TRY(hipMemsetAsync(h->cov_viol.p, 0, 16, h->stream));
        a.escapes = h->cov_viol.as<uint32_t>() + 1;
        a.guard_cells = (int)ceil(std::max(4.0, 8.0 * h->prm.vgicp_resolution) / h->prm.vgicp_resolution);      // the reach of a 20-neighbour covariance (as pcr_set_shard's halo)
    }
    const bool shard = sharded(h);      // every rank linearises its tile's share of the scan; H, b and the error are summed over the ranks

    Pose16 x0;
    for (int i = 0; i < 16; ++i) x0.m[i] = (double)(float)pose[i];     // guess handed over as Matrix4f (VgicpRegister.cpp:36)
    bool conv = false;
    h->vg_outer = h->vg_lin = h->vg_err = 0;
    // ---- device-resident loop (vgicp_opt.h): launches are enqueued ahead of the device, the host watches a progress word.  Not for
    // sharded targets (every pass's sums cross the ranks) and not when pcr_params.host_optimiser asks for the host loop below ----
    // Sharded over the peer exchange (pcr_comm_init_peer) the loop stays on the device too: an exchange launch in front of every pass
    // (vgicp.hip: vgicp_peer_exchange_kernel), and the host queues launches by a rule that gives every rank the same number of them (see run_ndt).
    const bool peer_loop = shard && h->peer_on && !h->host_ar && !h->comm;
    const bool on_device = n_src > 0 && (!shard || peer_loop) && h->prm.host_optimiser == 0 && h->prm.vgicp_max_iters > 0;
    if (h->roi_on && !on_device) return fail(h, "internal: a target prepared for one scan needs the device-resident loop");
    if (peer_loop && peer_check(h)) return 1;
    if (on_device) {
        if (!h->vg_out_host) {
            H_TRY(hipHostMalloc((void**)&h->vg_out_host, sizeof(VgOut), hipHostMallocMapped));
            memset(h->vg_out_host, 0, sizeof(VgOut));
            H_TRY(hipHostGetDevicePointer((void**)&h->vg_out_dev, h->vg_out_host, 0));
        }
        H_TRY(h->vg_ctl.reserve(2 * sizeof(VgCtl)));
        H_TRY(h->vg_partials.reserve((size_t)2 * 512 * 32 * sizeof(double)));
        a.partials = h->vg_partials.as<double>();
        VgCtl* d_ctl = h->vg_ctl.as<VgCtl>();
        VgOut* out = h->vg_out_host;
        h->seq += 1.0;
        const double seq = h->seq;
        H_TRY(vgicp_launch_ctl_init(d_ctl, x0, h->prm.vgicp_max_iters, h->prm.vgicp_lm_inner, h->prm.vgicp_lm_init_scale, h->prm.vgicp_rot_eps, h->prm.vgicp_trans_eps, h->stream,
                                    h->roi_on ? h->roi_esc.as<uint32_t>() : nullptr));
        // every outer iteration takes at most lm_inner passes, plus the first linearisation and the launch that finishes
        const long limit = (long)h->prm.vgicp_max_iters * std::max(1, h->prm.vgicp_lm_inner) + 3;
        if ((double)limit >= kProgressWindow) return fail(h, "vgicp_max_iters * vgicp_lm_inner exceeds the device loop's pass window (2^20)");
        long enq = 0;
        const volatile double* f_seq = &out->seq;
        const volatile double* f_prog = &out->progress;
        long spins = 0;
        if (peer_loop) {
            // never more than kAhead launches beyond the passes consumed, and exactly P + kAhead once the loop has finished after P passes: every
            // rank queues the same number of exchanges whatever it happens to see when (the launches beyond the end exchange nothing)
            const long kAhead = 3;
            H_TRY(h->vg_reduced.reserve(64 * sizeof(double)));
            auto launch = [&]() -> hipError_t {
                if (enq > 0) h->peer_seq += 1.0;      // (the first launch of a call has nothing to exchange)
                return vgicp_launch_pass_pro(a, d_ctl, h->vg_partials.as<double>(), h->vg_out_dev, h->stream, seq, (int)enq, &h->peer, h->peer_seq, h->vg_reduced.as<double>());
            };
            for (;;) {
                if (*f_seq == seq) break;
                const double pr = *f_prog;
                const long consumed = (pr >= seq * kProgressWindow && pr < (seq + 1.0) * kProgressWindow) ? (long)(pr - seq * kProgressWindow) : 0;
                if (enq - consumed < kAhead && enq < limit + kAhead) { H_TRY_DRAIN(launch()); ++enq; continue; }
                __builtin_ia32_pause();
                if (++spins > 400000) {
                    H_TRY(hipStreamSynchronize(h->stream));
                    if (*f_seq == seq) break;
                    if (peer_check(h)) return 1;
                    if (enq >= limit + kAhead) return fail(h, "vgicp: the optimiser did not finish within its pass budget");
                    spins = 0;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            for (const long want = (long)out->passes + kAhead; enq < want; ++enq) H_TRY_DRAIN(launch());
            if (peer_check(h)) return 1;
        } else {
        // (a pass is ~14 us, and the word that says one has begun is written ~6 us into it: with fewer than three launches ahead of
        // that word the queue runs dry while the host enqueues; a launch beyond the end costs ~5 us)
        for (; enq < 4 && enq < limit; ++enq) H_TRY_DRAIN(vgicp_launch_pass_pro(a, d_ctl, h->vg_partials.as<double>(), h->vg_out_dev, h->stream, seq, (int)enq));
        for (;;) {
            if (*f_seq == seq) break;
            const double pr = *f_prog;
            const long consumed = (pr >= seq * kProgressWindow && pr < (seq + 1.0) * kProgressWindow) ? (long)(pr - seq * kProgressWindow) : 0;
            if (enq - consumed < 3 && enq < limit) {
                H_TRY_DRAIN(vgicp_launch_pass_pro(a, d_ctl, h->vg_partials.as<double>(), h->vg_out_dev, h->stream, seq, (int)enq)); ++enq;
                continue;
            }
            __builtin_ia32_pause();
            if (++spins > 400000 || h->profile != 0) {          // a slow device (or a profiler): wait for what is queued, then look again
                H_TRY(hipStreamSynchronize(h->stream));
                if (*f_seq == seq) break;
                if (enq >= limit) return fail(h, "vgicp: the optimiser did not finish within its pass budget");      // (stream just drained)
                spins = 0;
                for (int k = 0; k < 4 && enq < limit; ++k, ++enq) H_TRY_DRAIN(vgicp_launch_pass_pro(a, d_ctl, h->vg_partials.as<double>(), h->vg_out_dev, h->stream, seq, (int)enq));
            }
        }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        // some pass looked up a voxel outside the region the target was prepared for: its sums lack that correspondence.  The caller
        // prepares the whole target and repeats the call (2).
        if (h->roi_on && out->roi_escapes > 0) { h->roi_repeats += 1; return 2; }
        x0 = out->x0; conv = out->conv != 0;
        h->vg_outer = out->outer; h->vg_lin = out->n_lin; h->vg_err = out->n_err;
        h->stats.attempts = out->passes;      // (the passes the device loop evaluated)
    }
    // ---- host-driven loop: the same state machine (vgicp_opt.h), one host round trip per pass; sharded, every pass's sums cross
    // the ranks.  The LM trial pass (vgicp_launch_error) also linearises at the trial pose: once a trial is accepted that pose IS the
    // next linearisation point, so its H, b, error and correspondences are already there (same values as a separate linearize()
    // would return) and the state's parity says which of the two correspondence buffers they are in ----
    if (!on_device) {
        VgCtl c;
        memset(&c, 0, sizeof c);
        vg_opt::ctl_init(&c, x0, h->prm.vgicp_max_iters, h->prm.vgicp_lm_inner, h->prm.vgicp_lm_init_scale, h->prm.vgicp_rot_eps, h->prm.vgicp_trans_eps);
        while (!c.done) {
            VgicpArgs ap = a;
            if (c.parity) { ap.corr_slot = a.corr_slot_next; ap.corr_M = a.corr_M_next; ap.corr_slot_next = a.corr_slot; ap.corr_M_next = a.corr_M; }
            h->seq += 1.0;
            if (c.kind == kVgPassLinearize) H_TRY(vgicp_launch_linearize(ap, c.xi, h->out32_dev, h->stream, h->seq));
            else H_TRY(vgicp_launch_error(ap, c.xi, h->out32_dev, h->stream, h->seq));
            if (wait_result(h, &h->out32_host[31], h->seq)) return 1;
            if (shard && ranks_allreduce(h, h->out32_host, 29)) return 1;
            double sums[29];
            for (int k = 0; k < 29; ++k) sums[k] = h->out32_host[k];
            vg_opt::ctl_step(&c, sums);
        }
        x0 = c.x0; conv = c.conv != 0;
        h->vg_outer = c.outer; h->vg_lin = c.n_lin; h->vg_err = c.n_err;
    }
    uint32_t esc = 0;
    if (a.escapes) {
        H_TRY(hipMemcpyAsync(&esc, a.escapes, sizeof esc, hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        // the scan reaches (the reach of a covariance of) a face the index was cut at: 3 -- the caller cuts the target around THIS scan
        // and repeats (vgicp_align_recut).  A rank of a sharded call cannot, and must not leave either (its peers would wait in their
        // next collective): the count travels with the fitness sums below and every rank fails the call.
        if (esc && !shard) { h->err = "the target is too spread out for the dense voxel tables (a stray point far from the map?) and was cut to its bulk, but the scan reaches "
                                      "the part that was left out"; return 3; }
    }
    for (int i = 0; i < 16; ++i) pose[i] = (double)(float)x0.m[i];     // final_transformation_ is a Matrix4f
    if (converged) *converged = conv ? 1 : 0;
    h->stats.iterations = h->vg_outer; h->stats.n_src = (int64_t)n_src; h->stats.n_dst = (int64_t)h->tgt_n;
    h->stats.kernel_launches = h->vg_lin + h->vg_err;
    if (!shard) {      // pcl::Registration::getFitnessScore() is evaluated when asked for (pcr_fitness), as in the reference
        for (int i = 0; i < 16; ++i) h->fit_pose[i] = pose[i];
        h->fit_n = n_src; h->fit_stride = stride_floats; h->fit_pending = true;
        h->fitness = 1.7976931348623157e308;
        return 0;
    }
    // sharded: every rank takes part in the sum, so the score is evaluated here, with the call (VgicpRegister.cpp:42-45)
    h->seq += 1.0;
    FitTile ft;
    memset(&ft, 0, sizeof ft);
    if (h->use_tile) {
        ft.use = 1;
        for (int d = 0; d < 3; ++d) { ft.lo[d] = h->tile_lo[d]; ft.hi[d] = h->tile_hi[d]; ft.ext_lo[d] = -1e300; ft.ext_hi[d] = 1e300; }
        if (h->have_halo) shard_extent(h, ft.ext_lo, ft.ext_hi);
    }
    H_TRY(fitness_launch(h->grid, d_src, n_src, stride_floats, pose, 1.7976931348623157e308, h->vg_partials.as<double>(), h->out32_dev, h->stream, h->seq,
                         h->use_tile ? &ft : nullptr));
    if (wait_result(h, &h->out32_host[31], h->seq)) return 1;
    h->out32_host[3] = (double)esc;      // (this rank's scan points that reached a cut face of its index)
    if (shard && ranks_allreduce(h, h->out32_host, 4)) return 1;
    if (h->out32_host[3] > 0) { h->err = "a rank's target is too spread out for the dense voxel tables (a stray point far from the map?) and was cut to its bulk, but the scan reaches "
                                         "the part that was left out"; return 3; }
    h->fitness = h->out32_host[1] > 0 ? h->out32_host[0] / h->out32_host[1] : 1.7976931348623157e308;
    // sharded: a source point farther from every map point than its rank's halo has its nearest neighbour on another rank; the
    // score is then not the map's and is reported as unavailable (the pose is unaffected)
    if (h->use_tile && h->out32_host[2] > 0) h->fitness = -1.0;
    h->stats.iterations = h->vg_outer; h->stats.n_src = (int64_t)n_src; h->stats.n_dst = (int64_t)h->tgt_n;
    h->stats.kernel_launches = h->vg_lin + h->vg_err;
    return 0;
}

// run_vgicp, and when the scan reached a cut face of an index that could not hold the whole target (a stray point kilometres away,
// a second cluster far off): the target cut around the scan itself -- its box at the initial pose plus the reach of a covariance plus
// room to move, doubled for as long as the scan still reaches a cut face -- and the alignment again from the same guess.  The
// reference's hash map and kd-tree serve any extent (fast_vgicp_voxel.hpp:129-156); with this a dense table does too, wherever the
// scan is.  d_dst: the target's points (the caller's buffer of a scan2map call, the staged copy of pcr_set_target, a sub-map).
int vgicp_align_recut(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats, double pose[16], int* converged) {
    double pose_in[16];
    memcpy(pose_in, pose, sizeof pose_in);
    int rc = run_vgicp(h, d_src, n_src, stride_floats, pose, converged);
    if (rc != 3) return rc;
    if (sharded(h)) return 1;      // (h->err says what happened)
    const double reach = std::max(4.0, 8.0 * h->prm.vgicp_resolution);
    h->clamp_margin = reach + kClampMargin;
    for (int attempt = 0; attempt < kClampRetries && rc == 3; ++attempt) {
        memcpy(pose, pose_in, sizeof pose_in);
        if (set_clamp_from_scan(h, d_src, n_src, stride_floats, pose_in)) return 1;
        if (vgicp_prepare_target(h, h->tgt_ptr, h->tgt_n, h->tgt_stride, nullptr, true)) return 1;
        rc = run_vgicp(h, d_src, n_src, stride_floats, pose, converged);
        h->clamp_margin *= 2.0;
    }
    if (rc == 3) return fail(h, "the pose left every region the target could be indexed over (a target too sparse for dense voxel tables and an optimiser that wanders)");
    return rc;
}

// ---------------------------------------------------------------------------------
// NDT host driver: pclomp::NormalDistributionsTransform::computeTransformation
// (ndt_omp_impl.hpp:81-171) with the More-Thuente line search (:649-932).
// ---------------------------------------------------------------------------------
// deferred: nothing is read back here -- the index is enqueued with the previous target's box as a hint and the cell table at its
// current size; whether either was wrong (header.overflow / header.stale) comes back with the result of the alignment
// (NdtOut), and the caller then prepares again the slow, checked way.  One host round trip less per scan.
int ndt_prepare_target(pcr_handle* h, const float* d_dst, size_t n_dst, size_t stride_floats, bool deferred = false, const RoiScan* roi_scan = nullptr) {
    h->nd_target_ready = false;
    h->roi_on = false;
    const double res = (double)(float)h->prm.ndt_resolution;     // resolution_ is a float (ndt_omp.h)
    if (!(res > 0)) return fail(h, "ndt_resolution must be positive");
    // the voxel lattice is VoxelGridCovariance's own (leaf index = floor(p * inverse_leaf) - min_b in float,
    // voxel_grid_covariance_omp_impl.hpp:218-220): GridHeader.pcl_mode, no pad cells
    h->grid.no_hints = h->prm.index_no_hints != 0;
    // prepared for one scan (pcr_scan2map): voxel Gaussians only where that scan can land (RoiView) -- and, when this build can take the
    // lattice and the tile layout of an earlier full build as they are, an index of the region's points only (BuildFilter)
    RoiView roi;
    memset(&roi, 0, sizeof roi);
    const bool want_roi = roi_scan && roi_scan->n_src > 0 && n_dst > 0;
    static const bool no_filter = dev_env("PCR_NDT_NO_FILTER") != nullptr;      // (A/B runs)
    BuildFilter bf;
    bf.enqueue_mask = [&]() -> hipError_t {
        if (roi_enqueue(h, *roi_scan, res, 1.0 + res, &roi)) return hipErrorUnknown;      // 1 m of translation + the DIRECT7 face voxels, + 0.05 rad
        bf.mask = roi.mask; bf.mshift = roi.mshift;
        return hipSuccess;
    };
    // setMinPointPerVoxel (pclomp/voxel_grid_covariance_omp.h:229-240): "Covariance calculation requires at least 3 points"
    const int min_points = std::max(3, h->prm.ndt_min_points);
    const size_t max_vox = n_dst / (size_t)min_points + 2;      // a voxel needs min_points points
    H_TRY(h->nd_vox.reserve(max_vox * sizeof(NdtVoxel)));
    H_TRY(h->nd_list.reserve((max_vox + 64) * sizeof(uint32_t)));
    if (!h->nd_count.p) {      // two counters used alternately; each call leaves the other one cleared for the next (ndt_candidates_kernel)
        H_TRY(h->nd_count.reserve(256));
        H_TRY(hipMemsetAsync(h->nd_count.p, 0, 256, h->stream));
    }
    h->nd_count_idx ^= 1;
    uint32_t* const nd_count = h->nd_count.as<uint32_t>() + 32 * h->nd_count_idx;
    uint32_t* const nd_count_next = h->nd_count.as<uint32_t>() + 32 * (h->nd_count_idx ^ 1);
    if (deferred) {
        // (the cell table keeps its size in a deferred build, so the slots can be sized before it: a region-only build lists the voxel cells
        //  and writes the slots in its tile pass -- TileTail)
        if (h->grid.ensure_tables(h->stream, &h->err) != hipSuccess) return 1;
        H_TRY(h->nd_slot.reserve(((size_t)h->grid.cell_capacity + 64) * sizeof(uint32_t)));
        bf.want_tail = true;
        bf.tail.vox_slot = h->nd_slot.as<uint32_t>(); bf.tail.list = h->nd_list.as<uint32_t>(); bf.tail.count = nd_count; bf.tail.count_next = nd_count_next;
        bf.tail.min_points = min_points; bf.tail.capacity = (uint32_t)std::min<size_t>(max_vox, 0xffffffffu);
        const size_t cap_before = h->grid.cell_capacity;
        const void* const slot_before = h->nd_slot.p;
        if (h->grid.build(d_dst, n_dst, stride_floats, res, h->stream, &h->err, 0.0, 1, nullptr, true, want_roi && !no_filter ? &bf : nullptr) != hipSuccess) return 1;
        // (the tile pass has been handed vox_slot and the list BEFORE the build: nothing in build() may move them -- it never grows the cell table, only
        //  callers do -- but a change that broke that would write through a freed pointer: refuse loudly instead)
        if (bf.tail_applied && (h->grid.cell_capacity != cap_before || h->nd_slot.p != slot_before))
            return fail(h, "internal: the cell table changed size inside a deferred NDT build whose tile pass lists the voxel cells");
    } else if (settle_grid(h, h->grid, d_dst, n_dst, stride_floats, res, 1)) return 1;
    h->tgt_ptr = d_dst; h->tgt_n = n_dst; h->tgt_stride = stride_floats; h->have_target = true;
    H_TRY(h->nd_slot.reserve(((size_t)h->grid.cell_capacity + 64) * sizeof(uint32_t)));
    if (bf.applied) { roi.filtered = 1; h->roi_on = true; }
    else if (want_roi) {
        if (roi_enqueue(h, *roi_scan, res, 1.0 + res, &roi)) return 1;
        h->roi_on = true;
    }
    H_TRY(ndt_launch_voxels(h->grid, h->nd_slot.as<uint32_t>(), h->nd_vox.as<NdtVoxel>(), nd_count, nd_count_next, h->nd_list.as<uint32_t>(),
                            max_vox, min_points, 0.01, h->stream, h->roi_on ? &roi : nullptr, bf.tail_applied));
    h->nd_target_ready = true;
    return 0;
}

namespace ndt_host {
using namespace ndt_opt;      // pose_from_p, angle_derivatives, svd6_solve, the line search: shared with the device (ndt_opt.h)
// Matrix3f::eulerAngles(0, 1, 2)
void euler_xyz(const float R[9], float out[3]) {
    const float pi = 3.14159265358979323846f;
    float r0 = atan2f(R[1 * 3 + 2], R[2 * 3 + 2]);
    const float c2 = sqrtf(R[0] * R[0] + R[1] * R[1]);
    float r1;
    if (r0 > 0.f) { r0 -= pi; r1 = atan2f(-R[2], -c2); }
    else r1 = atan2f(-R[2], c2);
    const float s1 = sinf(r0), c1 = cosf(r0);
    const float r2 = atan2f(s1 * R[2 * 3 + 0] - c1 * R[1 * 3 + 0], c1 * R[1 * 3 + 1] - s1 * R[2 * 3 + 1]);
    out[0] = -r0; out[1] = -r1; out[2] = -r2;
}
}  // namespace ndt_host

struct NdtRun {
    pcr_handle* h; NdtArgs a; NdtPose T; NdtAngles ang;
};
// the guess as the optimiser starts from it: handed over as Matrix4f (NdtRegister.cpp:27), its Euler angles as the parameters
void ndt_initial_pose(const double pose[16], NdtPose* T0, double p0[6]) {
    float G[16];
    for (int i = 0; i < 16; ++i) G[i] = (float)pose[i];
    for (int rr = 0; rr < 3; ++rr) { for (int c = 0; c < 3; ++c) T0->R[rr * 3 + c] = G[c * 4 + rr]; T0->t[rr] = G[12 + rr]; }
    float eul[3];
    ndt_host::euler_xyz(T0->R, eul);     // Transform::rotation() taken as the linear part
    p0[0] = T0->t[0]; p0[1] = T0->t[1]; p0[2] = T0->t[2]; p0[3] = eul[0]; p0[4] = eul[1]; p0[5] = eul[2];
}

// computeDerivatives at parameters p with the cloud transformed by T: score, gradient, Hessian
int ndt_derivatives(NdtRun* r, const double p[6], bool compute_hessian, double* score, double grad[6], double hess[36]) {
    pcr_handle* h = r->h;
    ndt_host::angle_derivatives(p, &r->ang);
    if (r->a.n_src == 0) { *score = 0; memset(grad, 0, 6 * sizeof(double)); memset(hess, 0, 36 * sizeof(double)); return 0; }
    h->seq += 1.0;
    H_TRY(ndt_launch_derivatives(r->a, r->T, r->ang, compute_hessian ? 1 : 0, h->out48_dev, h->stream, h->seq));
    if (wait_result(h, &h->out48_host[47], h->seq)) return 1;
    if (sharded(h) && ranks_allreduce(h, h->out48_host, 43)) return 1;      // score, gradient, Hessian summed over the ranks' tiles
    ++h->nd_deriv;
    *score = h->out48_host[0];
    for (int i = 0; i < 6; ++i) grad[i] = h->out48_host[1 + i];
    for (int i = 0; i < 36; ++i) hess[i] = compute_hessian ? h->out48_host[7 + i] : 0.0;
    return 0;
}

// One evaluation pass of the host-driven loop: the kernel the controller asked for, the fold, the ranks' sum.
int ndt_host_pass(NdtRun* r, const NdtCtl& c, double sums[43]) {
    pcr_handle* h = r->h;
    h->seq += 1.0;
    if (c.kind == kNdtPassHessian) H_TRY(ndt_launch_hessian(r->a, c.T, c.ang, h->out48_dev, h->stream, h->seq));
    else H_TRY(ndt_launch_derivatives(r->a, c.T, c.ang, c.kind == kNdtPassDerivH ? 1 : 0, h->out48_dev, h->stream, h->seq));
    if (wait_result(h, &h->out48_host[47], h->seq)) return 1;
    if (sharded(h) && ranks_allreduce(h, h->out48_host, 43)) return 1;      // score, gradient, Hessian summed over the ranks' tiles
    for (int i = 0; i < 43; ++i) sums[i] = h->out48_host[i];
    return 0;
}

int run_ndt(pcr_handle* h, const float* d_src, size_t n_src, size_t stride_floats, double pose[16], int* converged) {
    using namespace ndt_host;
    if (!h->nd_target_ready) return fail(h, "no target prepared");
    if (n_src > 0xfffffff0ull) return fail(h, "source cloud too large");
    if (!h->out48_host) {
        H_TRY(hipHostMalloc((void**)&h->out48_host, 48 * sizeof(double), hipHostMallocMapped));
        memset(h->out48_host, 0, 48 * sizeof(double));
        H_TRY(hipHostGetDevicePointer((void**)&h->out48_dev, h->out48_host, 0));
    }
    if (!h->nd_out_host) {
        H_TRY(hipHostMalloc((void**)&h->nd_out_host, sizeof(NdtOut), hipHostMallocMapped));
        memset(h->nd_out_host, 0, sizeof(NdtOut));
        H_TRY(hipHostGetDevicePointer((void**)&h->nd_out_dev, h->nd_out_host, 0));
    }
    H_TRY(h->nd_partials.reserve((size_t)1024 * 48 * sizeof(double)));      // (one buffer of 1024 rows, or the two of 256 of the one-launch-per-pass loop)
    H_TRY(h->nd_ctl.reserve(2 * sizeof(NdtCtl)));
    NdtRun r;
    memset(&r.a, 0, sizeof r.a);
    r.h = h;
    r.a.src = d_src; r.a.n_src = (uint32_t)n_src; r.a.src_stride = (uint32_t)stride_floats;
    r.a.hdr = h->grid.header.as<GridHeader>(); r.a.vox_slot = h->nd_slot.as<uint32_t>(); r.a.vox = h->nd_vox.as<NdtVoxel>();
    r.a.partials = h->nd_partials.as<double>();
    r.a.use_tile = h->use_tile; r.a.pad_ = 0;
    for (int d = 0; d < 3; ++d) { r.a.tile_lo[d] = h->tile_lo[d]; r.a.tile_hi[d] = h->tile_hi[d]; }
    r.a.roi_escapes = h->roi_on ? h->roi_esc.as<uint32_t>() : nullptr;
    r.a.pair_count = prof_counters(h);
    {   // Gauss constants (ndt_omp_impl.hpp:86-93)
        const double res = (double)(float)h->prm.ndt_resolution;
        const double c1 = 10 * (1 - h->prm.ndt_outlier_ratio), c2 = h->prm.ndt_outlier_ratio / pow(res, 3);
        const double d3 = -log(c2);
        r.a.d1 = -log(c1 + c2) - d3;
        r.a.d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / r.a.d1);
    }
    h->nd_iters = h->nd_deriv = h->nd_hess = 0;
    NdtPose T0;
    double p0[6];
    ndt_initial_pose(pose, &T0, p0);

    NdtPose final_T = T0;
    int conv = 0, nr_it = 0;
    double score = 0;
    // ---- device-resident loop: passes are enqueued ahead of the device, the host watches a progress word.  Not for sharded
    // targets (every pass's sums cross the ranks through the host) and not when pcr_params.host_optimiser asks for the host loop ----
    // Sharded over RCCL the loop stays on the device as well: fold -> ncclAllReduce on the handle's stream -> controller step, in
    // batches of a fixed number of passes -- whether another batch is due is decided from the controller state the ranks share, so
    // every rank enqueues the same collectives.  (A host-supplied collective needs the host in every pass: the host loop below.)
    // Sharded over the peer exchange (pcr_comm_init_peer) likewise, two launches per pass: the evaluation, then fold + exchange + controller step
    // in one (ndt.hip: ndt_fold_exchange_ctl_kernel).
    const bool dev_sharded = sharded(h) && (h->comm || h->peer_on) && !h->host_ar && n_src > 0 && h->prm.host_optimiser == 0;
    bool on_device = n_src > 0 && !sharded(h) && h->prm.host_optimiser == 0;
    if (h->roi_on && !on_device) return fail(h, "internal: a target prepared for one scan needs the device-resident loop");
    h->nd_grid_checked = false; h->nd_grid_bad = false;
    bool sharded_done = false;
    if (dev_sharded) {
        NdtCtl* d_ctl = h->nd_ctl.as<NdtCtl>();
        NdtOut* out = h->nd_out_host;
        H_TRY(h->nd_sums.reserve(64 * sizeof(double)));
        h->seq += 1.0;
        const double seq = h->seq;
        H_TRY(ndt_launch_ctl_init(d_ctl, T0, p0, h->prm.ndt_step_size, h->prm.ndt_trans_eps, h->prm.ndt_max_iters, h->stream, h->prm.ndt_evaluate_repeats));
        const int limit = (h->prm.ndt_max_iters + 3) * 13 + 4, kBatch = 6;
        const volatile double* f_batch = &out->batch;
        if (h->peer_on) {
            // Peer exchange: passes are queued a FIXED number ahead of the progress word, one at a time, like the unsharded loop -- with a rule that
            // makes every rank queue the same number of them whatever it happens to see when: never more than kAhead beyond the passes consumed,
            // and, once the loop has finished after P passes, exactly P + kAhead (a rank that saw the end early tops its queue up; the launches
            // beyond the end exchange nothing and leave).  The ranks' sequence numbers -- one per queued launch -- then agree in the next call too.
            const int kAhead = 3;
            const volatile double* f_seq = &out->seq;
            const volatile double* f_prog = &out->progress;
            if ((double)limit >= kProgressWindow) return fail(h, "ndt_max_iters exceeds the device loop's pass window (2^20 passes)");
            auto launch = [&]() -> hipError_t { h->peer_seq += 1.0; return ndt_launch_pass_peer(r.a, d_ctl, h->peer, h->peer_seq, h->nd_out_dev, h->stream, seq, 0); };
            int enq = 0;
            long spins = 0;
            for (;;) {
                if (*f_seq == seq) break;
                const double pr = *f_prog;
                const int consumed = (pr >= seq * kProgressWindow && pr < (seq + 1.0) * kProgressWindow) ? (int)(pr - seq * kProgressWindow) : 0;
                if (enq - consumed < kAhead && enq < limit + kAhead) { H_TRY(launch()); ++enq; continue; }
                __builtin_ia32_pause();
                if (++spins > 400000) {      // a slow device (or a profiler): wait for what is queued, then look again
                    H_TRY(hipStreamSynchronize(h->stream));
                    if (*f_seq == seq) break;
                    if (peer_check(h)) return 1;
                    if (enq >= limit + kAhead) return fail(h, "ndt: the optimiser did not finish within its pass budget");
                    spins = 0;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            for (const int want = out->passes + kAhead; enq < want; ++enq) H_TRY(launch());      // (every rank ends the call with the same number of launches queued)
            if (peer_check(h)) return 1;
        } else
        for (int batch = 1, enq = 0;; ++batch) {
            for (int b = 0; b < kBatch; ++b, ++enq) {
                H_TRY(ndt_launch_pass_fold(r.a, d_ctl, h->nd_sums.as<double>(), h->stream));      // (a rank with an empty scan still folds zeros and takes part)
                const int rc = g_rccl.allreduce(h->nd_sums.p, h->nd_sums.p, 48, /*ncclFloat64*/ 8, /*ncclSum*/ 0, h->comm, h->stream);
                if (rc != 0) return fail(h, "ncclAllReduce failed with code " + std::to_string(rc));
                H_TRY(ndt_launch_ctl(r.a, d_ctl, h->nd_sums.as<double>(), h->nd_out_dev, h->stream, seq, b == kBatch - 1 ? batch : 0));
            }
            const double want0 = seq * 65536.0 + 2.0 * batch;
            long spins = 0;
            while (!(*f_batch == want0 || *f_batch == want0 + 1.0)) {
                __builtin_ia32_pause();
                if (++spins > 400000 || h->profile != 0) { H_TRY(hipStreamSynchronize(h->stream)); if (!(*f_batch == want0 || *f_batch == want0 + 1.0)) return fail(h, "ndt: the sharded device loop lost its batch marker"); }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (peer_check(h)) return 1;      // (an exchange of this batch timed out: the loop has stopped on the ranks that waited, the session is over)
            if (*f_batch == want0 + 1.0) break;
            if (enq >= limit) return fail(h, "ndt: the optimiser did not finish within its pass budget");
        }
        if (!out->bail) {      // (a nearly singular Newton system: every rank saw the same pivots and goes to the host loop below, with the SVD)
            final_T = out->final_T; conv = out->conv; nr_it = out->nr_it; score = out->score;
            h->nd_deriv = out->n_deriv; h->nd_hess = out->n_hess; h->nd_last_passes = out->passes;
            sharded_done = true;
        }
    }
    if (on_device) {
        NdtCtl* d_ctl = h->nd_ctl.as<NdtCtl>();
        NdtOut* out = h->nd_out_host;
        h->seq += 1.0;
        const double seq = h->seq;
        // (a scan2map call whose region was marked has stored this state with the mark pass: do_scan2map, BlobStore)
        const bool stored = h->blob_stored && h->roi_on;
        h->blob_stored = false;
        if (!stored) H_TRY(ndt_launch_ctl_init(d_ctl, T0, p0, h->prm.ndt_step_size, h->prm.ndt_trans_eps, h->prm.ndt_max_iters, h->stream, h->prm.ndt_evaluate_repeats,
                                               r.a.roi_escapes));
        const int limit = (h->prm.ndt_max_iters + 3) * 13 + 5;        // an iteration takes at most 1 + 10 + 1 passes; one launch more finishes
        if ((double)limit >= kProgressWindow) return fail(h, "ndt_max_iters exceeds the device loop's pass window (2^20 passes)");
        int enq = 0;
        const int first = 3;      // (the host enqueues a pass in a quarter of the time the device needs for one: it only has to stay two ahead)
        static const bool two_launches = dev_env("PCR_NDT_TWO_LAUNCHES") != nullptr;      // the round's earlier form (pass kernel + fold/controller kernel), for A/B runs
        auto launch = [&](int index) -> hipError_t {
            if (two_launches) return ndt_launch_pass(r.a, d_ctl, h->nd_out_dev, h->stream, seq);
            if (h->profile >= 2) {      // events at every launch's own begin and end (pcr_stats.kernel_ms)
                while ((int)h->ev_kernel.size() < 2 * (index + 1)) { hipEvent_t e; hipError_t er = hipEventCreate(&e); if (er != hipSuccess) return er; h->ev_kernel.push_back(e); }
                h->nd_prof_launches = index + 1;
                return ndt_launch_pass_pro(r.a, d_ctl, h->nd_partials.as<double>(), h->nd_out_dev, h->stream, seq, index, h->ev_kernel[2 * index], h->ev_kernel[2 * index + 1]);
            }
            return ndt_launch_pass_pro(r.a, d_ctl, h->nd_partials.as<double>(), h->nd_out_dev, h->stream, seq, index);
        };
        for (; enq < first; ++enq) H_TRY_DRAIN(launch(enq));
        const volatile double* f_seq = &out->seq;
        const volatile double* f_prog = &out->progress;
        long spins = 0;
        bool finished = false;
        while (!finished) {
            if (*f_seq == seq) { finished = true; break; }
            const double pr = *f_prog;
            const int consumed = (pr >= seq * kProgressWindow && pr < (seq + 1.0) * kProgressWindow) ? (int)(pr - seq * kProgressWindow) : 0;
            if (enq - consumed < 2 && enq < limit) {       // two passes ahead of the device: a pass enqueued beyond the end costs ~10 us of device time
                H_TRY_DRAIN(launch(enq)); ++enq;
                continue;
            }
            __builtin_ia32_pause();
            if (++spins > 400000 || h->profile != 0) {          // a slow device (or a profiler): wait for what is queued, then look again
                H_TRY(hipStreamSynchronize(h->stream));
                if (*f_seq == seq) { finished = true; break; }
                if (enq >= limit) return fail(h, "ndt: the optimiser did not finish within its pass budget");
                spins = 0;
                for (int k = 0; k < 4 && enq < limit; ++k, ++enq) H_TRY_DRAIN(launch(enq));
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        h->nd_last_passes = out->passes;
        if (dev_env("PCR_NDT_TICKS")) fprintf(stderr, "ndt passes %d: fold %.2f us/pass, controller %.2f us/pass (decide %.2f, tables %.2f)\n", out->passes, out->ticks[0] * 0.01 / std::max(1, out->passes), out->ticks[1] * 0.01 / std::max(1, out->passes), out->ticks[2] * 0.01 / std::max(1, out->passes), out->ticks[3] * 0.01 / std::max(1, out->passes));
        h->nd_grid_bad = out->grid_overflow || out->grid_stale;
        h->nd_grid_empty = out->grid_empty != 0;
        h->nd_grid_cells = out->grid_cells;
        h->nd_grid_checked = true;
        if (h->nd_grid_bad) return 0;           // the caller prepares the target again and repeats the call
        // a pass looked up a voxel the target was not prepared for (RoiView), or the loop wants the host's SVD: whole target, again
        if (h->roi_on && (out->roi_escapes > 0 || out->bail)) { h->roi_repeats += 1; return 2; }
        if (out->bail) on_device = false;       // the Newton system was (nearly) singular: the host loop below decides, with the SVD
        else {
            final_T = out->final_T; conv = out->conv; nr_it = out->nr_it; score = out->score;
            h->nd_deriv = out->n_deriv; h->nd_hess = out->n_hess;
        }
    }
    if (!on_device && !sharded_done) {
        NdtCtl c;
        ctl_init(&c, T0, p0, h->prm.ndt_step_size, h->prm.ndt_trans_eps, h->prm.ndt_max_iters);
        c.replay_off = h->prm.ndt_evaluate_repeats ? 1 : 0;
        double sums[43];
        while (!c.done) {
            if (n_src == 0) memset(sums, 0, sizeof sums);        // an empty scan: computeDerivatives sums nothing
            else if (ndt_host_pass(&r, c, sums)) return 1;
            ctl_step(&c, sums);
        }
        final_T = c.final_T; conv = c.conv; nr_it = c.nr_it; score = c.score;
        h->nd_deriv = c.n_deriv; h->nd_hess = c.n_hess;
    }
    for (int i = 0; i < 16; ++i) pose[i] = 0;
    for (int rr = 0; rr < 3; ++rr) { for (int c = 0; c < 3; ++c) pose[c * 4 + rr] = (double)final_T.R[rr * 3 + c]; pose[12 + rr] = (double)final_T.t[rr]; }
    pose[15] = 1.0;
    if (converged) *converged = conv ? 1 : 0;
    h->nd_iters = nr_it; h->nd_score = score;
    h->stats.iterations = nr_it; h->stats.kernel_launches = h->nd_deriv + h->nd_hess;
    h->stats.attempts = (on_device || sharded_done) ? h->nd_last_passes : 0;      // (the passes a device loop actually evaluated; 0: not such a loop)
    h->stats.n_src = (int64_t)n_src; h->stats.n_dst = (int64_t)h->tgt_n;
    return 0;
}

// Sharded calls: before any rank enters an exchange loop every rank must know that ALL ranks have a usable target (a rank
// returning early would leave the others' collectives without a peer).  One MAX over the ranks of a status word.
int agree_prepared(pcr_handle* h, int rc_local) {
    if (!sharded(h)) return rc_local;
    const std::string err = h->err;
    double flag = rc_local ? 1.0 : 0.0;
    if (ranks_allreduce(h, &flag, 1, 1)) return 1;
    if (rc_local) { h->err = err; return 1; }
    if (flag != 0.0) return fail(h, "sharded call: another rank could not prepare its map tile");
    return 0;
}

// a new registration starts: whatever pcr_fitness() could evaluate belongs to the previous one (a call that fails during
// target preparation must not leave the new scan's points paired with the old pose)
void drop_fitness_state(pcr_handle* h) { h->fit_pending = false; h->fit_n = 0; h->fit_copied_from = nullptr; h->fitness = -1.0; }

int do_scan2map(pcr_handle* h, const void* src, size_t n_src, const void* dst, size_t n_dst, size_t stride_bytes,
                double pose[16], int* converged, bool on_device) {
    if (!h) return 1;
    h->err.clear();
    drop_fitness_state(h);
    h->map_id = 0; h->map_gen = 0;      // whatever target structures exist after this call were not built from a pcr_map generation
    if (!pose) return fail(h, "pose_inout is NULL");
    if ((n_src && !src) || (n_dst && !dst)) return fail(h, "NULL cloud with nonzero size");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    const float *d_src, *d_dst;
    if (h->method != kLoam && prof_begin(h)) return 1;
    if (h->profile >= 1) H_TRY(hipEventRecord(h->ev_start, h->stream));
    if (on_device) { d_src = (const float*)src; d_dst = (const float*)dst; }
    else {
        if (stage_host(h, &h->tgt_stage, dst, n_dst, stride_bytes, &d_dst)) return 1;
        if (stage_host(h, &h->src_stage, src, n_src, stride_bytes, &d_src)) return 1;
    }
    if (h->method == kNdt) {
        // NdtRegister::scan2Map calls setInputTarget every time, which rebuilds the voxel grid (NdtRegister.cpp:23)
        // Unsharded: the index is enqueued unchecked (previous box as a hint, cell table as it is) and the alignment's own result
        // says whether that held; if not, once more with the checked build.
        const bool try_deferred = !sharded(h) && n_src > 0 && h->prm.host_optimiser == 0;
        double pose_in[16];
        memcpy(pose_in, pose, sizeof pose_in);
        // (unsharded, device loop: the voxel Gaussians are prepared only where this scan can land -- RoiView; a call that leaves the
        //  region is repeated on the whole target)
        bool use_roi = try_deferred && h->prm.full_target == 0;
        const RoiScan rs{d_src, n_src, stride_bytes / 4, pose_in};
        for (int attempt = 0; attempt < 3; ++attempt) {
            const bool deferred = try_deferred && attempt == 0;
            h->blob_pending = h->blob_stored = false;
            if (use_roi) {      // the optimiser's initial state rides on the launch that marks the region (run_ndt then starts without a launch for it)
                NdtPose T0;
                double p0[6];
                ndt_initial_pose(pose, &T0, p0);
                H_TRY(h->nd_ctl.reserve(2 * sizeof(NdtCtl)));
                ndt_ctl_init_blob(&h->blob, h->nd_ctl.as<NdtCtl>(), T0, p0, h->prm.ndt_step_size, h->prm.ndt_trans_eps, h->prm.ndt_max_iters, h->prm.ndt_evaluate_repeats);
                h->blob_pending = true;
            }
            const int prep_rc = ndt_prepare_target(h, d_dst, n_dst, stride_bytes / 4, deferred, use_roi ? &rs : nullptr);
            h->blob_pending = false;
            if (agree_prepared(h, prep_rc)) { h->blob_stored = false; return 1; }
            if (h->profile >= 1) H_TRY(hipEventRecord(h->ev_index, h->stream));
            const int rrc = run_ndt(h, d_src, n_src, stride_bytes / 4, pose, converged);
            h->blob_stored = false;
            if (rrc == 2) { use_roi = false; memcpy(pose, pose_in, sizeof pose_in); if (deferred && h->nd_grid_checked && !h->nd_grid_bad && !h->nd_grid_empty) { h->grid.confirm(); h->grid.note_cells(h->nd_grid_cells); } continue; }
            if (rrc) return 1;
            if (!deferred) break;
            if (h->nd_grid_checked && !h->nd_grid_bad) { if (!h->nd_grid_empty) { h->grid.confirm(); h->grid.note_cells(h->nd_grid_cells); } break; }
            // the hint or the table size did not hold (or the device loop handed over to the host before it could tell): checked build
            h->grid.hint_margin = 8; h->grid.cells_hint = 0;      // (a cloud that left the old box: its cell count is anybody's guess too)
            memcpy(pose, pose_in, sizeof pose_in);
        }
        if (h->profile >= 1) {
            H_TRY(hipEventRecord(h->ev_end, h->stream));
            H_TRY(hipEventSynchronize(h->ev_end));
            float ms = 0;
            H_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_end)); h->stats.total_ms = ms;
            H_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_index)); h->stats.index_ms = ms;
            H_TRY(hipEventElapsedTime(&ms, h->ev_index, h->ev_end)); h->stats.solve_ms = ms;
            if (prof_end(h)) return 1;
        }
        return 0;
    }
    if (h->method == kVgicp) {
        // the reference keeps its target structures while the cloud POINTER is unchanged and goes stale
        // when the cloud is edited in place; this entry point always rebuilds them
        // (source side first: its ~15 launches run on the side stream while the host is still queueing the target's.  Measured the other
        //  way round -- target builds queued first, source side enqueued while the host waits for them -- 1.17 instead of 0.93 ms: the
        //  two covariance kernels then run side by side for their whole length and slow each other down)
        // The target exists for this scan only, so only the part of it the scan can reach is prepared (RoiView): the covariances of a
        // 1 M-point map were 0.4 of a 0.85 ms call, of which the scan looks up a tenth.  Should the optimiser carry the scan out of that
        // region, the whole target is prepared and the call repeated from the same guess.
        const bool use_roi = !sharded(h) && n_src > 0 && h->prm.host_optimiser == 0 && h->prm.vgicp_max_iters > 0 && h->prm.full_target == 0;
        double pose_in[16];
        memcpy(pose_in, pose, sizeof pose_in);
        const RoiScan rs{d_src, n_src, stride_bytes / 4, pose_in};
        // Order of the two sides.  The scan's own covariances used to be the longest thing in a call (344 us) and were queued first; since
        // they are searched in two classes (cov_search.hip: ~190 us with the scan's index levels) the target is the long pole, and the
        // dozen launches of the scan's side cost the host ~100 us during which the main stream sat empty.  Now the target's builds are
        // queued first and the scan's side while the host waits for their headers (settle_cov_levels: before_wait).
        // (Measured and not kept: the scan's side queued by a host thread of its own at the same time as the target's -- 0.481 / 0.495 against
        //  0.496 / 0.494 ms: with both sides on the device from the start the call is bound by the device's work, not by the host's launches.)
        static const bool src_first = dev_env("PCR_VG_SRC_FIRST") != nullptr;      // (development builds: the old order, for A/B runs)
        int prc = 0;
        bool src_queued = false;
        const std::function<int()> queue_src = [&]() -> int { src_queued = true; return vgicp_source_enqueue(h, d_src, n_src, stride_bytes / 4, true); };
        if (src_first) { prc = vgicp_source_enqueue(h, d_src, n_src, stride_bytes / 4); src_queued = true; }
        else prc = vgicp_source_mark(h);
        if (!prc) prc = vgicp_prepare_target(h, d_dst, n_dst, stride_bytes / 4, use_roi ? &rs : nullptr, false, src_first ? nullptr : &queue_src);
        if (!prc && !src_queued) prc = queue_src();
        if (prc && h->side_pending) { (void)hipEventSynchronize(h->ev_side_done); h->side_pending = false; }
        if (agree_prepared(h, prc)) {
            if (h->side_pending) { (void)hipEventSynchronize(h->ev_side_done); h->side_pending = false; }
            return 1;
        }
        if (h->profile >= 1) H_TRY(hipEventRecord(h->ev_index, h->stream));
        int rrc = vgicp_align_recut(h, d_src, n_src, stride_bytes / 4, pose, converged);
        if (rrc == 2) {
            memcpy(pose, pose_in, sizeof pose_in);
            if (vgicp_prepare_target(h, d_dst, n_dst, stride_bytes / 4, nullptr)) return 1;
            rrc = vgicp_align_recut(h, d_src, n_src, stride_bytes / 4, pose, converged);
        }
        if (rrc) return 1;
        if (h->profile >= 1) {
            H_TRY(hipEventRecord(h->ev_end, h->stream));
            H_TRY(hipEventSynchronize(h->ev_end));
            float ms = 0;
            H_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_end)); h->stats.total_ms = ms;
            H_TRY(hipEventElapsedTime(&ms, h->ev_start, h->ev_index)); h->stats.index_ms = ms;
            H_TRY(hipEventElapsedTime(&ms, h->ev_index, h->ev_end)); h->stats.solve_ms = ms;
            if (prof_end(h)) return 1;
        }
        return 0;
    }
    // the reference rebuilds its index on every call (LoamRegister.cpp:110); so do we
    h->clamp.use = 0;
    if (build_target(h, d_dst, n_dst, stride_bytes / 4)) {
        if (!sharded(h)) return 1;
        h->grid.valid = false;         // sharded: run_loam tells the other ranks and all of them fail together
        h->tgt_ptr = d_dst; h->tgt_n = n_dst; h->tgt_stride = stride_bytes / 4;
    }
    if (h->profile >= 1) H_TRY(hipEventRecord(h->ev_index, h->stream));
    h->clamp_allowed = true;           // this target exists for this scan only: a box that cannot be tabulated may be cut around it
    const int rc = run_loam(h, d_src, n_src, stride_bytes / 4, pose, converged, true);
    h->clamp_allowed = false;
    if (h->clamp.use) { h->clamp.use = 0; h->have_target = false; h->grid.valid = false; }      // not an index pcr_align may reuse
    return rc;
}

}  // namespace

extern "C" {

void pcr_default_params(pcr_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->struct_size = sizeof(pcr_params);
    p->device = -1;
    p->loam_iters = 8; p->loam_early_exit = 1;
    p->loam_knn_max_sq = (double)1.0f; p->loam_plane_thresh = (double)0.2f; p->loam_point_thresh = (double)0.1f;
    p->loam_pos_conv = (double)5e-3f; p->loam_rot_conv = (double)5e-3f;
    p->ndt_resolution = 1.0; p->ndt_step_size = 0.1; p->ndt_outlier_ratio = 0.55; p->ndt_trans_eps = 0.1;
    p->ndt_max_iters = 35; p->ndt_min_points = 6;
    p->vgicp_resolution = 1.0; p->vgicp_k_corr = 20; p->vgicp_max_iters = 64; p->vgicp_lm_inner = 10;
    p->vgicp_rot_eps = 2e-3; p->vgicp_trans_eps = 5e-4; p->vgicp_lm_init_scale = 1e-9;
    p->record_trace = 0;
}

pcr_handle* pcr_create(const char* method, const pcr_params* p) {
    g_create_error.clear();
    if (!method) { g_create_error = "method is NULL"; return nullptr; }
    Method m;
    if (!strcmp(method, "loam")) m = kLoam;
    else if (!strcmp(method, "ndt")) m = kNdt;
    else if (!strcmp(method, "vgicp")) m = kVgicp;
    else {
        // the reference factory throws on an unknown frontend.pcr (LidarOdometry.cpp:50-54)
        g_create_error = std::string("such pcr type(") + method + ") is not exist";
        return nullptr;
    }
    pcr_handle* h = new pcr_handle();
    h->method = m;
    pcr_default_params(&h->prm);
    if (p) {
        if (p->struct_size != sizeof(pcr_params)) { g_create_error = "pcr_params.struct_size mismatch (call pcr_default_params first)"; delete h; return nullptr; }
        h->prm = *p;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        delete h; return nullptr;
    }
    if (h->prm.device >= 0) h->device = h->prm.device;
    else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    if (h->device >= ndev) { g_create_error = "device ordinal out of range"; delete h; return nullptr; }
    if ((e = hipSetDevice(h->device)) != hipSuccess || (e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&h->ev_start)) != hipSuccess || (e = hipEventCreate(&h->ev_index)) != hipSuccess ||
        (e = hipEventCreate(&h->ev_end)) != hipSuccess) {
        g_create_error = std::string("HIP initialisation failed: ") + hipGetErrorString(e);
        delete h; return nullptr;
    }
    h->own_stream = true;
    memset(&h->stats, 0, sizeof(h->stats));
    return h;
}

void pcr_destroy(pcr_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); pin_forget_stream(h->stream); }
    if (dev_env("PCR_VF_DEBUG") && h->vf_builds) fprintf(stderr, "voxel filter: %llu index builds, %llu of them over a box or layout that did not hold the cloud\n", h->vf_builds, h->vf_stale);      // (development builds)
    if (h->comm && g_rccl.destroy) g_rccl.destroy(h->comm);
    h->grid.release(); h->tgt_stage.release(); h->src_stage.release();
    peer_close(h);
    if (h->peer_own) (void)hipFree(h->peer_own);
    if (h->peer_status_host) (void)hipHostFree(h->peer_status_host);
    if (h->vf_ret) (void)hipHostFree(h->vf_ret);
    h->prof_count.release();
    for (hipEvent_t e : h->ev_cov) if (e) (void)hipEventDestroy(e);
    h->vf_grid.release(); h->vf_in.release(); h->vf_out.release(); h->vf_head.release(); h->vf_sums.release(); h->vf_count.release();
    if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    h->src_grid.release(); h->cov_l1.release(); h->cov_l2.release(); h->src_l1.release(); h->src_l2.release(); h->tgt_cov6.release(); h->src_cov6.release(); h->vox.release(); h->src_scratch.release(); h->tgt_scratch.release();
    h->corr_slot.release(); h->corr_M.release(); h->corr_slot2.release(); h->corr_M2.release(); h->vg_partials.release(); h->vg_ctl.release(); h->vg_reduced.release(); h->fit_src.release();
    if (h->vg_out_host) (void)hipHostFree(h->vg_out_host);
    if (h->out32_host) (void)hipHostFree(h->out32_host);
    h->nd_slot.release(); h->nd_vox.release(); h->nd_count.release(); h->nd_list.release(); h->nd_partials.release();
    if (h->out48_host) (void)hipHostFree(h->out48_host);
    if (h->nd_out_host) (void)hipHostFree(h->nd_out_host);
    h->nd_ctl.release(); h->nd_sums.release();
    h->loam_state.release(); h->loam_partials.release(); h->loam_trace.release(); h->loam_reduced.release();
    h->dbg_status.release(); h->dbg_rows.release(); h->dbg_nn.release(); h->nn_cache.release(); h->timeline.release();
    if (h->result_host) (void)hipHostFree(h->result_host);
    if (h->red_host) (void)hipHostFree(h->red_host);
    h->ar_stage.release(); h->dummy_grid.release(); h->cov_viol.release();
    h->roi_mark[0].release(); h->roi_mark[1].release(); h->roi_tmp.release(); h->roi_mask.release(); h->roi_esc.release();
    for (hipEvent_t e : h->ev_kernel) (void)hipEventDestroy(e);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_index) (void)hipEventDestroy(h->ev_index);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->side_hdr) (void)hipHostFree(h->side_hdr);
    if (h->ev_side_in) (void)hipEventDestroy(h->ev_side_in);
    if (h->ev_hdr) (void)hipEventDestroy(h->ev_hdr);
    if (h->ev_aux_in) (void)hipEventDestroy(h->ev_aux_in);
    if (h->ev_aux_done) (void)hipEventDestroy(h->ev_aux_done);
    if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
    if (h->ev_side_done) (void)hipEventDestroy(h->ev_side_done);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* pcr_last_error(const pcr_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pcr_scan2map(pcr_handle* h, const void* src, size_t n_src, const void* dst, size_t n_dst, size_t stride_bytes,
                 double pose_inout[16], int* converged) {
    return do_scan2map(h, src, n_src, dst, n_dst, stride_bytes, pose_inout, converged, false);
}

int pcr_scan2map_device(pcr_handle* h, const void* d_src, size_t n_src, const void* d_dst, size_t n_dst,
                        size_t stride_bytes, double pose_inout[16], int* converged) {
    return do_scan2map(h, d_src, n_src, d_dst, n_dst, stride_bytes, pose_inout, converged, true);
}

// target structures of the handle's method from points in HBM that outlive them (the staging copy of pcr_set_target, or a
// pcr_map's sub-map for as long as its generation lasts)
static int prepare_target_from(pcr_handle* h, const float* d_dst, size_t n_dst, size_t stride_bytes) {
    h->map_id = 0; h->map_gen = 0;      // (pcr_scan2map_submap sets them after a successful preparation)
    if (h->method == kVgicp) return agree_prepared(h, vgicp_prepare_target(h, d_dst, n_dst, stride_bytes / 4));
    if (h->method == kNdt) return agree_prepared(h, ndt_prepare_target(h, d_dst, n_dst, stride_bytes / 4));
    h->clamp.use = 0;
    // settle the cell-table size now so that pcr_align never has to rebuild
    int rc = build_target(h, d_dst, n_dst, stride_bytes / 4);
    h->clamp_from_bulk = true;          // a box that cannot be tabulated is cut to the bulk of the cloud (set_clamp_from_target_sample)
    if (!rc) rc = settle_loam_index(h, nullptr, 0, 0, nullptr);
    h->clamp_from_bulk = false;
    if (rc) { h->have_target = false; h->grid.valid = false; }
    return agree_prepared(h, rc);
}

// The reference keeps its target after scan2Map (setInputTarget holds the cloud; test/align.cpp aligns and scores against it afterwards).  A
// pcr_scan2map of an NDT / VGICP handle prepares the target for that scan's region only (RoiView): a later pcr_align, pcr_vgicp_linearize or
// pcr_ndt_derivatives on the same target finds it prepared IN FULL here -- from this handle's own staging copy when the target came in as a host
// buffer (the plugin adapter's path).  A device buffer is the caller's and may be gone: that case asks for pcr_set_target / full_target.
static int ensure_full_target(pcr_handle* h) {
    if (!h->roi_on) return 0;
    if (h->tgt_ptr != h->tgt_stage.as<float>() || !h->tgt_n || sharded(h))
        return fail(h, "no target: pcr_scan2map prepared this device-resident target for that one scan only; call pcr_set_target (or set pcr_params.full_target) "
                       "for a target that is kept");
    const uint64_t id = h->map_id, gen = h->map_gen;
    if (prepare_target_from(h, h->tgt_stage.as<float>(), h->tgt_n, h->tgt_stride * sizeof(float))) return 1;
    h->map_id = id; h->map_gen = gen;
    return 0;
}

int pcr_set_target(pcr_handle* h, const void* dst, size_t n_dst, size_t stride_bytes, int on_device) {
    if (!h) return 1;
    h->err.clear();
    if (n_dst && !dst) return fail(h, "NULL cloud with nonzero size");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    // the library copies what it keeps: the index holds its own sorted copy, but a rebuild after a
    // cell-table overflow needs the raw points, so they are staged in HBM either way
    const size_t bytes = n_dst * stride_bytes;
    if (on_device) {
        H_TRY(h->tgt_stage.reserve(bytes ? bytes : 16));
        if (bytes) H_TRY(hipMemcpyAsync(h->tgt_stage.p, dst, bytes, hipMemcpyDeviceToDevice, h->stream));
    } else {
        const float* staged = nullptr;
        if (stage_host(h, &h->tgt_stage, dst, n_dst, stride_bytes, &staged)) return 1;
    }
    return prepare_target_from(h, h->tgt_stage.as<float>(), n_dst, stride_bytes);
}

int pcr_scan2map_submap(pcr_handle* h, const void* src, size_t n_src, int src_on_device, const pcr_map* m, double pose_inout[16],
                        int* converged) {
    if (!h) return 1;
    h->err.clear();
    if (!m) return fail(h, "map is NULL");
    uint64_t id = 0, gen = 0;
    size_t n_dst = 0, stride_bytes = 0;
    if (pcr_map_generation(m, &id, &gen)) return fail(h, "map is not usable");
    const void* d_dst = pcr_map_submap(m, &n_dst, &stride_bytes);
    if (!d_dst || n_dst == 0 || stride_bytes == 0) {
        // an empty sub-map (no key frame selected yet): the ordinary entry point defines what registration against nothing returns
        if (stride_bytes == 0) stride_bytes = 16;
        pcr_invalidate_target(h);
        return do_scan2map(h, src, n_src, nullptr, 0, stride_bytes, pose_inout, converged, src_on_device != 0);
    }
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    const bool current = h->map_id == id && h->map_gen == gen && h->have_target &&
                         (h->method == kLoam ? h->grid.valid && !h->clamp.use : (h->method == kNdt ? h->nd_target_ready : h->vg_target_ready));
    // Sharded: prepare_target_from() ends in a collective (agree_prepared), so either every rank rebuilds or none does -- a rank
    // whose tile index was cut, or whose map is at another generation, must not enter that exchange while its peers are already
    // summing normal equations.  One MAX over the ranks of "I have to rebuild" decides for all.
    double need = current ? 0.0 : 1.0;
    if (sharded(h) && ranks_allreduce(h, &need, 1, 1)) return 1;
    if (need != 0.0) {
        // VGICP, a scan that is in HBM already: its side goes onto the side stream FIRST and runs beside the new sub-map's preparation (clouds of a few ten
        // thousand points: the device is mostly idle behind either)
        if (h->method == kVgicp && !sharded(h) && src_on_device && n_src > 0 && src && vgicp_source_enqueue(h, static_cast<const float*>(src), n_src, stride_bytes / 4)) return 1;
        if (prepare_target_from(h, static_cast<const float*>(d_dst), n_dst, stride_bytes)) {
            if (h->side_pending) { (void)hipEventSynchronize(h->ev_side_done); h->side_pending = false; }      // (nothing of this call stays in flight)
            return 1;
        }
        h->map_id = id; h->map_gen = gen;
        h->target_builds += 1;
    }
    return pcr_align(h, src, n_src, stride_bytes, src_on_device, pose_inout, converged);
}

int pcr_align(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device, double pose_inout[16],
              int* converged) {
    if (!h) return 1;
    h->err.clear();
    if (!pose_inout) return fail(h, "pose_inout is NULL");
    if (n_src && !src) return fail(h, "NULL cloud with nonzero size");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    if (!h->have_target || !h->grid.valid) return fail(h, "no target: call pcr_set_target first");
    if (ensure_full_target(h)) return 1;
    drop_fitness_state(h);
    const float* d_src = (const float*)src;
    if (!on_device && stage_host(h, &h->src_stage, src, n_src, stride_bytes, &d_src)) return 1;
    if (h->method == kNdt) return run_ndt(h, d_src, n_src, stride_bytes / 4, pose_inout, converged);
    if (h->method == kVgicp) {
        // the scan's own side (two index levels, the 20-neighbour search, the covariances) queued on the side stream with its headers mirrored to the host, as
        // pcr_scan2map queues it -- unless pcr_scan2map_submap has queued it already, ahead of the target's preparation.  Settled the checked way (a header
        // read back after every level: two host round trips with an idle device in between, ~55 us of a 0.3 ms call against a kept sub-map) only if that fails.
        if (!sharded(h) && n_src > 0 && !(h->side_pending && h->side_src == d_src && h->side_n == n_src && h->side_stride == stride_bytes / 4) &&
            vgicp_source_enqueue(h, d_src, n_src, stride_bytes / 4)) return 1;
        const int rc = vgicp_align_recut(h, d_src, n_src, stride_bytes / 4, pose_inout, converged) ? 1 : 0;
        if (h->side_pending) { (void)hipEventSynchronize(h->ev_side_done); h->side_pending = false; }      // (an alignment that failed before it collected the scan's side: nothing stays in flight behind the caller's back)
        return rc;
    }
    return run_loam(h, d_src, n_src, stride_bytes / 4, pose_inout, converged, false);
}

int pcr_invalidate_target(pcr_handle* h) {
    if (!h) return 1;
    h->have_target = false; h->grid.valid = false; h->vg_target_ready = false; h->nd_target_ready = false;
    h->map_id = 0; h->map_gen = 0; h->roi_on = false;
    return 0;
}

}  // extern "C"
namespace {
// The voxel filter in two halves: everything queued (vf_enqueue), then the one synchronisation and what its result asks for (vf_settle).
// pcr_voxel_filter is the two back to back; the sub-map assembly (submap.hip) queues an assembly with the first and collects it with the second when the
// sub-map is next asked for -- the kernels of an assembly then run beside the next scan's own filter (pcr_map_update_begin).
int vf_enqueue(pcr_handle* h) {
    const pcr_handle::VfJob& j = h->vf_job;
    if (h->vf_grid.build(j.d_pts, j.n, j.sf, j.leaf, h->stream, &h->err, 0.0, 1, nullptr, true) != hipSuccess) return 1;
    H_TRY(voxel_filter_launch(h->vf_grid, j.d_pts, j.sf, j.n, h->vf_head.as<uint32_t>(), h->vf_sums.as<uint32_t>(), h->vf_count.p, j.d_out, j.cap,
                              h->vf_ret, h->stream));
    return 0;
}
int vf_begin(pcr_handle* h, const float* d_pts, size_t n, size_t sf, double leaf, float* d_out, size_t cap) {
    // ONE round trip: the two launches of the filter are queued right behind the index build -- they read the header themselves and do nothing
    // when it says overflow, stale or empty -- and their last block writes the voxel count and the header's verdict into page-locked memory.
    // (Round 4 read the header first, then the count: two more synchronisations and an idle device in between, ~35 us of the 0.16 ms a
    //  65 536-point scan took: round 5, scripts/seq_breakdown.py.)
    // The index reuses the previous call's box and tile layout when the cloud still fits (GridIndex::hint_ok): the order of the voxels -- idx sorts by
    // (z, y, x) -- does not depend on where the box starts, so the output is the same either way; a cloud that does not fit comes back `stale` and is
    // built afresh, from then on with room around the box (a sub-map's box moves with the vehicle) and half as much again per bin.
    const size_t nr = std::max(n, h->vf_grid.reserve_points);
    H_TRY(h->vf_head.reserve((nr + 4096) * sizeof(uint32_t)));
    H_TRY(h->vf_sums.reserve((nr / 2048 + 2) * sizeof(uint32_t)));
    H_TRY(h->vf_count.reserve(voxel_filter_wave_bytes(nr)));
    if (!h->vf_ret) H_TRY(hipHostMalloc((void**)&h->vf_ret, sizeof(VfResult) + 64, hipHostMallocDefault));
    h->vf_grid.no_hints = h->prm.index_no_hints != 0;
    h->vf_grid.cut_sparse = true; h->vf_grid.coherent_input = true;
    // (room around the box and in the bins from the first build on: a handle's second cloud never fits the first one's tight box, and that build was made twice)
    if (h->vf_grid.hint_margin == 0) { h->vf_grid.hint_margin = 16; h->vf_grid.hint_margin_z_pcl = 4; h->vf_grid.lay_room_shift = 1; h->vf_grid.lay_room_add = 256; }
    h->vf_job = pcr_handle::VfJob{d_pts, n, sf, leaf, d_out, cap};
    return vf_enqueue(h);
}
int vf_settle(pcr_handle* h, uint32_t* count, int* too_fine) {
    volatile VfResult& ret = *reinterpret_cast<VfResult*>(h->vf_ret);
    const size_t n = h->vf_job.n;
    for (int attempt = 0; attempt < 6; ++attempt) {
        if (attempt && vf_enqueue(h)) return 1;
        H_TRY(hipStreamSynchronize(h->stream));
        ++h->vf_builds;
        if (dev_env("PCR_VF_DEBUG") && n > 100000) {      // (development builds: the layout this build planned -- bins, tiles, the heaviest bin)
            std::vector<uint32_t> lay(3 * kMaxBins + 32);
            (void)hipMemcpy(lay.data(), h->vf_grid.layout[h->vf_grid.lay_idx].p, lay.size() * 4, hipMemcpyDeviceToHost);
            const uint32_t nb = lay[3 * kMaxBins + 24], nt = lay[3 * kMaxBins + 25];
            uint32_t room_max = 0, kmax = 0;
            for (uint32_t b = 0; b < nb && b < (uint32_t)kMaxBins; ++b) { room_max = std::max(room_max, lay[b + 1] - lay[b]); kmax = std::max(kmax, lay[2 * kMaxBins + 16 + b] >> 26); }
            fprintf(stderr, "voxel filter: n %zu used_hint %d used_layout %d tshift %d stale %d overflow %d | next layout: %u bins over %u tiles, widest room %u, deepest cut %u\n",
                    n, (int)h->vf_grid.used_hint, (int)h->vf_grid.used_layout, h->vf_grid.tiled_shift, (int)ret.stale, (int)ret.overflow, nb, nt, room_max, kmax);
        }
        if (ret.stale) {      // the box (or a bin's room) taken over from the previous call does not hold this cloud
            ++h->vf_stale;
            if (dev_env("PCR_VF_DEBUG")) fprintf(stderr, "voxel filter: n %zu leaf %g stale %d (1 box, 2 room, 3 layout)\n", n, h->vf_job.leaf, (int)ret.stale);
            // (the cell count the tile size goes by is kept once the box has its margin: a build without it takes the dense path -- 23 us instead of 11 for a scan)
            if (h->vf_grid.hint_margin == 0) { h->vf_grid.hint_margin = 16; h->vf_grid.hint_margin_z_pcl = 4; h->vf_grid.cells_hint = 0; }
            h->vf_grid.lay_room_shift = 1; h->vf_grid.lay_room_add = 256;
            continue;
        }
        if (!ret.overflow) {
            *count = ret.count; *too_fine = ret.too_fine;
            h->vf_grid.note_cells(ret.n_cells);
            if (!ret.empty && !ret.too_fine) h->vf_grid.confirm();
            return 0;
        }
        if (h->vf_grid.grow_cells(ret.n_cells, &h->err) != hipSuccess) return 1;
    }
    return fail(h, "voxel table could not be sized");
}
}  // namespace
// (library-internal, for submap.hip: a filter of device memory into device memory, queued / collected; out_capacity >= n)
int pcr_internal_vf_begin(pcr_handle* h, const void* d_pts, size_t n, size_t stride_bytes, double leaf, void* d_out, size_t out_capacity) {
    if (!h) return 1;
    h->err.clear();
    if (!d_pts || !d_out || n == 0 || n > 0xfffffff0ull || out_capacity < n) return fail(h, "voxel filter: bad arguments");
    if (!(leaf > 0)) return fail(h, "leaf size must be positive");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    return vf_begin(h, static_cast<const float*>(d_pts), n, stride_bytes / 4, leaf, static_cast<float*>(d_out), out_capacity);
}
void pcr_internal_vf_reserve(pcr_handle* h, size_t points) { if (h && points <= 0xfffffff0ull) h->vf_grid.reserve_points = std::max(h->vf_grid.reserve_points, points); }
int pcr_internal_vf_end(pcr_handle* h, size_t* n_out) {
    if (!h || !n_out) return 1;
    *n_out = 0;
    if (set_device(h)) return 1;
    uint32_t count = 0;
    int too_fine = 0;
    if (vf_settle(h, &count, &too_fine)) return 1;
    if (too_fine) {      // pcl::VoxelGrid: output = input (see pcr_voxel_filter)
        const pcr_handle::VfJob& j = h->vf_job;
        H_TRY(hipMemcpyAsync(j.d_out, j.d_pts, j.n * j.sf * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        *n_out = j.n;
        return 0;
    }
    *n_out = count;
    return 0;
}
extern "C" {

int pcr_voxel_filter(pcr_handle* h, const void* pts, size_t n, size_t stride_bytes, int on_device, double leaf, void* out,
                     size_t out_capacity, int out_on_device, size_t* n_out) {
    if (!h) return 1;
    h->err.clear();
    if (!n_out) return fail(h, "n_out is NULL");
    *n_out = 0;
    if (n && !pts) return fail(h, "NULL cloud with nonzero size");
    if (out_capacity && !out) return fail(h, "NULL output with nonzero capacity");
    if (!(leaf > 0)) return fail(h, "leaf size must be positive");
    if (n > 0xfffffff0ull) return fail(h, "cloud too large");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    if (h->vf_inflight) return fail(h, "a voxel filter is queued on this handle (pcr_voxel_filter_begin): collect it with pcr_voxel_filter_end first");
    if (n == 0) return 0;
    const size_t sf = stride_bytes / 4;
    const float* d_pts = static_cast<const float*>(pts);
    if (!on_device && stage_host(h, &h->vf_in, pts, n, stride_bytes, &d_pts, true)) return 1;
    float* d_out = static_cast<float*>(out);
    size_t cap = out_capacity;
    if (!out_on_device) {
        cap = std::min(out_capacity, n);
        H_TRY(h->vf_out.reserve((cap ? cap : 1) * stride_bytes));
        d_out = h->vf_out.as<float>();
    }
    uint32_t count = 0;
    int too_fine = 0;
    if (vf_begin(h, d_pts, n, sf, leaf, d_out, cap) || vf_settle(h, &count, &too_fine)) return 1;
    if (too_fine) {
        // pcl::VoxelGrid: "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = input
        *n_out = n;
        if (out_capacity < n) return fail(h, "output capacity too small (leaf too small for the data: the input is returned unfiltered)");
        H_TRY(hipMemcpyAsync(out, d_pts, n * stride_bytes, out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        h->err = "leaf size too small for the input: integer voxel indices would overflow; input returned unfiltered";
        return 0;
    }
    *n_out = count;
    if (count > out_capacity) return fail(h, "output capacity too small: " + std::to_string(count) + " voxels are occupied");
    if (!out_on_device && count) H_TRY(hipMemcpy(out, d_out, (size_t)count * stride_bytes, hipMemcpyDeviceToHost));
    return 0;
}

int pcr_voxel_filter_begin(pcr_handle* h, const void* d_pts, size_t n, size_t stride_bytes, double leaf, void* d_out, size_t out_capacity) {
    if (!h) return 1;
    h->err.clear();
    if (h->vf_inflight) return fail(h, "a voxel filter is already queued on this handle: collect it with pcr_voxel_filter_end first");
    if (n && (!d_pts || !d_out)) return fail(h, "NULL cloud or output with nonzero size");
    if (out_capacity < n) return fail(h, "pcr_voxel_filter_begin needs room for n points (a leaf too small for the data returns the input unfiltered)");
    if (!(leaf > 0)) return fail(h, "leaf size must be positive");
    if (n > 0xfffffff0ull) return fail(h, "cloud too large");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    h->vf_inflight_n = n;
    if (n && pcr_internal_vf_begin(h, d_pts, n, stride_bytes, leaf, d_out, out_capacity)) return 1;
    h->vf_inflight = true;
    return 0;
}

int pcr_voxel_filter_end(pcr_handle* h, size_t* n_out) {
    if (!h) return 1;
    if (!n_out) return fail(h, "n_out is NULL");
    *n_out = 0;
    if (!h->vf_inflight) return fail(h, "no voxel filter is queued on this handle");
    h->vf_inflight = false;
    h->err.clear();
    if (h->vf_inflight_n == 0) return 0;
    return pcr_internal_vf_end(h, n_out);
}

}  // extern "C"
hipStream_t pcr_internal_stream(const pcr_handle* h) { return h ? h->stream : nullptr; }
extern "C" {

double pcr_fitness(pcr_handle* h) {
    if (!h) return -1.0;
    // PointCloudRegister::getFitnessScore() returns 0 unless overridden (PointCloudRegister.hpp:34);
    // only VgicpRegister overrides it (VgicpRegister.cpp:42-45)
    if (h->method != kVgicp) return 0.0;
    if (h->fit_pending) {
        // mean squared distance of the aligned scan to its nearest target points, against the target the handle holds NOW (PCL does
        // the same: input_ transformed by final_transformation_, searched in the current target tree)
        h->fit_pending = false;
        h->err.clear();
        h->fitness = 1.7976931348623157e308;
        if (h->vg_target_ready && h->fit_n > 0) {
            if (set_device(h) || ensure_out32(h)) return -1.0;
            h->seq += 1.0;
            // (a lattice that holds the scan's region only cannot answer a nearest-neighbour question; the grid its covariances were searched on holds every point)
            const GridIndex& fit_grid = (h->grid.filtered && h->cov_l1.valid && !h->cov_l1.filtered) ? h->cov_l1 : h->grid;
            if (fitness_launch(fit_grid, h->fit_src.as<float>(), h->fit_n, h->fit_stride, h->fit_pose, 1.7976931348623157e308, h->vg_partials.as<double>(),
                               h->out32_dev, h->stream, h->seq, nullptr) != hipSuccess) { h->err = "fitness_launch failed"; return -1.0; }
            if (wait_result(h, &h->out32_host[31], h->seq)) return -1.0;
            h->fitness = h->out32_host[1] > 0 ? h->out32_host[0] / h->out32_host[1] : 1.7976931348623157e308;
        }
    }
    return h->fitness;
}

int pcr_loam_linearize(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device,
                       const double pose[16], double JtJ[36], double JtE[6], int64_t* n_accepted, int8_t* status,
                       double* rows, int32_t* nn) {
    if (!h) return 1;
    h->err.clear();
    if (h->method != kLoam) return fail(h, "pcr_loam_linearize needs a loam handle");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    if (!h->have_target || !h->grid.valid) return fail(h, "no target: call pcr_set_target first");
    if (ensure_loam_buffers(h)) return 1;
    const float* d_src = (const float*)src;
    if (!on_device && stage_host(h, &h->src_stage, src, n_src, stride_bytes, &d_src)) return 1;
    LoamArgs a;
    fill_loam_args(h, &a, d_src, n_src, stride_bytes / 4, pose);
    a.trace = nullptr;
    if (status) { H_TRY(h->dbg_status.reserve(n_src + 16)); a.dbg_status = h->dbg_status.as<int8_t>(); }
    if (rows) { H_TRY(h->dbg_rows.reserve(n_src * 7 * sizeof(double) + 16)); a.dbg_rows = h->dbg_rows.as<double>(); }
    if (nn) { H_TRY(h->dbg_nn.reserve(n_src * 5 * sizeof(int32_t) + 16)); a.dbg_nn = h->dbg_nn.as<int32_t>(); }
    H_TRY(loam_launch_iteration(a, 0, h->stream));
    H_TRY(loam_launch_reduce(a, 0, h->loam_reduced.as<double>(), h->stream));
    double sums[kAccum];
    H_TRY(hipMemcpyAsync(sums, h->loam_reduced.p, sizeof(sums), hipMemcpyDeviceToHost, h->stream));
    if (status) H_TRY(hipMemcpyAsync(status, h->dbg_status.p, n_src, hipMemcpyDeviceToHost, h->stream));
    if (rows) H_TRY(hipMemcpyAsync(rows, h->dbg_rows.p, n_src * 7 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (nn) H_TRY(hipMemcpyAsync(nn, h->dbg_nn.p, n_src * 5 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));
    int q = 0;
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { JtJ[r * 6 + c] = JtJ[c * 6 + r] = sums[q++]; }
    for (int r = 0; r < 6; ++r) JtE[r] = sums[21 + r];
    if (n_accepted) *n_accepted = (int64_t)sums[27];
    return 0;
}

int pcr_vgicp_covariances(pcr_handle* h, const void* pts, size_t n, size_t stride_bytes, int on_device, double* cov_out) {
    if (!h) return 1;
    h->err.clear();
    if (h->method != kVgicp) return fail(h, "pcr_vgicp_covariances needs a vgicp handle");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    const float* d_pts = (const float*)pts;
    if (!on_device && stage_host(h, &h->src_stage, pts, n, stride_bytes, &d_pts)) return 1;
    if (h->side_pending) { H_TRY(hipEventSynchronize(h->ev_side_done)); h->side_pending = false; }
    if (settle_cov_levels(h, h->src_grid, h->src_l1, h->src_l2, d_pts, n, stride_bytes / 4, h->prm.vgicp_resolution, 0.0, nullptr, false, 0.0, nullptr, nullptr, nullptr, nullptr, true)) return 1;
    H_TRY(h->src_cov6.reserve((n + 1) * 6 * sizeof(double)));
    H_TRY(hipMemsetAsync(h->src_cov6.p, 0, (n + 1) * 6 * sizeof(double), h->stream));
    H_TRY(vgicp_launch_cov(h->src_grid, cov_levels(n) > 1 ? &h->src_l1 : nullptr, cov_levels(n) > 2 ? &h->src_l2 : nullptr, d_pts, stride_bytes / 4, n,
                           h->src_cov6.as<double>(), h->stream, nullptr, nullptr, &h->src_scratch));
    H_TRY(hipMemcpyAsync(cov_out, h->src_cov6.p, n * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

#ifdef PCR_DEV_SWITCHES
int pcr_dev_read_stamps(unsigned long long* out, size_t count) { return pcr::dev_stamps(out, count); }
#endif

int pcr_vgicp_neighbours(pcr_handle* h, size_t n, uint32_t* nbr_out, uint32_t* queued_out) {
    if (!h) return 1;
    h->err.clear();
    if (h->method != kVgicp) return fail(h, "pcr_vgicp_neighbours needs a vgicp handle");
    if (set_device(h)) return 1;
    const CovScratch& sc = h->src_scratch;
    if (!sc.nbr.p || !h->src_grid.valid || h->src_grid.n_points > n || n > 300000) return fail(h, "no neighbour lists of a cloud of that size: call pcr_vgicp_covariances on a scan-sized cloud first");
    const size_t n_cap = std::min(sc.queue.cap / sizeof(uint32_t), sc.nbr.cap / (20 * sizeof(uint32_t))), ns = h->src_grid.n_points;
    std::vector<uint32_t> lists(n_cap * 20);
    std::vector<float> sorted(ns * 4);
    uint32_t queued = 0;
    H_TRY(hipStreamSynchronize(h->stream));
    H_TRY(hipMemcpy(lists.data(), sc.nbr.p, lists.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    H_TRY(hipMemcpy(sorted.data(), h->src_grid.sorted.p, sorted.size() * sizeof(float), hipMemcpyDeviceToHost));
    H_TRY(hipMemcpy(&queued, sc.count.p, sizeof queued, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n * 20; ++i) nbr_out[i] = 0xffffffffu;
    for (size_t j = 0; j < ns; ++j) {
        uint32_t orig;
        memcpy(&orig, &sorted[j * 4 + 3], 4);
        if (orig >= n) return fail(h, "the scan index does not belong to a cloud of that size");
        for (int k = 0; k < 20; ++k) nbr_out[(size_t)orig * 20 + k] = lists[(size_t)k * n_cap + j];
    }
    if (queued_out) *queued_out = queued;
    return 0;
}

int pcr_vgicp_linearize(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device, const double pose[16],
                        double H[36], double b[6], double* error, int64_t* n_corr) {
    if (!h) return 1;
    h->err.clear();
    if (h->method != kVgicp) return fail(h, "pcr_vgicp_linearize needs a vgicp handle");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    if (!h->vg_target_ready) return fail(h, "no target: call pcr_set_target first");
    if (ensure_full_target(h)) return 1;
    const float* d_src = (const float*)src;
    if (!on_device && stage_host(h, &h->src_stage, src, n_src, stride_bytes, &d_src)) return 1;
    // run the driver's set-up with zero iterations, then one linearisation at the given pose
    const int saved = h->prm.vgicp_max_iters;
    h->prm.vgicp_max_iters = 0;
    double tmp[16];
    memcpy(tmp, pose, sizeof tmp);
    int conv = 0;
    const int rc = run_vgicp(h, d_src, n_src, stride_bytes / 4, tmp, &conv);
    h->prm.vgicp_max_iters = saved;
    if (rc) return 1;
    VgicpArgs a;
    memset(&a, 0, sizeof a);      // (roi.mask = nullptr: the whole target is prepared, unless set below)
    a.src = d_src; a.n_src = (uint32_t)n_src; a.src_stride = (uint32_t)(stride_bytes / 4);
    a.src_cov6 = h->src_cov6.as<double>();
    a.hdr = h->grid.header.as<GridHeader>();
    a.cell_start = h->grid.cell_start.as<uint32_t>();
    a.vox = h->vox.as<VgicpVoxel>();
    a.corr_slot = h->corr_slot.as<uint32_t>(); a.corr_M = h->corr_M.as<double>();
    a.corr_slot_next = h->corr_slot2.as<uint32_t>(); a.corr_M_next = h->corr_M2.as<double>();
    a.partials = h->vg_partials.as<double>();
    a.use_tile = h->use_tile; a.pad_ = 0;
    a.escapes = nullptr; a.guard_cells = 0; a.pad2_ = 0;
    for (int d = 0; d < 3; ++d) { a.tile_lo[d] = h->tile_lo[d]; a.tile_hi[d] = h->tile_hi[d]; }
    Pose16 T;
    memcpy(T.m, pose, sizeof T.m);
    H_TRY(vgicp_launch_linearize(a, T, h->out32_dev, h->stream));
    std::vector<uint32_t> slots(n_src);
    if (n_src) H_TRY(hipMemcpyAsync(slots.data(), h->corr_slot.p, n_src * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));
    if (sharded(h) && ranks_allreduce(h, h->out32_host, 29)) return 1;
    int q = 0;
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { H[r * 6 + c] = H[c * 6 + r] = h->out32_host[q++]; }
    for (int r = 0; r < 6; ++r) b[r] = h->out32_host[21 + r];
    if (error) *error = h->out32_host[27];
    if (n_corr) { int64_t c = 0; for (uint32_t v : slots) c += v != 0; *n_corr = c; }
    return 0;
}

int pcr_ndt_derivatives(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device, const double p[6],
                        double* score, double grad[6], double hess[36], double* hess_d) {
    if (!h) return 1;
    h->err.clear();
    if (h->method != kNdt) return fail(h, "pcr_ndt_derivatives needs an ndt handle");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    if (!h->nd_target_ready) return fail(h, "no target: call pcr_set_target first");
    if (ensure_full_target(h)) return 1;
    const float* d_src = (const float*)src;
    if (!on_device && stage_host(h, &h->src_stage, src, n_src, stride_bytes, &d_src)) return 1;
    if (!h->out48_host) {
        H_TRY(hipHostMalloc((void**)&h->out48_host, 48 * sizeof(double), hipHostMallocMapped));
        memset(h->out48_host, 0, 48 * sizeof(double));
        H_TRY(hipHostGetDevicePointer((void**)&h->out48_dev, h->out48_host, 0));
    }
    H_TRY(h->nd_partials.reserve((size_t)1024 * 48 * sizeof(double)));
    NdtRun r;
    memset(&r.a, 0, sizeof r.a);
    r.h = h;
    r.a.src = d_src; r.a.n_src = (uint32_t)n_src; r.a.src_stride = (uint32_t)(stride_bytes / 4);
    r.a.hdr = h->grid.header.as<GridHeader>(); r.a.vox_slot = h->nd_slot.as<uint32_t>(); r.a.vox = h->nd_vox.as<NdtVoxel>();
    r.a.partials = h->nd_partials.as<double>();
    r.a.use_tile = h->use_tile; r.a.pad_ = 0;
    for (int d = 0; d < 3; ++d) { r.a.tile_lo[d] = h->tile_lo[d]; r.a.tile_hi[d] = h->tile_hi[d]; }
    const double res = (double)(float)h->prm.ndt_resolution;
    const double c1 = 10 * (1 - h->prm.ndt_outlier_ratio), c2 = h->prm.ndt_outlier_ratio / pow(res, 3), d3 = -log(c2);
    r.a.d1 = -log(c1 + c2) - d3;
    r.a.d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / r.a.d1);
    ndt_host::pose_from_p(p, &r.T);
    double sc = 0;
    if (ndt_derivatives(&r, p, true, &sc, grad, hess)) return 1;
    if (score) *score = sc;
    if (hess_d && n_src) {
        H_TRY(ndt_launch_hessian(r.a, r.T, r.ang, h->out48_dev, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        if (sharded(h) && ranks_allreduce(h, h->out48_host, 43)) return 1;
        for (int i = 0; i < 36; ++i) hess_d[i] = h->out48_host[7 + i];
    }
    return 0;
}

struct pcr_ndt_opt { NdtCtl c; };

pcr_ndt_opt* pcr_ndt_opt_create(const double pose_guess[16], double step_size, double trans_eps, int max_iters) {
    if (!pose_guess) return nullptr;
    pcr_ndt_opt* o = new pcr_ndt_opt;
    memset(&o->c, 0, sizeof o->c);
    // guess handed over as Matrix4f, Euler angles of its linear part (NdtRegister.cpp:27, ndt_omp_impl.hpp:103-111): as run_ndt does
    float G[16];
    for (int i = 0; i < 16; ++i) G[i] = (float)pose_guess[i];
    NdtPose T0;
    for (int rr = 0; rr < 3; ++rr) { for (int c = 0; c < 3; ++c) T0.R[rr * 3 + c] = G[c * 4 + rr]; T0.t[rr] = G[12 + rr]; }
    float eul[3];
    ndt_host::euler_xyz(T0.R, eul);
    const double p0[6] = {T0.t[0], T0.t[1], T0.t[2], eul[0], eul[1], eul[2]};
    ndt_opt::ctl_init(&o->c, T0, p0, step_size, trans_eps, max_iters);
    return o;
}
void pcr_ndt_opt_destroy(pcr_ndt_opt* o) { delete o; }
static void ndt_pose_out(const NdtPose& T, double pose16[16]) {
    for (int i = 0; i < 16; ++i) pose16[i] = 0;
    for (int rr = 0; rr < 3; ++rr) { for (int c = 0; c < 3; ++c) pose16[c * 4 + rr] = (double)T.R[rr * 3 + c]; pose16[12 + rr] = (double)T.t[rr]; }
    pose16[15] = 1.0;
}
int pcr_ndt_opt_request(const pcr_ndt_opt* o, int* kind, double p6[6], double pose16[16]) {
    if (!o || !kind) return 1;
    *kind = o->c.done ? kNdtPassNone : o->c.kind;
    if (p6) for (int i = 0; i < 6; ++i) p6[i] = o->c.x_t[i];
    if (pose16) ndt_pose_out(o->c.T, pose16);
    return 0;
}
int pcr_ndt_opt_feed(pcr_ndt_opt* o, const double sums[43]) {
    if (!o || !sums || o->c.done) return 1;
    ndt_opt::ctl_step(&o->c, sums);
    return 0;
}
int pcr_ndt_opt_result(const pcr_ndt_opt* o, double pose16[16], int* converged, int* iterations, int* done) {
    if (!o) return 1;
    if (pose16) ndt_pose_out(o->c.final_T, pose16);
    if (converged) *converged = o->c.conv;
    if (iterations) *iterations = o->c.nr_it;
    if (done) *done = o->c.done;
    return 0;
}

int pcr_ndt_opt_counts(const pcr_ndt_opt* o, int* evaluations, int* hessians, int* replayed) {
    if (!o) return 1;
    if (evaluations) *evaluations = o->c.n_deriv;
    if (hessians) *hessians = o->c.n_hess;
    if (replayed) *replayed = o->c.replayed;
    return 0;
}

struct pcr_vgicp_opt { VgCtl c; };

pcr_vgicp_opt* pcr_vgicp_opt_create(const double pose_guess[16], int max_iters, int lm_inner, double lm_init_scale, double rot_eps, double trans_eps) {
    if (!pose_guess) return nullptr;
    pcr_vgicp_opt* o = new pcr_vgicp_opt;
    memset(&o->c, 0, sizeof o->c);
    Pose16 g;
    for (int i = 0; i < 16; ++i) g.m[i] = (double)(float)pose_guess[i];      // guess handed over as Matrix4f (VgicpRegister.cpp:36), as run_vgicp does
    vg_opt::ctl_init(&o->c, g, max_iters, lm_inner, lm_init_scale, rot_eps, trans_eps);
    return o;
}
void pcr_vgicp_opt_destroy(pcr_vgicp_opt* o) { delete o; }
int pcr_vgicp_opt_request(const pcr_vgicp_opt* o, int* kind, double pose_eval[16], double pose_lin[16]) {
    if (!o || !kind) return 1;
    *kind = o->c.done ? 2 : o->c.kind;
    if (pose_eval) for (int i = 0; i < 16; ++i) pose_eval[i] = o->c.xi.m[i];
    if (pose_lin) for (int i = 0; i < 16; ++i) pose_lin[i] = o->c.x0.m[i];
    return 0;
}
int pcr_vgicp_opt_feed(pcr_vgicp_opt* o, const double sums[29]) {
    if (!o || !sums || o->c.done) return 1;
    vg_opt::ctl_step(&o->c, sums);
    return 0;
}
int pcr_vgicp_opt_result(const pcr_vgicp_opt* o, double pose16[16], int* converged, int* outer_iterations, int* done) {
    if (!o) return 1;
    if (pose16) for (int i = 0; i < 16; ++i) pose16[i] = o->c.x0.m[i];
    if (converged) *converged = o->c.conv;
    if (outer_iterations) *outer_iterations = o->c.outer;
    if (done) *done = o->c.done;
    return 0;
}

int pcr_get_trace(pcr_handle* h, int32_t* n_iters, double* JtJ, double* JtE, int64_t* n, double* x) {
    if (!h) return 1;
    if (!h->prm.record_trace) return fail(h, "trace not recorded: set pcr_params.record_trace");
    if (n_iters) *n_iters = h->trace_iters;
    for (int i = 0; i < h->trace_iters && i < (int)h->trace_host.size(); ++i) {
        const LoamTrace& t = h->trace_host[i];
        if (JtJ) memcpy(JtJ + i * 36, t.JtJ, sizeof(t.JtJ));
        if (JtE) memcpy(JtE + i * 6, t.JtE, sizeof(t.JtE));
        if (x) memcpy(x + i * 6, t.x, sizeof(t.x));
        if (n) n[i] = t.n;
    }
    return 0;
}

int pcr_get_trace_counts(pcr_handle* h, int64_t* cache_hits, int64_t* searches) {
    if (!h) return 1;
    if (!h->prm.record_trace) return fail(h, "trace not recorded: set pcr_params.record_trace");
    for (int i = 0; i < h->trace_iters && i < (int)h->trace_host.size(); ++i) {
        if (cache_hits) cache_hits[i] = h->trace_host[i].cache_hits;
        if (searches) searches[i] = h->trace_host[i].searches;
    }
    return 0;
}

int pcr_get_timeline(pcr_handle* h, uint64_t* out, size_t capacity, int* launches, int* blocks) {
    if (!h) return 1;
    if (h->prm.record_timeline != 1 || !h->timeline.p) return fail(h, "timeline not recorded: set pcr_params.record_timeline = 1");
    const int nl = std::max(1, h->prm.loam_iters), nb = (int)h->last_blocks;
    if (launches) *launches = nl;
    if (blocks) *blocks = nb;
    if (!out) return 0;
    if (capacity < (size_t)nl * nb * kTimelineSlots) return fail(h, "timeline buffer too small");
    H_TRY(hipStreamSynchronize(h->stream));
    for (int l = 0; l < nl; ++l)
        H_TRY(hipMemcpy(out + (size_t)l * nb * kTimelineSlots, h->timeline.as<unsigned long long>() + (size_t)l * kMaxPartials * kTimelineSlots,
                        (size_t)nb * kTimelineSlots * sizeof(uint64_t),
                        hipMemcpyDeviceToHost));
    return 0;
}

int pcr_get_stats(pcr_handle* h, pcr_stats* out) {
    if (!h || !out) return 1;
    *out = h->stats;
    out->target_builds = (int32_t)h->target_builds;
    out->region_repeats = (int32_t)h->roi_repeats; out->region_index = (h->roi_on && h->grid.filtered) ? 1 : 0;
    out->index_box_hint = h->grid.used_hint ? 1 : 0; out->index_layout_hint = h->grid.used_layout ? 1 : 0;
    return 0;
}

int pcr_set_profile(pcr_handle* h, int level) {
    if (!h) return 1;
    h->profile = level < 0 ? 0 : (level > 2 ? 2 : level);
    return 0;
}

int pcr_set_stream(pcr_handle* h, void* hip_stream) {
    if (!h) return 1;
    if (h->stream) { (void)hipStreamSynchronize(h->stream); pin_forget_stream(h->stream); }      // (copies out of pinned ranges are noted per stream)
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return 0;
}

int pcr_set_query_tile(pcr_handle* h, const double lo[3], const double hi[3]) {
    if (!h) return 1;
    h->err.clear();
    if (!lo || !hi || lo[0] > hi[0]) { h->use_tile = 0; h->have_halo = false; return 0; }
    // NDT and VGICP tiles must sit on the voxel lattice and come with a halo that is checked: pcr_set_shard
    if (h->method != kLoam) return fail(h, "pcr_set_query_tile serves loam handles; ndt and vgicp tiles are set with pcr_set_shard (voxel-aligned bounds + halo)");
    h->use_tile = 1; h->have_halo = false;
    for (int d = 0; d < 3; ++d) { h->tile_lo[d] = lo[d]; h->tile_hi[d] = hi[d]; }
    return 0;
}

int pcr_set_shard(pcr_handle* h, const double lo[3], const double hi[3], double halo) {
    if (!h) return 1;
    h->err.clear();
    if (!lo || !hi || lo[0] > hi[0]) { h->use_tile = 0; h->have_halo = false; return 0; }
    if (!(halo >= 0.0)) return fail(h, "halo must be >= 0");
    for (int d = 0; d < 3; ++d) if (!(lo[d] < hi[d])) return fail(h, "tile bounds must satisfy lo < hi on every axis");
    auto on_lattice = [](double v, double res, double shift) {      // v = (k + shift) * res for an integer k, or an open face
        if (open_face(v)) return true;
        const double k = v / res - shift;
        return fabs(k - nearbyint(k)) <= 1e-9 * std::max(1.0, fabs(k));
    };
    auto pow2 = [](double r) { int e; return frexp(r, &e) == 0.5; };
    if (h->method == kLoam) {
        const double gate = sqrt(std::max(0.0, h->prm.loam_knn_max_sq));
        if (halo < gate) return fail(h, "loam: the halo must cover the k-NN gate radius (" + std::to_string(gate) + " m, LoamRegister.cpp:59)");
    } else if (h->method == kNdt) {
        const double res = (double)(float)h->prm.ndt_resolution;
        for (int d = 0; d < 3; ++d)
            if (!on_lattice(lo[d], res, 0.0) || !on_lattice(hi[d], res, 0.0)) return fail(h, "ndt: tile bounds must be multiples of ndt_resolution (whole voxels per rank)");
        const double need = (pow2(res) ? 1.0 : 2.0) * res;
        if (halo < need * (1.0 - 1e-12)) return fail(h, "ndt: the halo must hold the DIRECT7 face voxels: >= " + std::to_string(need) + " m at this resolution");
    } else {
        const double res = h->prm.vgicp_resolution;
        for (int d = 0; d < 3; ++d)
            if (!on_lattice(lo[d], res, 0.5) || !on_lattice(hi[d], res, 0.5))
                return fail(h, "vgicp: tile bounds must lie on the voxel lattice, (k + 0.5) * vgicp_resolution (fast_vgicp_voxel.hpp:158-160)");
        if (halo < res) return fail(h, "vgicp: the halo must be at least one voxel (and hold every tile point's 20 nearest neighbours: checked per call)");
    }
    h->use_tile = 1; h->have_halo = true; h->halo = halo;
    for (int d = 0; d < 3; ++d) { h->tile_lo[d] = lo[d]; h->tile_hi[d] = hi[d]; }
    // a target prepared for another tile was checked against another halo
    h->vg_target_ready = false;
    return 0;
}

int pcr_comm_init_host(pcr_handle* h, pcr_allreduce_fn fn, void* user, int rank, int nranks) {
    if (!h) return 1;
    h->err.clear();
    if (!fn) { h->host_ar = nullptr; h->host_ar_user = nullptr; if (!h->comm) { h->nranks = 1; h->rank = 0; } return 0; }
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(h, "bad communicator arguments");
    if (h->comm) return fail(h, "an RCCL communicator is already set on this handle");
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    peer_close(h);      // (one transport at a time)
    h->host_ar = fn; h->host_ar_user = user; h->rank = rank; h->nranks = nranks;
    return 0;
}

int pcr_get_params(const pcr_handle* h, pcr_params* out) {
    if (!h || !out) return 1;
    *out = h->prm;
    return 0;
}

int pcr_set_params(pcr_handle* h, const pcr_params* p) {
    if (!h) return 1;
    h->err.clear();
    if (!p) return fail(h, "params is NULL");
    if (p->struct_size != sizeof(pcr_params)) return fail(h, "pcr_params.struct_size mismatch (start from pcr_get_params or pcr_default_params)");
    if (p->device >= 0 && p->device != h->device) return fail(h, "a handle cannot move to another device");
    if (h->method == kVgicp && p->vgicp_k_corr != 20) return fail(h, "this build supports vgicp_k_corr = 20 (the reference's value) only");
    const pcr_params old = h->prm;
    h->prm = *p;
    h->prm.device = old.device;
    // whatever was derived from a parameter that changed is dropped; the optimiser settings are read at every call
    if (p->loam_knn_max_sq != old.loam_knn_max_sq && h->method == kLoam) { h->have_target = false; h->grid.valid = false; }
    if ((p->ndt_resolution != old.ndt_resolution || p->ndt_min_points != old.ndt_min_points) && h->method == kNdt) { h->nd_target_ready = false; h->have_target = false; h->grid.valid = false; }
    if (p->vgicp_resolution != old.vgicp_resolution && h->method == kVgicp) { h->vg_target_ready = false; h->have_target = false; h->grid.valid = false; }
    return 0;
}

int pcr_fitness_gated(pcr_handle* h, const void* src, size_t n_src, size_t stride_bytes, int on_device, const double pose[16], double max_sq,
                      double* score, int64_t* n_in) {
    if (!h) return 1;
    h->err.clear();
    if (!pose || !score) return fail(h, "pose or score is NULL");
    if (n_src && !src) return fail(h, "NULL cloud with nonzero size");
    if (check_stride(h, stride_bytes) || set_device(h)) return 1;
    if (!h->have_target || !h->grid.valid) return fail(h, "no target: register a scan or call pcr_set_target first");
    if (n_src > 0xfffffff0ull) return fail(h, "source cloud too large");
    if (h->grid.filtered) {
        // The last pcr_scan2map (NDT) indexed only the target points of its scan's region; a nearest-neighbour search needs them all.
        // A target that came in as a HOST buffer still lies in this handle's staging copy and is indexed again, in full; a device buffer
        // is the caller's and may be gone.
        if (h->method != kNdt || h->tgt_ptr != h->tgt_stage.as<float>() || !h->tgt_n)
            return fail(h, "the target index of the last pcr_scan2map holds the scan's region only (pcr_stats.region_index) and the target was a device buffer: "
                           "call pcr_set_target, or set pcr_params.full_target, before asking for a fitness score against it");
        h->nd_target_ready = false;
        if (settle_grid(h, h->grid, h->tgt_ptr, h->tgt_n, h->tgt_stride, (double)(float)h->prm.ndt_resolution, 1)) return 1;
    }
    const float* d_src = (const float*)src;
    if (!on_device && stage_host(h, &h->src_stage, src, n_src, stride_bytes, &d_src)) return 1;
    if (ensure_out32(h)) return 1;
    FitTile ft;
    memset(&ft, 0, sizeof ft);
    if (h->use_tile) {
        ft.use = 1;
        for (int d = 0; d < 3; ++d) { ft.lo[d] = h->tile_lo[d]; ft.hi[d] = h->tile_hi[d]; ft.ext_lo[d] = -1e300; ft.ext_hi[d] = 1e300; }
        if (h->have_halo) shard_extent(h, ft.ext_lo, ft.ext_hi);
    }
    h->seq += 1.0;
    H_TRY(fitness_launch(h->grid, d_src, n_src, stride_bytes / 4, pose, max_sq, h->vg_partials.as<double>(), h->out32_dev, h->stream, h->seq,
                         h->use_tile ? &ft : nullptr));
    if (wait_result(h, &h->out32_host[31], h->seq)) return 1;
    if (sharded(h) && ranks_allreduce(h, h->out32_host, 3)) return 1;
    const double cnt = h->out32_host[1];
    *score = cnt > 0 ? h->out32_host[0] / cnt : -1.0;      // align.cpp:56-59
    if (n_in) *n_in = (int64_t)cnt;
    if (h->use_tile && h->out32_host[2] > 0) return fail(h, "sharded fitness: a source point's nearest map point may lie beyond this rank's halo");
    return 0;
}

// ---- host buffers handed to pcr_scan2map / pcr_set_target / pcr_align: page-locking them is the caller's call ----
// hipMemcpyAsync from pageable memory goes through the runtime's own staging buffers; from a registered range the copy engine reads the
// caller's pages directly.  The registration is keyed on what the caller SAYS -- (pointer, size), pinned until pcr_host_unpin -- never on
// what the library guesses from a pointer it has seen before: a buffer that was freed and allocated again at the same address would be
// read through stale page mappings (the hazard of SURVEY F10, one level down).
int pcr_host_pin(const void* ptr, size_t bytes) {
    g_create_error.clear();
    if (!ptr || !bytes) { g_create_error = "pcr_host_pin: NULL or empty range"; return 1; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const PinnedRange& r : g_pinned)
        if (r.p == ptr) {
            if (r.bytes == bytes) return 0;      // the same range again: nothing to do
            g_create_error = "pcr_host_pin: this address is already pinned with another size (pcr_host_unpin it first)";
            return 1;
        }
    const hipError_t e = hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); g_create_error = std::string("hipHostRegister: ") + hipGetErrorString(e); return 1; }
    g_pinned.push_back(PinnedRange{ptr, bytes, {}});
    return 0;
}

int pcr_host_unpin(const void* ptr) {
    g_create_error.clear();
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pinned.size(); ++i)
        if (g_pinned[i].p == ptr) {
            {   // no copy out of the range may still be in flight: the streams that have copied from it (stage_host notes them), and only those
                int cur = -1;
                const bool have_cur = hipGetDevice(&cur) == hipSuccess;
                for (const PinUse& u : g_pinned[i].uses) {
                    if (hipSetDevice(u.device) != hipSuccess) { (void)hipGetLastError(); continue; }
                    if (hipStreamSynchronize(u.stream) != hipSuccess) (void)hipGetLastError();
                }
                if (have_cur && !g_pinned[i].uses.empty()) (void)hipSetDevice(cur);
            }
            const hipError_t e = hipHostUnregister(const_cast<void*>(ptr));
            g_pinned.erase(g_pinned.begin() + (long)i);
            if (e != hipSuccess) { (void)hipGetLastError(); g_create_error = std::string("hipHostUnregister: ") + hipGetErrorString(e); return 1; }
            return 0;
        }
    g_create_error = "pcr_host_unpin: this address was not pinned with pcr_host_pin";
    return 1;
}

int pcr_comm_unique_id(void* out128) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!out128 || !g_rccl.load(&g_create_error)) return 1;
    NcclId id;
    int rc = g_rccl.get_id(&id);
    if (rc != 0) { g_create_error = "ncclGetUniqueId failed with code " + std::to_string(rc); return 1; }
    memcpy(out128, &id, sizeof(id));
    return 0;
}

int pcr_comm_info(const pcr_handle* h, int* rank, int* nranks, int* transport) {
    if (!h) return 1;
    int r = h->rank, n = h->nranks, t = 0;
    if (h->comm) {
        // what the communicator itself reports, not what the caller passed to pcr_comm_init
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        t = 1;
        if (g_rccl.comm_count && g_rccl.comm_count(h->comm, &n) != 0) return 1;
        if (g_rccl.comm_user_rank && g_rccl.comm_user_rank(h->comm, &r) != 0) return 1;
    } else if (h->host_ar) t = 2;
    else if (h->peer_on) t = 3;
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    if (transport) *transport = t;
    return 0;
}

int pcr_comm_peer_export(pcr_handle* h, void* ipc_handle64) {
    if (!h) return 1;
    h->err.clear();
    if (!ipc_handle64) return fail(h, "ipc_handle64 is NULL");
    if (set_device(h)) return 1;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI says 64 bytes");
    const size_t bytes = (size_t)2 * kMaxPeers * kPeerSlot * sizeof(double);
    if (h->stream) H_TRY(hipStreamSynchronize(h->stream));      // (nothing of an earlier session is in flight)
    if (!h->peer_own) {
        // fine-grained: the peers' stores and this rank's polls meet in memory, not in a cache that nobody invalidates inside a kernel.  No
        // fallback to ordinary (coarse-grained) memory: remote stores might never be seen there, and every exchange would run into its timeout.
        const hipError_t e = hipExtMallocWithFlags((void**)&h->peer_own, bytes, hipDeviceMallocFinegrained);
        if (e != hipSuccess) { (void)hipGetLastError(); h->peer_own = nullptr; return fail(h, std::string("peer exchange: no fine-grained device memory for the receive buffer (") + hipGetErrorString(e) + ")"); }
    }
    if (!h->peer_status_host) {
        H_TRY(hipHostMalloc((void**)&h->peer_status_host, 64, hipHostMallocMapped));
        H_TRY(hipHostGetDevicePointer((void**)&h->peer_status_dev, h->peer_status_host, 0));
    }
    // EVERY export starts a session from nothing: sequence words of an earlier session could otherwise match the new one's.  The ranks share their
    // handles only after every rank has exported (that exchange is the barrier): no peer writes into this buffer before it has been cleared.
    H_TRY(hipMemset(h->peer_own, 0, bytes));
    H_TRY(hipDeviceSynchronize());
    *h->peer_status_host = 0;
    h->peer_exported = true;
    hipIpcMemHandle_t mh;
    H_TRY(hipIpcGetMemHandle(&mh, h->peer_own));
    memcpy(ipc_handle64, &mh, 64);
    return 0;
}

int pcr_comm_init_peer(pcr_handle* h, const void* ipc_handles, int rank, int nranks) {
    if (!h) return 1;
    h->err.clear();
    if (!ipc_handles || nranks < 1 || nranks > kMaxPeers || rank < 0 || rank >= nranks) return fail(h, "bad peer-exchange arguments (at most 8 ranks)");
    if (!h->peer_own || !h->peer_exported) return fail(h, "call pcr_comm_peer_export first (every rank, for every session), then share the handles");
    if (h->comm) return fail(h, "an RCCL communicator is already set on this handle");
    if (set_device(h)) return 1;
    if (h->stream) H_TRY(hipStreamSynchronize(h->stream));
    peer_close(h);      // (the mappings of an earlier session)
    h->peer_exported = false;
    for (int p = 0; p < nranks; ++p) {
        if (p == rank) { h->peer.buf[p] = h->peer_own; continue; }
        hipIpcMemHandle_t mh;
        memcpy(&mh, (const char*)ipc_handles + (size_t)p * 64, 64);
        void* mapped = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&mapped, mh, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int q = 0; q < p; ++q) if (q != rank && h->peer.buf[q]) (void)hipIpcCloseMemHandle(h->peer.buf[q]);
            memset(&h->peer, 0, sizeof h->peer);
            return fail(h, std::string("hipIpcOpenMemHandle of rank ") + std::to_string(p) + "'s receive buffer: " + hipGetErrorString(e));
        }
        h->peer.buf[p] = (double*)mapped;
    }
    h->peer.rank = rank; h->peer.nranks = nranks; h->peer.status = h->peer_status_dev;
    h->rank = rank; h->nranks = nranks;
    h->peer_seq = 0.0;
    h->peer_on = true; h->peer_broken = false;
    h->host_ar = nullptr; h->host_ar_user = nullptr;
    return 0;
}

int pcr_comm_init(pcr_handle* h, const void* unique_id128, int rank, int nranks) {
    if (!h) return 1;
    h->err.clear();
    if (!unique_id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, "bad communicator arguments");
    if (set_device(h)) return 1;
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.load(&h->err)) return 1;
    NcclId id;
    memcpy(&id, unique_id128, sizeof(id));
    if (h->comm) return fail(h, "an RCCL communicator is already set on this handle");
    int rc = g_rccl.init_rank(&h->comm, nranks, id, rank);
    if (rc != 0) { h->comm = nullptr; return fail(h, "ncclCommInitRank failed with code " + std::to_string(rc)); }
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    peer_close(h);      // (one transport at a time)
    h->nranks = nranks; h->rank = rank;
    h->host_ar = nullptr; h->host_ar_user = nullptr;
    return 0;
}

}  // extern "C"
