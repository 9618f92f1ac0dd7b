// submap.hip -- MapManager's key-frame store and sub-map assembly on the device (gfx950).
//
// The producer of `dst` for every scan2Map call:
//   MapManager::updateMap            frontend/src/MapManager.cpp:151-201
//     key frames within mSurroundingKeyframeSearchRadius (8 m, MapManager.hpp:68) of the current position:
//     KeyFramesKdtree::radiusSearch  third_parties/nanoflann/include/nanoflann/kfs_adaptor.hpp:57-75  (squared L2 in
//     double, strict '<' as nanoflann's RadiusResultSet accepts)
//     per key frame pcp::transformPointCloud(pose cast to float)   common/pcp/pcp.hpp:38-62   -> concatenate
//     pcp::voxelDownSample(mSubmap, mGridSize)                     common/pcp/pcp.hpp:14-20   (pcl::VoxelGrid)
// Key frames stay in HBM; an update is one transform+concatenate launch plus pcr_voxel_filter, and the sub-map it
// leaves in HBM is what pcr_scan2map_device takes as `dst` -- no host copy of the map on the registration path.
// The result does not depend on the order in which the key frames are concatenated (centroids are summed in f64).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/pcr_hip.h"

// (library-internal, defined in capi.hip) the stream a handle queues its work on; the voxel filter in two halves (queue; synchronise and collect)
hipStream_t pcr_internal_stream(const pcr_handle* h);
int pcr_internal_vf_begin(pcr_handle* h, const void* d_pts, size_t n, size_t stride_bytes, double leaf, void* d_out, size_t out_capacity);
int pcr_internal_vf_end(pcr_handle* h, size_t* n_out);
void pcr_internal_vf_reserve(pcr_handle* h, size_t points);

namespace {

struct KfDesc {              // one selected key frame in the concatenation
    unsigned long long src_off;   // first float of its points in the store
    unsigned int first_out;       // first output point
    unsigned int n;
    float R[9], t[3];             // pose_t cast to float (pcp.hpp:41-46), row-major rotation
};

// One thread per output point.  pto = tr * pfrom in float: ((r0 x + r1 y) + r2 z) + t per row -- Eigen's coefficient-based
// 3x3 * 3x1 product followed by the translation; compiled without FMA contraction so that the voxel a point falls
// into is the one the CPU computes.
__device__ inline void submap_transform_point(const float* __restrict__ store, const KfDesc* __restrict__ kf, int n_kf, unsigned int i, unsigned int stride,
                                              float* __restrict__ out) {
    int lo = 0, hi = n_kf - 1;                    // last key frame whose first_out <= i
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (kf[mid].first_out <= i) lo = mid; else hi = mid - 1; }
    const KfDesc d = kf[lo];
    const float* p = store + d.src_off + (size_t)(i - d.first_out) * stride;
    float* o = out + (size_t)i * stride;
    const float x = p[0], y = p[1], z = p[2];
    o[0] = ((d.R[0] * x + d.R[1] * y) + d.R[2] * z) + d.t[0];
    o[1] = ((d.R[3] * x + d.R[4] * y) + d.R[5] * z) + d.t[1];
    o[2] = ((d.R[6] * x + d.R[7] * y) + d.R[8] * z) + d.t[2];
    for (unsigned int c = 3; c < stride; ++c) o[c] = p[c];          // data[3] = 1 and the intensity travel unchanged
}

__global__ __launch_bounds__(256) void submap_transform_kernel(const float* __restrict__ store, const KfDesc* __restrict__ kf, int n_kf,
                                                               unsigned int n_total, unsigned int stride, float* __restrict__ out) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_total) submap_transform_point(store, kf, n_kf, i, stride, out);
}

// ... the same with the descriptors in the kernel's ARGUMENTS (up to 48 key frames: 3 KB of the 4 KB a launch carries): no copy in front of the launch,
// no buffer to keep alive behind it
static constexpr int kPackMax = 48;
struct KfPack { KfDesc d[kPackMax]; };
__global__ __launch_bounds__(256) void submap_transform_pack_kernel(const float* __restrict__ store, const KfPack pack, int n_kf,
                                                                    unsigned int n_total, unsigned int stride, float* __restrict__ out) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_total) submap_transform_point(store, pack.d, n_kf, i, stride, out);
}

struct Buf {
    void* p = nullptr; size_t cap = 0;
    // keep: the contents move along (the key-frame store); the other buffers are written anew by every assembly
    hipError_t reserve(size_t bytes, bool keep = true) {
        if (bytes <= cap) return hipSuccess;
        void* q = nullptr;
        const size_t want = 2 * bytes + 4096;      // (doubling: every growth is a device-wide stop -- allocation, copy, free -- and a session's first minutes are made of them)
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess) return e;
        if (p && !keep) (void)hipFree(p);
        else if (p) { e = hipMemcpy(q, p, cap, hipMemcpyDeviceToDevice); (void)hipFree(p); if (e != hipSuccess) { (void)hipFree(q); p = nullptr; cap = 0; return e; } }
        p = q; cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

static constexpr int kWindow = 16;      // key frames a sub-map is expected to hold (8 m radius, a key frame per metre: MapManager.hpp:67-68)

struct pcr_map {
    int device = 0;
    pcr_handle* filter = nullptr;     // device context of the voxel filter
    size_t stride = 0;                // bytes per point, fixed by the first key frame
    Buf store, concat, submap, desc;
    KfDesc* desc_host = nullptr;      // page-locked staging of the descriptors
    size_t desc_host_cap = 0;
    size_t store_floats = 0;
    struct Kf { size_t off_floats, n; double pose[16]; };
    std::vector<Kf> kfs;
    std::vector<long long> selected;  // mSubmapIdx
    size_t n_submap = 0;
    size_t max_kf_points = 0;         // the largest key frame so far: buffers are sized for a window of kWindow such key frames from the start
    bool pending = false;             // an assembly is queued on the filter's stream and has not been collected (pcr_map_update_begin): n_submap is not known yet
    uint64_t id = 0, generation = 0;  // identity of this store and of the sub-map it currently holds (every update is a new generation)
    std::string err;
};

static thread_local std::string g_map_err;
static int mfail(pcr_map* m, const std::string& s) { if (m) m->err = s; else g_map_err = s; return 1; }
#define M_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return mfail(m, std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

// collect the queued assembly, if any: the one synchronisation of an update (and the filter's second try when the index's hints did not hold)
static int finish_pending(pcr_map* m) {
    if (!m->pending) return 0;
    m->pending = false;
    m->n_submap = 0;
    M_TRY(hipSetDevice(m->device));
    size_t n_out = 0;
    if (pcr_internal_vf_end(m->filter, &n_out)) return mfail(m, std::string("voxel filter: ") + pcr_last_error(m->filter));
    m->n_submap = n_out;
    return 0;
}

extern "C" {

pcr_map* pcr_map_create(int device) {
    pcr_params p;
    pcr_default_params(&p);
    p.device = device;
    pcr_handle* h = pcr_create("loam", &p);
    if (!h) { g_map_err = pcr_last_error(nullptr); return nullptr; }
    static std::atomic<uint64_t> next_id{1};
    pcr_map* m = new pcr_map;
    m->id = next_id.fetch_add(1);
    m->filter = h;
    if (device >= 0) m->device = device; else (void)hipGetDevice(&m->device);
    return m;
}

void pcr_map_destroy(pcr_map* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    pcr_destroy(m->filter);           // (first: it waits for whatever is still queued on its stream -- an assembly nobody collected reads the buffers below)
    m->store.release(); m->concat.release(); m->submap.release(); m->desc.release();
    if (m->desc_host) (void)hipHostFree(m->desc_host);
    delete m;
}

const char* pcr_map_last_error(const pcr_map* m) { return m ? m->err.c_str() : g_map_err.c_str(); }

int pcr_map_add_keyframe(pcr_map* m, const void* pts, size_t n, size_t stride_bytes, int on_device, const double pose[16]) {
    if (!m) return 1;
    m->err.clear();
    if (n && !pts) return mfail(m, "NULL cloud with nonzero size");
    if (!pose) return mfail(m, "NULL pose");
    if (stride_bytes < 12 || stride_bytes % 4) return mfail(m, "stride_bytes must be a multiple of 4 and >= 12");
    if (m->stride == 0) m->stride = stride_bytes;
    else if (m->stride != stride_bytes) return mfail(m, "all key frames of a map must share one point layout");
    M_TRY(hipSetDevice(m->device));
    if (finish_pending(m)) return 1;      // (a queued assembly reads the store, which may move below)
    const size_t nf = n * (stride_bytes / 4);
    m->max_kf_points = std::max(m->max_kf_points, n);
    // (room for a window of key frames from the first one on: a store that grows is copied, and every growth is a device-wide stop)
    M_TRY(m->store.reserve((std::max(m->store_floats + nf, (size_t)kWindow * nf) + 4) * sizeof(float)));
    if (nf && on_device) {      // on the filter's stream, waited for: the caller may reuse its buffer when this returns (the blocking copy on the null stream took 60 us for 1 MB, round 5)
        hipStream_t fs = pcr_internal_stream(m->filter);
        M_TRY(hipMemcpyAsync(static_cast<float*>(m->store.p) + m->store_floats, pts, nf * sizeof(float), hipMemcpyDeviceToDevice, fs));
        M_TRY(hipStreamSynchronize(fs));
    } else if (nf) M_TRY(hipMemcpy(static_cast<float*>(m->store.p) + m->store_floats, pts, nf * sizeof(float), hipMemcpyHostToDevice));
    pcr_map::Kf k;
    k.off_floats = m->store_floats; k.n = n;
    for (int i = 0; i < 16; ++i) k.pose[i] = pose[i];
    m->kfs.push_back(k);
    m->store_floats += nf;
    return 0;
}

int pcr_map_clear(pcr_map* m) {
    if (!m) return 1;
    m->err.clear();
    M_TRY(hipSetDevice(m->device));
    (void)finish_pending(m); m->err.clear();
    m->kfs.clear(); m->selected.clear();
    m->store_floats = 0; m->n_submap = 0;
    m->generation += 1;           // whatever was built from the previous sub-map is stale
    return 0;
}

int pcr_map_keyframes(const pcr_map* m, size_t* n_keyframes) {
    if (!m || !n_keyframes) return 1;
    *n_keyframes = m->kfs.size();
    return 0;
}

// transform + concatenate the selected key frames (m->selected, ascending) and voxel-filter the result into m->submap
static int assemble_selected(pcr_map* m, double grid_size, size_t* n_submap, bool wait = true) {
    std::vector<KfDesc> desc;
    size_t total = 0;
    for (long long i : m->selected) {
        const pcr_map::Kf& k = m->kfs[(size_t)i];
        if (k.n == 0) continue;
        if (total + k.n > 0xfffffff0ull) return mfail(m, "sub-map too large");
        KfDesc e;
        e.src_off = k.off_floats; e.first_out = (unsigned int)total; e.n = (unsigned int)k.n;
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) e.R[r * 3 + c] = (float)k.pose[c * 4 + r]; e.t[r] = (float)k.pose[12 + r]; }
        desc.push_back(e);
        total += k.n;
    }
    if (total == 0) return 0;
    const size_t sf = m->stride / 4;
    const size_t room = std::max(total, (size_t)kWindow * m->max_kf_points);      // (the concatenation of a window of key frames: the buffers, the filter's and its index's, start at that size)
    M_TRY(m->concat.reserve(room * m->stride, false));
    M_TRY(m->submap.reserve(room * m->stride, false));
    pcr_internal_vf_reserve(m->filter, room);
    M_TRY(m->desc.reserve(desc.size() * sizeof(KfDesc), false));
    // the descriptors and the transform pass go onto the FILTER's stream, in front of the filter's own launches: one synchronisation -- the filter's,
    // for its voxel count -- serves the whole assembly (round 4: a blocking copy, the pass on the null stream, a device-wide synchronisation, then the
    // filter with two more: 0.49 ms for a 500 k-point concatenation of which a third was waiting; round 5, scripts/seq_breakdown.py)
    hipStream_t fs = pcr_internal_stream(m->filter);
    if (desc.size() <= (size_t)kPackMax) {
        KfPack pack;
        memcpy(pack.d, desc.data(), desc.size() * sizeof(KfDesc));
        hipLaunchKernelGGL(submap_transform_pack_kernel, dim3((unsigned int)((total + 255) / 256)), dim3(256), 0, fs, static_cast<const float*>(m->store.p),
                           pack, (int)desc.size(), (unsigned int)total, (unsigned int)sf, static_cast<float*>(m->concat.p));
    } else {
    if (m->desc_host_cap < desc.size()) {
        if (m->desc_host) (void)hipHostFree(m->desc_host);
        m->desc_host = nullptr; m->desc_host_cap = 0;
        M_TRY(hipHostMalloc((void**)&m->desc_host, (desc.size() + 64) * sizeof(KfDesc), hipHostMallocDefault));
        m->desc_host_cap = desc.size() + 64;
    }
    memcpy(m->desc_host, desc.data(), desc.size() * sizeof(KfDesc));      // (page-locked staging: the copy below does not block; the previous assembly's has completed -- its filter synchronised)
    M_TRY(hipMemcpyAsync(m->desc.p, m->desc_host, desc.size() * sizeof(KfDesc), hipMemcpyHostToDevice, fs));
    hipLaunchKernelGGL(submap_transform_kernel, dim3((unsigned int)((total + 255) / 256)), dim3(256), 0, fs, static_cast<const float*>(m->store.p),
                       static_cast<const KfDesc*>(m->desc.p), (int)desc.size(), (unsigned int)total, (unsigned int)sf, static_cast<float*>(m->concat.p));
    }
    M_TRY(hipGetLastError());
    if (pcr_internal_vf_begin(m->filter, m->concat.p, total, m->stride, grid_size, m->submap.p, total))
        return mfail(m, std::string("voxel filter: ") + pcr_last_error(m->filter));
    m->pending = true;
    if (!wait) return 0;
    if (finish_pending(m)) return 1;
    if (n_submap) *n_submap = m->n_submap;
    return 0;
}

static int update_by_radius(pcr_map* m, const double position[3], double radius, double grid_size, size_t* n_submap, bool wait) {
    if (!m) return 1;
    m->err.clear();
    if (!position) return mfail(m, "NULL position");
    if (!(grid_size > 0)) return mfail(m, "grid_size must be positive");
    M_TRY(hipSetDevice(m->device));
    (void)finish_pending(m); m->err.clear();      // (an assembly nobody collected: waited for -- its buffers are about to be written again -- and superseded, whatever became of it)
    m->selected.clear();
    m->n_submap = 0;
    m->generation += 1;           // whatever happens below, the previous sub-map is gone
    if (n_submap) *n_submap = 0;
    if (m->kfs.empty()) return 0;                          // "no any keyframes to update!!" (MapManager.cpp:166-169)
    // radius search over the key-frame positions: squared distance in double, accumulated x, y, z; strict '<'
    const double r2 = radius * radius;
    for (size_t i = 0; i < m->kfs.size(); ++i) {
        double d = 0;
        for (int c = 0; c < 3; ++c) { const double e = position[c] - m->kfs[i].pose[12 + c]; d += e * e; }
        if (d < r2) m->selected.push_back((long long)i);
    }
    return assemble_selected(m, grid_size, n_submap, wait);
}

int pcr_map_update(pcr_map* m, const double position[3], double radius, double grid_size, size_t* n_submap) {
    return update_by_radius(m, position, radius, grid_size, n_submap, true);
}

int pcr_map_update_begin(pcr_map* m, const double position[3], double radius, double grid_size) {
    return update_by_radius(m, position, radius, grid_size, nullptr, false);
}

int pcr_map_wait(pcr_map* m, size_t* n_submap) {
    if (!m) return 1;
    if (n_submap) *n_submap = 0;
    if (m->pending) m->err.clear();
    if (finish_pending(m)) return 1;
    if (n_submap) *n_submap = m->n_submap;
    return 0;
}

int pcr_map_update_window(pcr_map* m, long long key, int search_num, double grid_size, size_t* n_submap) {
    if (!m) return 1;
    m->err.clear();
    if (!(grid_size > 0)) return mfail(m, "grid_size must be positive");
    if (search_num < 0) return mfail(m, "search_num must not be negative");
    M_TRY(hipSetDevice(m->device));
    (void)finish_pending(m); m->err.clear();
    m->selected.clear();
    m->n_submap = 0;
    m->generation += 1;           // whatever happens below, the previous sub-map is gone
    if (n_submap) *n_submap = 0;
    const long long count = (long long)m->kfs.size();
    for (long long i = -(long long)search_num; i <= (long long)search_num; ++i) {      // LoopClosureManager.cpp:46-57
        const long long near = key + i;
        if (near >= 0 && near < count) m->selected.push_back(near);
    }
    return assemble_selected(m, grid_size, n_submap);
}

const void* pcr_map_submap(const pcr_map* m, size_t* n, size_t* stride_bytes) {
    if (!m) return nullptr;
    if (m->pending && finish_pending(const_cast<pcr_map*>(m))) { if (n) *n = 0; if (stride_bytes) *stride_bytes = m->stride; return nullptr; }      // (the queued assembly is collected by whoever asks for the sub-map first; its error is pcr_map_last_error's)
    if (n) *n = m->n_submap;
    if (stride_bytes) *stride_bytes = m->stride;
    return m->n_submap ? m->submap.p : nullptr;
}

int pcr_map_generation(const pcr_map* m, uint64_t* id, uint64_t* generation) {
    if (!m) return 1;
    if (id) *id = m->id;
    if (generation) *generation = m->generation;
    return 0;
}

int pcr_map_submap_indices(const pcr_map* m, int64_t* idx, size_t capacity, size_t* n) {
    if (!m || !n) return 1;
    *n = m->selected.size();
    if (!idx) return 0;
    if (capacity < m->selected.size()) return 1;
    for (size_t i = 0; i < m->selected.size(); ++i) idx[i] = m->selected[i];
    return 0;
}

}  // extern "C"
