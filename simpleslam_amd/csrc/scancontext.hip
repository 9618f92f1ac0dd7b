// scancontext.hip -- the loop-closure descriptor of the reference's back end:
//   backend/src/ScanContext.cpp (adapted there from irapkaist/scancontext), backend/include/backend/ScanContext.hpp
// makeScanContext (:151-196) -- the per-point part: polar binning of a down-sampled scan into 20 rings x 60 sectors over
// 80 m, maximum of z + lidar_height per bin -- is a device kernel (float atomic max: order-independent, so the
// descriptor is bit-identical to a sequential evaluation).  Everything after it works on 20 x 60 doubles per key frame
// and stays host C++, restated line by line: ring key / sector key (:198-229), fastAlignUsingVkey (:94-114),
// distanceBtnScanContext (:116-150), computeSimularity (:68-92), query with its lazily rebuilt candidate tree (:231-279;
// an exact 10-NN over the 20-dimensional ring keys, by brute force here instead of nanoflann's VectorOfVectorsKdTree).
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>
#include <limits>
#include <string>
#include <vector>

#include "../../include/pcr_hip.h"

namespace {

constexpr int kRings = 20, kSectors = 60;            // ScanContext.hpp:17-18
constexpr float kMaxRadius = 80.0f;                  // ScanContext.hpp:19
constexpr float kNoPoint = -1000.0f;                 // ScanContext.cpp:157

// bin of one point, arithmetic type for type as the reference writes it (ScanContext.cpp:27-32,163-181)
__host__ __device__ inline bool sc_bin(float x, float y, int* ring, int* sector) {
    const float azim_range = sqrtf(x * x + y * y);
    float res = (float)((double)atan2f(y, x) + 3.14159265358979323846);          // xy2theta: float atan2 + M_PI, stored as float
    res = fmaxf(0.0f, fminf((float)(2 * 3.14159265358979323846), res));
    const float azim_angle = (float)((double)res * 180.0 / 3.14159265358979323846);   // trans::rad2deg<float>
    if (azim_range > kMaxRadius) return false;
    const int r = (int)ceilf((azim_range / kMaxRadius) * kRings);
    const int s = (int)ceil(((double)azim_angle / 360.0) * kSectors);
    *ring = max(min(kRings, r), 1) - 1;
    *sector = max(min(kSectors, s), 1) - 1;
    return true;
}

__global__ __launch_bounds__(256) void sc_polar_kernel(const float* __restrict__ pts, unsigned int n, unsigned int stride, float lidar_height,
                                                       float* __restrict__ desc /* [ring][sector], preset to kNoPoint */) {
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float* p = pts + (size_t)i * stride;
        const float x = p[0], y = p[1], z = p[2] + lidar_height;
        int ring, sector;
        if (!(isfinite(x) && isfinite(y) && z == z)) continue;   // the reference's int(ceil(NaN)) is undefined behaviour; a NaN z never wins its '<'
        if (!sc_bin(x, y, &ring, &sector)) continue;
        atomicMax(&desc[ring * kSectors + sector], z);
    }
}

struct Desc { double m[kRings * kSectors]; };        // row-major [ring][sector]

}  // namespace

struct pcr_sc {
    int device = 0;
    pcr_sc_params prm;
    std::vector<Desc> polar;
    std::vector<std::vector<double>> ring, sector;
    size_t tree_size = 0;                                // ring_sub_.size(): contexts visible to the candidate search
    float* d_desc = nullptr;
    float* d_stage = nullptr; size_t stage_cap = 0;
    std::string err;
};

static thread_local std::string g_sc_err;
static int scfail(pcr_sc* s, const std::string& m) { if (s) s->err = m; else g_sc_err = m; return 1; }
#define S_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return scfail(sc, std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

namespace {

double col_norm(const Desc& d, int col) { double s = 0; for (int r = 0; r < kRings; ++r) s += d.m[r * kSectors + col] * d.m[r * kSectors + col]; return sqrt(s); }

// computeSimularity of sc1 and sc2 circularly shifted by `shift` columns to the right (ScanContext.cpp:34-54,68-92)
double sc_distance(const Desc& a, const Desc& b, int shift) {
    int eff = 0;
    double sum = 0;
    for (int col = 0; col < kSectors; ++col) {
        const int src = ((col - shift) % kSectors + kSectors) % kSectors;      // shifted(col) = b(col - shift)
        double na = 0, nb = 0, dot = 0;
        for (int r = 0; r < kRings; ++r) {
            const double x = a.m[r * kSectors + col], y = b.m[r * kSectors + src];
            na += x * x; nb += y * y; dot += x * y;
        }
        na = sqrt(na); nb = sqrt(nb);
        if (na == 0 || nb == 0) continue;
        sum = sum + dot / (na * nb);
        eff = eff + 1;
    }
    return 1.0 - sum / eff;                              // eff == 0 gives NaN, as in the reference
}

int fast_align(const std::vector<double>& k1, const std::vector<double>& k2) {        // :94-114 (keys as 1 x 60 rows)
    int arg = 0;
    double best = std::numeric_limits<double>::max();
    for (int shift = 0; shift < kSectors; ++shift) {
        double s = 0;
        for (int c = 0; c < kSectors; ++c) { const double d = k1[c] - k2[((c - shift) % kSectors + kSectors) % kSectors]; s += d * d; }
        const double nrm = sqrt(s);
        if (nrm < best) { arg = shift; best = nrm; }
    }
    return arg;
}

}  // namespace

extern "C" {

void pcr_sc_default_params(pcr_sc_params* p) {
    if (!p) return;
    p->lidar_height = 2.0;            /* config/params.json: tf.lidar_height */
    p->num_exclude_recent = 40;       /* backend.context.scancontext.* */
    p->build_tree_gap = 10;
    p->num_candidates = 10;
    p->search_ratio = 0.1;
    p->dist_thres = 0.4;
}

pcr_sc* pcr_sc_create(int device, const pcr_sc_params* p) {
    pcr_sc* sc = new pcr_sc;
    if (p) sc->prm = *p; else pcr_sc_default_params(&sc->prm);
    if (device >= 0) sc->device = device; else (void)hipGetDevice(&sc->device);
    if (hipSetDevice(sc->device) != hipSuccess || hipMalloc((void**)&sc->d_desc, kRings * kSectors * sizeof(float)) != hipSuccess) {
        g_sc_err = "pcr_sc_create: no usable HIP device (there is no CPU fallback)";
        delete sc;
        return nullptr;
    }
    return sc;
}

void pcr_sc_destroy(pcr_sc* sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    if (sc->d_desc) (void)hipFree(sc->d_desc);
    if (sc->d_stage) (void)hipFree(sc->d_stage);
    delete sc;
}

const char* pcr_sc_last_error(const pcr_sc* sc) { return sc ? sc->err.c_str() : g_sc_err.c_str(); }

int pcr_sc_size(const pcr_sc* sc, size_t* n) { if (!sc || !n) return 1; *n = sc->polar.size(); return 0; }

/* addContext (:56-66): descriptor of a (down-sampled) scan in the lidar frame + its ring and sector keys */
int pcr_sc_add(pcr_sc* sc, const void* pts, size_t n, size_t stride_bytes, int on_device) {
    if (!sc) return 1;
    sc->err.clear();
    if (n && !pts) return scfail(sc, "NULL cloud with nonzero size");
    if (stride_bytes < 12 || stride_bytes % 4) return scfail(sc, "stride_bytes must be a multiple of 4 and >= 12");
    if (n > 0xfffffff0ull) return scfail(sc, "cloud too large");
    S_TRY(hipSetDevice(sc->device));
    const float* d_pts = static_cast<const float*>(pts);
    if (!on_device && n) {
        const size_t bytes = n * stride_bytes;
        if (bytes > sc->stage_cap) {
            if (sc->d_stage) (void)hipFree(sc->d_stage);
            sc->d_stage = nullptr; sc->stage_cap = 0;
            S_TRY(hipMalloc((void**)&sc->d_stage, bytes + bytes / 2));
            sc->stage_cap = bytes + bytes / 2;
        }
        S_TRY(hipMemcpy(sc->d_stage, pts, bytes, hipMemcpyHostToDevice));
        d_pts = sc->d_stage;
    }
    float init[kRings * kSectors];
    for (float& v : init) v = kNoPoint;
    S_TRY(hipMemcpy(sc->d_desc, init, sizeof init, hipMemcpyHostToDevice));
    if (n) {
        const int blocks = (int)std::min<size_t>(1024, (n + 255) / 256);
        hipLaunchKernelGGL(sc_polar_kernel, dim3(blocks), dim3(256), 0, 0, d_pts, (unsigned int)n, (unsigned int)(stride_bytes / 4),
                           (float)sc->prm.lidar_height, sc->d_desc);
        S_TRY(hipGetLastError());
    }
    float host[kRings * kSectors];
    S_TRY(hipMemcpy(host, sc->d_desc, sizeof host, hipMemcpyDeviceToHost));
    Desc d;
    for (int i = 0; i < kRings * kSectors; ++i) d.m[i] = host[i] == kNoPoint ? 0.0 : (double)host[i];      // :186-190
    std::vector<double> rk(kRings), sk(kSectors);
    for (int r = 0; r < kRings; ++r) { double s = 0; for (int c = 0; c < kSectors; ++c) s += d.m[r * kSectors + c]; rk[r] = s / kSectors; }
    for (int c = 0; c < kSectors; ++c) { double s = 0; for (int r = 0; r < kRings; ++r) s += d.m[r * kSectors + c]; sk[c] = s / kRings; }
    sc->polar.push_back(d); sc->ring.push_back(rk); sc->sector.push_back(sk);
    return 0;
}

int pcr_sc_descriptor(const pcr_sc* sc, size_t id, double* desc_row_major_20x60, double* ring_key_20, double* sector_key_60) {
    if (!sc || id >= sc->polar.size()) return 1;
    if (desc_row_major_20x60) std::copy(sc->polar[id].m, sc->polar[id].m + kRings * kSectors, desc_row_major_20x60);
    if (ring_key_20) std::copy(sc->ring[id].begin(), sc->ring[id].end(), ring_key_20);
    if (sector_key_60) std::copy(sc->sector[id].begin(), sc->sector[id].end(), sector_key_60);
    return 0;
}

/* distanceBtnScanContext (:116-150): minimum over the shifts around the sector-key alignment */
int pcr_sc_distance(const pcr_sc* sc, size_t id1, size_t id2, double* dist, int* shift) {
    if (!sc || id1 >= sc->polar.size() || id2 >= sc->polar.size()) return 1;
    const int a0 = fast_align(sc->sector[id1], sc->sector[id2]);
    const int radius = (int)round(0.5 * (double)(float)sc->prm.search_ratio * kSectors);
    std::vector<int> space{a0};
    for (int ii = 1; ii < radius + 1; ++ii) { space.push_back((a0 + ii + kSectors) % kSectors); space.push_back((a0 - ii + kSectors) % kSectors); }
    std::sort(space.begin(), space.end());
    int arg = 0;
    double best = std::numeric_limits<double>::max();
    for (int s : space) {
        const double d = sc_distance(sc->polar[id1], sc->polar[id2], s);
        if (d < best) { arg = s; best = d; }
    }
    if (dist) *dist = best;
    if (shift) *shift = arg;
    return 0;
}

/* query (:231-279): *match = -1 when there is no loop candidate; yaw = deg2rad(6 deg * shift) as a float */
int pcr_sc_query(pcr_sc* sc, long long id, long long* match, float* yaw_rad, double* min_dist_out) {
    if (!sc || !match) return 1;
    sc->err.clear();
    *match = -1;
    if (yaw_rad) *yaw_rad = 0.f;
    if (min_dist_out) *min_dist_out = std::numeric_limits<double>::max();
    if (id < 0 || (size_t)id >= sc->polar.size()) return scfail(sc, "no such context");
    const long long excl = sc->prm.num_exclude_recent, ncand = sc->prm.num_candidates;
    if (id <= excl + ncand) return 0;
    // the candidate set is a snapshot, refreshed only every build_tree_gap contexts (:240-248)
    if (sc->tree_size == 0 || (size_t)id - sc->tree_size > (size_t)(excl + sc->prm.build_tree_gap)) sc->tree_size = (size_t)(id - excl);
    // exact k nearest ring keys (squared L2, ties on the lower index)
    std::vector<std::pair<double, size_t>> cand;
    for (size_t i = 0; i < sc->tree_size; ++i) {
        double s = 0;
        for (int r = 0; r < kRings; ++r) { const double d = sc->ring[(size_t)id][r] - sc->ring[i][r]; s += d * d; }
        cand.emplace_back(s, i);
    }
    const size_t k = std::min<size_t>((size_t)ncand, cand.size());
    std::partial_sort(cand.begin(), cand.begin() + k, cand.end());
    double min_dist = std::numeric_limits<double>::max();
    int nn_align = 0;
    size_t nn_idx = 0;
    for (size_t c = 0; c < k; ++c) {
        double d; int sh;
        pcr_sc_distance(sc, (size_t)id, cand[c].second, &d, &sh);
        if (d < min_dist) { min_dist = d; nn_align = sh; nn_idx = cand[c].second; }
    }
    if (min_dist_out) *min_dist_out = min_dist;
    if (min_dist > (double)(float)sc->prm.dist_thres) return 0;
    *match = (long long)nn_idx;
    if (yaw_rad) *yaw_rad = (float)((double)((360.0f / (float)kSectors) * (float)nn_align) * 3.14159265358979323846 / 180.0);      // deg2rad<float>
    return 0;
}

}  // extern "C"
