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

// atan2f as glibc computes it.  xy2theta (ScanContext.cpp:28-33) calls std::atan2 on floats, i.e. the C library's atan2f, and a point
// whose azimuth is within an ulp of a sector edge changes bins with the last bit of that result.  glibc's atan2f (2.31 ... 2.35, the
// releases under the reference's ROS targets) is the fdlibm routine in plain float arithmetic -- it is NOT correctly rounded (16 % of
// arguments differ from the rounded double atan2), so the device's own libm cannot stand in for it.  The routine below is that
// published algorithm (Sun fdlibm e_atan2f / s_atanf: argument reduction against atan(0.5), atan(1), atan(1.5), atan(inf) with
// split constants, an odd polynomial of degree 21); built without FMA contraction it reproduced this image's glibc bit for bit on
// 20.4 million arguments (a 1/4 m lattice and random points).  Only IEEE float +, -, *, / are involved, which the GPU rounds identically.
__host__ __device__ inline float sc_atanf(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    union { float f; int i; } u;
    u.f = x;
    const int hx = u.i, ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c800000) {                      // |x| >= 2^26
        if (ix > 0x7f800000) return x + x;       // NaN
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                       // |x| < 0.4375
        if (ix < 0x31000000 && 1.0e30f + x > 1.0f) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {                   // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
        else { id = 3; x = -1.0f / x; }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = (id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3]) -
        ((x * (s1 + s2) - (id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3])) - x);
    return hx < 0 ? -z : z;
}
__host__ __device__ inline float sc_atan2f(float y, float x) {
    const float pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f, tiny = 1.0e-30f;
    union { float f; int i; } ux, uy;
    ux.f = x; uy.f = y;
    const int hx = ux.i, ix = hx & 0x7fffffff, hy = uy.i, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return sc_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = sc_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// bin of one point, arithmetic type for type as the reference writes it (ScanContext.cpp:27-32,163-181)
__host__ __device__ inline bool sc_bin(float x, float y, int* ring, int* sector) {
    const float azim_range = sqrtf(x * x + y * y);
    float res = (float)((double)sc_atan2f(y, x) + 3.14159265358979323846);       // xy2theta: float atan2 (the C library's: above) + M_PI, stored as float
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
