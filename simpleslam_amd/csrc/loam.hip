// loam.hip -- LOAM scan-to-map Gauss-Newton on gfx950 (hand-written HIP).
//
// One launch = one Gauss-Newton iteration of the reference's
//   PCR::LoamRegister::scan2Map            PCR/src/LoamRegister.cpp:99-223
// with its per-point body fused into a single kernel:
//   L2 transform + f32 cast                LoamRegister.cpp:126-130
//   L3 exact 5-NN + squared-distance gate  LoamRegister.cpp:47-72 (nanoflann there, uniform grid here)
//   L4 plane LS (col-piv Householder QR)   LoamRegister.cpp:29-45
//   L5 residual, weight, weight gate       LoamRegister.cpp:144-151, LoamRegister.hpp:75-77
//   L6 Jacobian row s*n^T [I | -p^]        LoamRegister.cpp:153-159, manifolds.hpp:63-68
//   L7 J^T J / J^T r reduction             LoamRegister.cpp:170-188 (omp critical + dense GEMM there)
// and, in the PROLOGUE of the next launch, executed redundantly by every block
// (bitwise identical, so no broadcast and no extra kernel boundary):
//   L8 LDLT solve, convergence test, SE(3) exp update   LoamRegister.cpp:198-216, manifolds.hpp:33-60
// The last launch (loam_finalize_kernel) runs only that prologue plus
//   L9 T2SE3 re-orthonormalisation        LoamRegister.cpp:220, trans.hpp:54-65
//
// Arithmetic is f64 with contraction off (-ffp-contract=off), in the reference's
// operation order, so per-point results are bit-identical to the CPU oracle
// wherever only IEEE +,-,*,/,sqrt are involved.
#include <hip/hip_ext.h>

#include "pcr_internal.h"
#include "peer_exchange.h"
#include "small_math.h"

namespace pcr {

// The phase-skipping switches of a development build (PCR_ABLATE, scripts/timeline_ablate.py) exist only in a build with
// -DPCR_ABLATION: in the candidate loop their tests alone were two scalar branches per slot.
#ifdef PCR_ABLATION
static constexpr bool kAblation = true;
#else
static constexpr bool kAblation = false;
#endif



// ------------------------------------------------------------------------------
// exact nearest neighbours on the grid
//
// The reference asks nanoflann for the 5 nearest map points of every scan point (LoamRegister.cpp:47-72).  Here a search
// keeps the kNb = 8 nearest: five feed the plane fit, the other three are what lets the temporal-coherence cache below
// survive a Gauss-Newton step (on a voxel-filtered map the 5th and 6th neighbour are often a few millimetres apart -- the
// 5th and the 9th are a lattice shell apart).
// ------------------------------------------------------------------------------
static constexpr int kNb = 8;
static constexpr uint32_t kNoPos = 0xffffffffu;
static constexpr uint32_t kKeyEmpty = 0xffffffffu;

struct Nb8 {                 // the neighbours of one query in (distance, original index) order
    double d[kNb];           // exact squared distance, nanoflann's accumulation order; +inf: empty slot
    uint32_t idx[kNb];       // original target index; 0xffffffff: empty
    float x[kNb], y[kNb], z[kNb];
};

// (d, idx) lexicographic order, branch-free: distance ties are broken on the original index
__device__ __forceinline__ bool knn_less(double d, uint32_t i, double d2, uint32_t i2) {
    return (d < d2) | ((d == d2) & (i < i2));
}

// nanoflann L2_Simple_Adaptor::evalMetric order (nanoflann.hpp:523-535), double distances on float coordinates
__device__ __forceinline__ double sqdist(double qx, double qy, double qz, float px, float py, float pz) {
    const double dx = qx - (double)px, dy = qy - (double)py, dz = qz - (double)pz;
    double d = dx * dx;
    d += dy * dy;
    d += dz * dz;
    return d;
}

// 19-comparator sorting network on (distance, index) with the coordinates as payload
__device__ __forceinline__ void nb8_sort(Nb8& s) {
#define CSWAP(i, j) { const bool sw = knn_less(s.d[j], s.idx[j], s.d[i], s.idx[i]); \
        const double di = s.d[i], dj = s.d[j]; s.d[i] = sw ? dj : di; s.d[j] = sw ? di : dj; \
        const uint32_t ii = s.idx[i], ij = s.idx[j]; s.idx[i] = sw ? ij : ii; s.idx[j] = sw ? ii : ij; \
        const float xi = s.x[i], xj = s.x[j]; s.x[i] = sw ? xj : xi; s.x[j] = sw ? xi : xj; \
        const float yi = s.y[i], yj = s.y[j]; s.y[i] = sw ? yj : yi; s.y[j] = sw ? yi : yj; \
        const float zi = s.z[i], zj = s.z[j]; s.z[i] = sw ? zj : zi; s.z[j] = sw ? zi : zj; }
    CSWAP(0, 1) CSWAP(2, 3) CSWAP(4, 5) CSWAP(6, 7) CSWAP(0, 2) CSWAP(1, 3) CSWAP(4, 6) CSWAP(5, 7) CSWAP(1, 2) CSWAP(5, 6)
    CSWAP(0, 4) CSWAP(3, 7) CSWAP(1, 5) CSWAP(2, 6) CSWAP(1, 4) CSWAP(3, 6) CSWAP(2, 4) CSWAP(3, 5) CSWAP(3, 4)
#undef CSWAP
}

__device__ __forceinline__ void nb8_set(Nb8& s, int j, const float4 p, bool real, double qx, double qy, double qz) {
    const float inf = __uint_as_float(0x7f800000u);
    s.x[j] = real ? p.x : inf; s.y[j] = real ? p.y : inf; s.z[j] = real ? p.z : inf;
    s.idx[j] = real ? __float_as_uint(p.w) : 0xffffffffu;
    const double d = sqdist(qx, qy, qz, p.x, p.y, p.z);
    s.d[j] = (real && d == d) ? d : __longlong_as_double(0x7ff0000000000000ll);
}

// Per-block LDS scratch of the search: the non-empty row runs {start, end} of every lane's 3x3x3 block, compacted;
// entry 9 (and every entry past a lane's last run) is {0, 0}.
struct KnnRuns {
    uint2 run[10][256];
};

// Rows of the 3x3 (y,z) neighbourhood, centre first.  4-bit code per row: (dy+1) | (dz+1) << 2.
static constexpr unsigned long long kRowOrder = 0xA82091645ull;
__device__ __forceinline__ constexpr int row_dy(int r) { return (int)((kRowOrder >> (4 * r)) & 3) - 1; }
__device__ __forceinline__ constexpr int row_dz(int r) { return (int)((kRowOrder >> (4 * r + 2)) & 3) - 1; }

// Rare path, out of line: plain sequential scan of the 3x3x3 block with the full (distance, original index) order, for the
// queries whose screened list could not be proven (ties at the edge of the list, or more candidates than the key format
// holds).  Returns the kNb nearest and the exact squared distance of the next one (max_sq if there is none inside the gate).
// (The grid geometry is passed by value: handing over a reference to the kernel's GridHeader would force that copy
// into scratch memory in EVERY launch.)
__device__ __noinline__ void knn8_exact(double org_x, double org_y, double org_z, double inv_cell, uint32_t dim0, uint32_t dim1,
                                        const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start,
                                        double qx, double qy, double qz, double max_sq, uint32_t pos_out[kNb], double* next_sq) {
    double d9[kNb + 1]; uint32_t i9[kNb + 1], p9[kNb + 1];
    for (int j = 0; j <= kNb; ++j) { d9[j] = max_sq; i9[j] = 0xffffffffu; p9[j] = kNoPos; }
    const int cx = (int)floor((qx - org_x) * inv_cell), cy = (int)floor((qy - org_y) * inv_cell), cz = (int)floor((qz - org_z) * inv_cell);
    for (int r = 0; r < 9; ++r) {
        const uint32_t key = ((uint32_t)(cz + r / 3 - 1) * dim1 + (uint32_t)(cy + r % 3 - 1)) * dim0 + (uint32_t)cx;
        for (uint32_t j = cell_start[key - 1]; j < cell_start[key + 2]; ++j) {
            const float4 p = pts[j];
            const double d = sqdist(qx, qy, qz, p.x, p.y, p.z);
            const uint32_t idx = __float_as_uint(p.w);
            if (!knn_less(d, idx, d9[kNb], i9[kNb])) continue;
            int k = kNb;
            while (k > 0 && knn_less(d, idx, d9[k - 1], i9[k - 1])) { d9[k] = d9[k - 1]; i9[k] = i9[k - 1]; p9[k] = p9[k - 1]; --k; }
            d9[k] = d; i9[k] = idx; p9[k] = j;
        }
    }
    for (int j = 0; j < kNb; ++j) pos_out[j] = p9[j];
    *next_sq = d9[kNb];
}

// ------------------------------------------------------------------------------
// Per-lane search: one lane streams the candidates of its query's 3x3x3 block (9 contiguous x-runs) ONCE, as a flat
// sequence, and keeps the kNb smallest 32-bit keys
//     key = (float distance bits, low bits cleared) | sequence number of the candidate in that stream
// Two instructions per list slot (v_min_u32 / v_max_u32) instead of the five a (double distance, position) pair costs,
// float arithmetic for the distance, and nothing else per candidate but the cursor of the stream.  The float distance is a
// SCREEN: d~ differs from the exact double distance by < 1e-6 relative, the cleared bits by 2^-(23 - id bits).  What is
// exact: the smallest key that falls off the list is kept too; a candidate that did not make the list has a key >= that one,
// hence an exact distance
//     >= LB = float(that key, low bits cleared) * (1 - 1e-6).
// The owner recomputes the kNb listed candidates in double, sorts them by (distance, index) and accepts the five nearest
// iff the fifth is strictly nearer than LB (nothing outside the list can tie with it or beat it); otherwise, a handful of
// queries per million, the out-of-line exact scan decides.  LB also bounds every target point that is not in the list,
// which is what the temporal cache needs.
//
// part / parts_log2: this lane takes the part-th of 2^parts_log2 equal slices of the stream; the caller merges the lists.
// Returns 0: query outside the grid (no candidate can pass the gate), 1: keys valid, 2: more candidates than the key format
// numbers (the caller falls back to the exact scan).
// ------------------------------------------------------------------------------
struct KnnCursor { uint32_t j, e, k; uint2 nx; };

__device__ __forceinline__ uint32_t knn_cursor_step(KnnCursor& c, const KnnRuns& sh, int tid) {
    const uint32_t pos = c.j;
    const uint32_t j1 = c.j + 1u;
    const bool adv = j1 >= c.e;
    c.k += adv ? 1u : 0u;
    c.j = adv ? c.nx.x : j1;
    c.e = adv ? c.nx.y : c.e;
    c.nx = sh.run[min(c.k + 1u, 9u)][tid];
    return pos;
}

// The 18 range look-ups of a query's 3x3x3 block.  A lane that will search its OWN query requests them as soon as it knows that
// its cached neighbours do not hold (the round trip then overlaps the posting of the misses).
struct KnnRanges { uint32_t ra[9], rb[9]; bool inside; };

__device__ __forceinline__ void knn_ranges_issue(const GridHeader& h, const uint32_t* __restrict__ cell_start, float qxf, float qyf, float qzf, bool active,
                                                 KnnRanges& g) {
    const double qx = (double)qxf, qy = (double)qyf, qz = (double)qzf;
    // cell coordinates (exact: q is a float widened to double, origin a multiple of cell)
    const double rx = qx - h.origin[0], ry = qy - h.origin[1], rz = qz - h.origin[2];
    const double fx = floor(rx * h.inv_cell), fy = floor(ry * h.inv_cell), fz = floor(rz * h.inv_cell);
    // queries in the outermost cell (or beyond, or NaN) are >= one cell away from every point
    g.inside = active && (fx >= 1.0 && fx <= (double)(h.dims[0] - 2) && fy >= 1.0 && fy <= (double)(h.dims[1] - 2) &&
                          fz >= 1.0 && fz <= (double)(h.dims[2] - 2));
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
#pragma unroll
    for (int r = 0; r < 9; ++r) {   // 18 independent loads in flight
        uint32_t ck = ((uint32_t)(cz + row_dz(r)) * (uint32_t)h.dims[1] + (uint32_t)(cy + row_dy(r))) * (uint32_t)h.dims[0] + (uint32_t)cx;
        ck = g.inside ? ck : 1u;      // the others read the first entries of the table (unconditional loads; never outside it)
        g.ra[r] = cell_start[ck - 1]; g.rb[r] = cell_start[ck + 2];
    }
}

template <int kGroup>
__device__ __forceinline__ int knn_scan(const GridHeader& h, const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start,
                                        float qxf, float qyf, float qzf, float gate_f, bool active, KnnRuns& sh, uint32_t part, uint32_t parts_log2,
                                        uint32_t key[kNb + 1], uint32_t* idmask_out, unsigned long long* tl, const KnnRanges* pre = nullptr) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j <= kNb; ++j) key[j] = kKeyEmpty;      // (key[kNb]: the smallest key that is NOT in the list -- the bound of everything else)
    *idmask_out = 0u;
    KnnRanges g;
    if (pre) g = *pre;
    else knn_ranges_issue(h, cell_start, qxf, qyf, qzf, active && !h.empty && !h.overflow, g);
    const bool inside = g.inside && active;
    uint32_t n_runs = 0, total = 0;
    if (inside) {
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (g.rb[r] > g.ra[r]) { sh.run[n_runs][tid] = make_uint2(g.ra[r], g.rb[r]); ++n_runs; total += g.rb[r] - g.ra[r]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) if ((uint32_t)r >= n_runs) sh.run[r][tid] = make_uint2(0u, 0u);
    if (tl) tl[8] = wall_clock64();       // row ranges in LDS
    // (each lane reads back only what it wrote itself: no barrier needed)
    const uint32_t bits = 32u - (uint32_t)__clz((int)(total | 1u));
    const uint32_t idmask = (1u << bits) - 1u;
    *idmask_out = idmask;
    const bool too_many = bits > 16u;
    const uint32_t s0 = (uint32_t)(((uint64_t)total * part) >> parts_log2), s1 = (uint32_t)(((uint64_t)total * (part + 1u)) >> parts_log2);
    const uint32_t n_mine = too_many ? 0u : s1 - s0;
    // position the cursor at flat index s0
    KnnCursor cur;
    {
        uint32_t k = 0, rem = s0;
        if (parts_log2) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint2 r = sh.run[t][tid];
                const uint32_t len = r.y - r.x;
                const bool skip = (uint32_t)t == k && (uint32_t)t < n_runs && rem >= len;
                rem -= skip ? len : 0u; k += skip ? 1u : 0u;
            }
        }
        const uint2 r0 = sh.run[min(k, 9u)][tid];
        cur.k = k; cur.j = r0.x + rem; cur.e = r0.y; cur.nx = sh.run[min(k + 1u, 9u)][tid];
    }
    // wave-uniform number of groups
    uint32_t nmax = n_mine;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, m, 64));
    const uint32_t groups = __builtin_amdgcn_readfirstlane((nmax + (uint32_t)kGroup - 1u) / (uint32_t)kGroup);
    float4 c[kGroup];
    if (groups) {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
            const uint32_t pos = knn_cursor_step(cur, sh, tid);
            c[u] = pts[(uint32_t)u < n_mine ? pos : 0u];
        }
    }
    if (tl) { float t_ = 0.f; if (groups) { for (int u = 0; u < kGroup; ++u) t_ += c[u].x; } if (t_ == 1.2345e38f) key[0] = 0u; tl[9] = wall_clock64(); }   // first group arrived
    for (uint32_t g = 0; g < groups; ++g) {
        float4 n[kGroup];
        if (g + 1u < groups) {
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
                const uint32_t pos = knn_cursor_step(cur, sh, tid);
                n[u] = pts[(g + 1u) * (uint32_t)kGroup + (uint32_t)u < n_mine ? pos : 0u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < kGroup; ++u) n[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
            const uint32_t seq = g * (uint32_t)kGroup + (uint32_t)u;
            const float dx = qxf - c[u].x, dy = qyf - c[u].y, dz = qzf - c[u].z;
            float d = dx * dx;
            d = __builtin_fmaf(dy, dy, d);
            d = __builtin_fmaf(dz, dz, d);
            uint32_t t = (__float_as_uint(d) & ~idmask) | (s0 + seq);
            t = (seq < n_mine && d <= gate_f) ? t : kKeyEmpty;      // beyond the end of this lane's stream, beyond the gate, or NaN
#pragma unroll
            for (int k = 0; k < kNb; ++k) { const uint32_t lo = min(key[k], t), hi = max(key[k], t); key[k] = lo; t = hi; }
            key[kNb] = min(key[kNb], t);          // what falls off the list: its minimum is the (kNb + 1)-th smallest key
        }
#pragma unroll
        for (int u = 0; u < kGroup; ++u) c[u] = n[u];
    }
    if (tl) tl[10] = wall_clock64();      // candidate stream done
    return !inside ? 0 : (too_many ? 2 : 1);
}

// sequence number of a candidate in the lane's stream -> position in the cell-sorted array (the lane's run table is still in LDS)
__device__ __forceinline__ void knn_decode(const KnnRuns& sh, int tid, const uint32_t key[kNb + 1], uint32_t idmask, uint32_t pos[kNb]) {
    // run t holds the sequence numbers [P_t, P_t + len_t): position = sequence number + (start_t - P_t) of the last run with P_t <= it
    uint32_t P[9], off[9];
    uint32_t acc = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) { const uint2 r = sh.run[t][tid]; P[t] = acc; off[t] = r.x - acc; acc += r.y - r.x; }
#pragma unroll
    for (int j = 0; j < kNb; ++j) {
        const uint32_t seq = key[j] & idmask;
        uint32_t o = off[0];
#pragma unroll
        for (int t = 1; t < 9; ++t) o = seq >= P[t] ? off[t] : o;      // (the empty runs past the last real one have P_t = total > seq)
        pos[j] = key[j] == kKeyEmpty ? kNoPos : seq + o;
    }
}

// squared-distance bound of everything that is not in a screened list (see knn_scan)
__device__ __forceinline__ double knn_list_bound(const uint32_t key[kNb + 1], uint32_t idmask) {
    if (key[kNb] == kKeyEmpty) return __longlong_as_double(0x7ff0000000000000ll);   // every candidate inside the gate is in the list
    return (double)__uint_as_float(key[kNb] & ~idmask) * (1.0 - 1e-6);
}

// ------------------------------------------------------------------------------
// Temporal coherence between Gauss-Newton iterations (exact, not approximate).
// After a search we keep, per scan point: the query position q0, its kNb nearest map points (coordinates and original
// indices) and L, a lower bound of the distance from q0 to every OTHER target point.  At a later iteration the query has
// moved by delta = |q - q0|.  For any other point p: |p - q| >= |p - q0| - delta >= L - delta, so if the fifth nearest of
// the kept points is strictly closer to q than that, the five nearest of the kept points ARE the exact 5-NN of q.
// Otherwise the search runs again.  The plane through the five (which does not depend on the query) is kept with the
// indices it was fitted to and reused while the five and their order stay the same.
// ------------------------------------------------------------------------------
struct NnCacheEntry {          // 192 bytes per scan point = 12 x 16 (fetched by LDS-DMA)
    float4 nb[kNb];            // x y z | original index bits; +inf coordinates and index 0xffffffff: empty slot
    float q0[3];               // query position of the SEARCH that produced the entry (never moved afterwards: a bound
                               // anchored there is at least as tight as one chained through the iterations)
    float l9;                  // lower bound (rounded down) of the DISTANCE from q0 to every target point not in nb
    double x[3];               // plane A x = -1 through the points pidx, in that order (LoamRegister.cpp:29-35)
    uint32_t pidx[5];
    uint32_t flags;            // bit0: entry valid  bit1: x valid  bit2: plane passed its validity gate
};
static_assert(sizeof(NnCacheEntry) == 192, "cache entry layout");
static constexpr int kEntryVec = sizeof(NnCacheEntry) / 16;

// Distances from the new query to the kept points, put in (distance, index) order.  Returns true when the first five are
// proven to be the exact 5-NN (see above).  *moved: the order of the kept points changed (the entry is rewritten so that the
// next iteration finds it sorted).
__device__ __forceinline__ bool knn_from_cache(const NnCacheEntry& ce, double qx, double qy, double qz, Nb8& s, bool* moved) {
#pragma unroll
    for (int j = 0; j < kNb; ++j) {
        const float4 p = ce.nb[j];
        s.x[j] = p.x; s.y[j] = p.y; s.z[j] = p.z; s.idx[j] = __float_as_uint(p.w);
        const double d = sqdist(qx, qy, qz, p.x, p.y, p.z);
        s.d[j] = d == d ? d : __longlong_as_double(0x7ff0000000000000ll);      // (an empty slot holds +inf coordinates)
    }
    // the usual case late in the loop: the five are still the five, in order -- then the network is skipped by the whole wave
    bool ord = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) ord = ord & !knn_less(s.d[j + 1], s.idx[j + 1], s.d[j], s.idx[j]);
#pragma unroll
    for (int j = 5; j < kNb; ++j) ord = ord & !knn_less(s.d[j], s.idx[j], s.d[4], s.idx[4]);
    *moved = false;
    if (!__all(ord)) {
        uint32_t before[kNb];
#pragma unroll
        for (int j = 0; j < kNb; ++j) before[j] = s.idx[j];
        nb8_sort(s);
        bool mv = false;
#pragma unroll
        for (int j = 0; j < kNb; ++j) mv = mv | (before[j] != s.idx[j]);
        *moved = mv;
    }
    // |p - q| >= |p - q0| - |q - q0| >= l9 - delta for every other target point p.  Double precision: the stored bound
    // loses at most one float ulp.
    const double ex = qx - (double)ce.q0[0], ey = qy - (double)ce.q0[1], ez = qz - (double)ce.q0[2];
    const double delta = sqrt(ex * ex + ey * ey + ez * ez) * (1.0 + 1e-12) + 1e-13;
    const double slack = (double)ce.l9 - delta;
    return s.idx[4] != 0xffffffffu && slack > 0.0 && s.d[4] < slack * slack * (1.0 - 1e-12);
}

// ------------------------------------------------------------------------------
// plane fit: Eigen::ColPivHouseholderQR<Matrix<double,5,3>>::solve(-1) restated
// (LoamRegister.cpp:29-35), all indices static so everything stays in registers.
// ------------------------------------------------------------------------------
__device__ __forceinline__ void swap_d(double& a, double& b) { const double t = a; a = b; b = t; }

__device__ __forceinline__ void plane_qr_solve(double a[5][3], double x[3]) {
    double c[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    int perm[3] = {0, 1, 2};
    double tau[3] = {0, 0, 0};
    double nu[3], nd[3];
    double maxn = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) s += a[i][j] * a[i][j];
        nd[j] = nu[j] = sqrt(s);
        if (nu[j] > maxn) maxn = nu[j];
    }
    const double eps = 2.220446049250313e-16;
    const double thr_helper = (maxn * eps) * (maxn * eps) / 5.0;
    const double downdate_thr = 1.4901161193847656e-08;  // sqrt(eps)
    int nonzero = 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int big = k; double bign = nu[k];
#pragma unroll
        for (int j = k + 1; j < 3; ++j) if (nu[j] > bign) { bign = nu[j]; big = j; }
        const double big_sq = bign * bign;
        if (nonzero == 3 && big_sq < thr_helper * (double)(5 - k)) nonzero = k;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (big == j) {
#pragma unroll
                for (int i = 0; i < 5; ++i) swap_d(a[i][k], a[i][j]);
                swap_d(nu[k], nu[j]); swap_d(nd[k], nd[j]);
                const int t = perm[k]; perm[k] = perm[j]; perm[j] = t;
            }
        }
        double tail_sq = 0;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) tail_sq += a[i][k] * a[i][k];
        const double c0 = a[k][k];
        double beta;
        if (tail_sq <= 2.2250738585072014e-308) {
            tau[k] = 0; beta = c0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][k] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0) beta = -beta;
            const double den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][k] /= den;
            tau[k] = (beta - c0) / beta;
        }
        a[k][k] = beta;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            double tmp = 0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) tmp += a[i][k] * a[i][j];
            tmp += a[k][j];
            a[k][j] -= tau[k] * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][j] -= tau[k] * a[i][k] * tmp;
        }
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (nu[j] != 0) {
                double temp = fabs(a[k][j]) / nu[j];
                temp = (1.0 + temp) * (1.0 - temp);
                if (temp < 0) temp = 0;
                const double r = nu[j] / nd[j];
                const double temp2 = temp * r * r;
                if (temp2 <= downdate_thr) {
                    double s = 0;
#pragma unroll
                    for (int i = k + 1; i < 5; ++i) s += a[i][j] * a[i][j];
                    nd[j] = sqrt(s); nu[j] = nd[j];
                } else {
                    nu[j] *= sqrt(temp);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < nonzero) {
            double tmp = 0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) tmp += a[i][k] * c[i];
            tmp += c[k];
            c[k] -= tau[k] * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) c[i] -= tau[k] * a[i][k] * tmp;
        }
    }
    double y[3] = {0, 0, 0};
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        if (i < nonzero) {
            double s = c[i];
#pragma unroll
            for (int j = i + 1; j < 3; ++j) if (j < nonzero) s -= a[i][j] * y[j];
            y[i] = s / a[i][i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double v = (i < nonzero) ? y[i] : 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (perm[i] == j) x[j] = v;
    }
}


// ------------------------------------------------------------------------------
// Wave-cooperative exact search of ONE query: lanes = candidates.  Used when only a handful of queries of a
// block still need a search (late Gauss-Newton iterations): the per-lane search above is a chain of dependent memory
// round trips whether 1 or 64 lanes are busy, this one is three (ranges, points, done).
// Every candidate of the 3x3x3 block gets its exact (distance, index) key; a candidate's rank is the number
// of smaller keys, counted against the wave's table in LDS -- no ordering between lanes is needed.
// All 64 lanes call it with the same query.  tab: >= kWaveTab + 16 entries of this wave.  Returns false when the
// block holds more candidates than the table (the caller then falls back to the per-lane search).
// ------------------------------------------------------------------------------
struct WaveCand { double d; uint32_t idx, pos; };
static constexpr int kWaveTab = 280;     // 4 waves x (280 + 16 result slots) x 16 B = 18.5 KB of the 20 KB of KnnRuns, which this path reuses

__device__ __forceinline__ bool knn8_wave(const GridHeader& h, const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start,
                                          double qx, double qy, double qz, double max_sq, double seed_sq, WaveCand* tab, uint32_t pos_out[kNb],
                                          double* next_sq, bool* inside_out) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < kNb; ++j) pos_out[j] = kNoPos;
    *next_sq = max_sq;
    const double rx = qx - h.origin[0], ry = qy - h.origin[1], rz = qz - h.origin[2];
    const double fx = floor(rx * h.inv_cell), fy = floor(ry * h.inv_cell), fz = floor(rz * h.inv_cell);
    const bool inside = (fx >= 1.0 && fx <= (double)(h.dims[0] - 2) && fy >= 1.0 && fy <= (double)(h.dims[1] - 2) && fz >= 1.0 &&
                         fz <= (double)(h.dims[2] - 2));
    *inside_out = inside;
    if (!inside) return true;
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    // lanes 0..8 fetch the 9 row ranges
    uint32_t rs = 0, len = 0;
    if (lane < 9) {
        const uint32_t key = ((uint32_t)(cz + lane / 3 - 1) * (uint32_t)h.dims[1] + (uint32_t)(cy + lane % 3 - 1)) * (uint32_t)h.dims[0] + (uint32_t)cx;
        rs = cell_start[key - 1];
        len = cell_start[key + 2] - rs;
    }
    uint32_t incl = len;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const uint32_t total = __shfl(incl, 8, 64);
    if (total > (uint32_t)kWaveTab) return false;
    uint32_t r_start[9], r_excl[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) { r_start[r] = __shfl(rs, r, 64); r_excl[r] = __shfl(incl - len, r, 64); }
    // Every lane computes the exact keys of its candidates.  Only those within seed_sq -- an upper bound of the kNb-th
    // distance (kNb known points lie inside it; max_sq when nothing is known) -- can be among the kNb nearest: they are
    // compacted into the table and ranked among themselves (a handful instead of ~100); the others only bound the
    // next neighbour from below through their minimum.
    uint32_t n_in = 0;                 // wave-uniform
    double out_min = max_sq;
    for (uint32_t c0 = 0; c0 < total; c0 += 64) {
        const uint32_t c = c0 + (uint32_t)lane;
        bool in = false;
        WaveCand w; w.d = max_sq; w.idx = 0xffffffffu; w.pos = kNoPos;
        if (c < total) {
            uint32_t pos = 0;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (c >= r_excl[r]) pos = r_start[r] + (c - r_excl[r]);
            const float4 p = pts[pos];
            const double d = sqdist(qx, qy, qz, p.x, p.y, p.z);
            w.d = d; w.idx = __float_as_uint(p.w); w.pos = pos;
            in = d <= seed_sq;
            if (!in) out_min = fmin(out_min, d);
        }
        const unsigned long long m = __ballot(in);
        if (in) tab[n_in + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = w;
        n_in += (uint32_t)__popcll(m);
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) out_min = fmin(out_min, __shfl_xor(out_min, sft, 64));
    // (same wave: its LDS operations complete in order, so the table is visible to the reads below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    for (uint32_t c = lane; c < n_in; c += 64) {
        const WaveCand me = tab[c];
        uint32_t rank = 0;
#pragma unroll 4
        for (uint32_t j = 0; j < n_in; ++j) {
            const WaveCand o = tab[j];
            rank += knn_less(o.d, o.idx, me.d, me.idx) ? 1u : 0u;
        }
        if (rank <= (uint32_t)kNb) { tab[kWaveTab + rank] = me; }       // kNb + 1 result slots behind the table
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
    for (int j = 0; j < kNb; ++j) {
        if ((uint32_t)j < n_in) pos_out[j] = tab[kWaveTab + j].pos;
    }
    double nx = out_min;
    if (n_in > (uint32_t)kNb) nx = fmin(nx, tab[kWaveTab + kNb].d);
    *next_sq = fmin(max_sq, nx);
    return true;
}

// Block-level compaction of the queries that need a full search.  After the first iterations most lanes
// are served by the neighbour cache; the search body is straight-line code that costs a wave the same
// whether 1 or 64 of its lanes are active, so the remaining queries are gathered into dense waves:
// owners post their query, threads 0..count-1 search one posted query each and post the result back.
static constexpr int kRowStride = 258;   // doubles per component row in LDS (256 + pad: conflict-free reads)
static constexpr int kSparseMisses = 8;  // up to this many posted queries per block go the wave-cooperative way
struct MissExchange {
    uint32_t count, fallback;
    uint32_t list[256];                  // owner thread of the m-th posted query (order immaterial)
    float qx[256], qy[256], qz[256];     // query (float-valued)
    double seed[256];                    // upper bound of the kNb-th squared distance (max_sq when nothing is known)
    union {
        struct {
            uint32_t pos[kNb][256];      // by owner: positions of the listed candidates in the cell-sorted array
            double bound[256];           // by owner: squared-distance bound of every target point not listed
            uint32_t state[256];         // by owner: 0 no candidates / outside, 1 screened list (to be proven), 2 exact list, 3 redo exactly
            uint32_t keys[kNb + 1][256]; // by worker thread: partial lists of the lanes that share a query (+ the first key off each list)
            uint32_t idmask[256];
        } res;
        double rows[8 * kRowStride];     // later: [component][point] rows of the normal equations
    } u;
};

// ------------------------------------------------------------------------------
// one scan point: returns status (0 accepted, 1 k-NN gate, 2 plane gate, 3 weight gate,
// 4 outside this rank's query tile / no point);  row[0..5] = s*[n ; p x n], row[6] = s*d
// Called by all threads of the block together (it contains barriers).
// ------------------------------------------------------------------------------
template <int kGroup>
__device__ __forceinline__ int loam_point(const LoamArgs& a, const GridHeader& h, const double* __restrict__ pose,
                                          float sx, float sy, float sz, bool valid, const NnCacheEntry& ce_in, bool have_entry,
                                          KnnRuns& sh, MissExchange& ex, double row[7], uint32_t nn_idx[5], uint32_t qi, int* how,
                                          bool* escaped, unsigned long long* tl, const bool all_search = false) {
    const double ox = (double)sx, oy = (double)sy, oz = (double)sz;
    // LoamRegister.cpp:126-130: Isometry3d * Vector4d in f64, then cast to f32
    const float px = (float)(pose[0] * ox + pose[4] * oy + pose[8] * oz + pose[12] * 1.0);
    const float py = (float)(pose[1] * ox + pose[5] * oy + pose[9] * oz + pose[13] * 1.0);
    const float pz = (float)(pose[2] * ox + pose[6] * oy + pose[10] * oz + pose[14] * 1.0);
    const double qx = (double)px, qy = (double)py, qz = (double)pz;
    const bool in_range = valid;
    bool active = valid && !h.empty && !h.overflow;
    if (a.use_tile) {
        const bool in_tile = qx >= a.tile_lo[0] && qx < a.tile_hi[0] && qy >= a.tile_lo[1] && qy < a.tile_hi[1] &&
                             qz >= a.tile_lo[2] && qz < a.tile_hi[2];
        if (!in_tile) { active = false; valid = false; }
    }
    *escaped = false;
    if (h.clamped && active) {
        // The index covers only part of the target (capi.hip, ClampBox).  A query whose 3x3x3 block touches cells beyond a
        // cut face could miss neighbours there: it is counted, and the host redoes the call on a wider region.
        const double fx = floor((qx - h.origin[0]) * h.inv_cell), fy = floor((qy - h.origin[1]) * h.inv_cell), fz = floor((qz - h.origin[2]) * h.inv_cell);
        const int cm = h.cut_mask;
        *escaped = ((cm & 1) && fx <= (double)kPad) || ((cm & 8) && fx >= (double)(h.dims[0] - kPad - 1)) ||
                   ((cm & 2) && fy <= (double)kPad) || ((cm & 16) && fy >= (double)(h.dims[1] - kPad - 1)) ||
                   ((cm & 4) && fz <= (double)kPad) || ((cm & 32) && fz >= (double)(h.dims[2] - kPad - 1));
    }
    Nb8 s;
    bool moved = false;
    const bool have_seed = have_entry && active && (ce_in.flags & 1u);
    bool hit = false;
    if (have_seed) hit = knn_from_cache(ce_in, qx, qy, qz, s, &moved);
    const int tid = threadIdx.x;
    // ---- post the queries that need a search ----
    const bool miss = active && !hit;
    // (its row ranges are requested now, for the case that this lane ends up searching its own query: the round trip overlaps the
    //  posting and the barriers)
    KnnRanges own_ranges;
    own_ranges.inside = false;
#pragma unroll
    for (int r = 0; r < 9; ++r) { own_ranges.ra[r] = 0u; own_ranges.rb[r] = 0u; }
    if (__any(miss)) knn_ranges_issue(h, a.grid.cell_start, px, py, pz, miss, own_ranges);      // (wave-uniform: nothing is loaded in an all-hit wave)
    if (!all_search) {      // (block-uniform)
    if (tid == 0) ex.count = 0;
    __syncthreads();
    if (miss) {
        const uint32_t m = atomicAdd(&ex.count, 1u);
        ex.list[m] = (uint32_t)tid;
        ex.qx[tid] = px; ex.qy[tid] = py; ex.qz[tid] = pz;
        // kept points that could not be proven final still bound the kNb-th distance from above (when all kNb are real)
        double sb = a.c.knn_max_sq;
        if (have_seed) {
            double mx = s.d[0];        // (+inf as soon as one slot is empty)
#pragma unroll
            for (int j = 1; j < kNb; ++j) mx = fmax(mx, s.d[j]);
            sb = fmin(sb, mx);
        }
        ex.seed[tid] = sb;
    }
    __syncthreads();
    }
    if (tl) tl[2] = wall_clock64();
    // ---- the posted queries are searched: a few -> one wave per query (lanes = candidates); many -> thread m
    //      serves the m-th posted query (lanes = queries) ----
    bool self = false;
    uint32_t self_pos[kNb], self_state = 0u;
    double self_bound = 0.0;
    {
        const uint32_t n_miss = all_search ? 256u : ex.count;           // block-uniform
        bool dense = n_miss > (uint32_t)kSparseMisses;
        if (n_miss && !dense) {
            if (tid == 0) ex.fallback = 0;
            __syncthreads();
            const int wave = tid >> 6;
            WaveCand* tab = reinterpret_cast<WaveCand*>(&sh) + (size_t)wave * (kWaveTab + 16);
            for (uint32_t m = wave; m < n_miss; m += 4) {
                const uint32_t owner = ex.list[m];
                uint32_t rp[kNb];
                double nxt = 0.0;
                bool ins = false;
                const bool done = knn8_wave(h, a.grid.pts, a.grid.cell_start, (double)ex.qx[owner], (double)ex.qy[owner], (double)ex.qz[owner],
                                            a.c.knn_max_sq, ex.seed[owner], tab, rp, &nxt, &ins);
                if (!done) { if ((tid & 63) == 0) ex.fallback = 1; }
                else if ((tid & 63) == 0) {
#pragma unroll
                    for (int j = 0; j < kNb; ++j) ex.u.res.pos[j][owner] = rp[j];
                    ex.u.res.bound[owner] = nxt; ex.u.res.state[owner] = ins ? 2u : 0u;
                }
            }
            __syncthreads();
            dense = ex.fallback != 0;               // a cell block too crowded for the wave table: redo all of them per lane
            __syncthreads();
        }
        if (n_miss > 128u && dense) {
            // more than half the block: every lane searches its OWN query (a compaction would not free a single wave), with the
            // ranges it asked for before the barrier and without the trip through the exchange
            self = true;
            uint32_t key[kNb + 1], idmask = 0u;
            const float gate_f = __double2float_ru(a.c.knn_max_sq * (1.0 + 1e-5));      // every candidate inside the gate has a float distance <= this
            const int rc = knn_scan<kGroup>(h, a.grid.pts, a.grid.cell_start, px, py, pz, gate_f, miss, sh, 0u, 0u, key, &idmask, tl, &own_ranges);
            knn_decode(sh, tid, key, idmask, self_pos);
            self_bound = knn_list_bound(key, idmask);
            self_state = rc == 0 ? 0u : (rc == 2 ? 3u : 1u);
        } else if (n_miss && dense) {
            // A block with few misses would leave three of its four waves idle while one wave walks whole streams: up to 64
            // (128) posted queries are searched by four (two) lanes each, in different waves, every lane taking its slice of
            // the query's candidate stream; the lists are merged through LDS.
            const uint32_t pl2 = n_miss <= 64u ? 2u : 1u;
            const uint32_t per = 256u >> pl2;                          // lanes per part
            const uint32_t m = (uint32_t)tid & (per - 1u), part = (uint32_t)tid >> (8u - pl2);
            const bool worker = m < n_miss;
            const uint32_t owner = worker ? ex.list[m] : 0u;
            uint32_t key[kNb + 1], idmask = 0u;
            const float gate_f = __double2float_ru(a.c.knn_max_sq * (1.0 + 1e-5));      // every candidate inside the gate has a float distance <= this
            const int rc = knn_scan<kGroup>(h, a.grid.pts, a.grid.cell_start, ex.qx[owner], ex.qy[owner], ex.qz[owner], gate_f, worker, sh, part, pl2, key,
                                            &idmask, tl);
            if (pl2) {
                if (worker && part) {
#pragma unroll
                    for (int j = 0; j <= kNb; ++j) ex.u.res.keys[j][tid] = key[j];
                }
                __syncthreads();
                if (worker && !part) {
                    for (uint32_t q = 1; q < (1u << pl2); ++q) {
                        const uint32_t t2 = m + q * per;
#pragma unroll
                        for (int j = 0; j <= kNb; ++j) {
                            uint32_t t = ex.u.res.keys[j][t2];
#pragma unroll
                            for (int k = 0; k < kNb; ++k) { const uint32_t lo = min(key[k], t), hi = max(key[k], t); key[k] = lo; t = hi; }
                            key[kNb] = min(key[kNb], t);
                        }
                    }
                }
                __syncthreads();          // every partial list has been read: the union may now take final results (indexed by owner)
            }
            if (worker && !part) {
                uint32_t rp[kNb];
                knn_decode(sh, tid, key, idmask, rp);
#pragma unroll
                for (int j = 0; j < kNb; ++j) ex.u.res.pos[j][owner] = rp[j];
                ex.u.res.bound[owner] = knn_list_bound(key, idmask);
                ex.u.res.state[owner] = rc == 0 ? 0u : (rc == 2 ? 3u : 1u);
            }
        }
    }
    __syncthreads();
    if (tl) tl[3] = wall_clock64();
    bool searched = hit;
    double bound_sq = 0.0;      // (miss) squared-distance bound of every target point not in s
    if (miss) {
        uint32_t rp[kNb];
#pragma unroll
        for (int j = 0; j < kNb; ++j) rp[j] = self ? self_pos[j] : ex.u.res.pos[j][tid];
        bound_sq = self ? self_bound : ex.u.res.bound[tid];
        uint32_t state = self ? self_state : ex.u.res.state[tid];
        searched = state != 0u;
        float4 p8[kNb];
#pragma unroll
        for (int j = 0; j < kNb; ++j) p8[j] = a.grid.pts[(searched && rp[j] != kNoPos) ? rp[j] : 0u];
#pragma unroll
        for (int j = 0; j < kNb; ++j) nb8_set(s, j, p8[j], searched && state != 3u && rp[j] != kNoPos, qx, qy, qz);
        nb8_sort(s);
        // a screened list is final when its fifth entry is strictly nearer than everything that is not listed
        const bool proven = state == 2u || (state == 1u && (s.idx[4] == 0xffffffffu ? !(bound_sq < a.c.knn_max_sq) : s.d[4] < bound_sq));
        if (searched && !proven) {
            double nxt;
            uint32_t rp2[kNb];      // (an array of its own: handing &rp to the out-of-line function made the compiler keep rp in scratch memory on EVERY path -- eight stores per point)
            knn8_exact(h.origin[0], h.origin[1], h.origin[2], h.inv_cell, (uint32_t)h.dims[0], (uint32_t)h.dims[1], a.grid.pts, a.grid.cell_start, qx, qy, qz,
                       a.c.knn_max_sq, rp2, &nxt);
            for (int j = 0; j < kNb; ++j) nb8_set(s, j, a.grid.pts[rp2[j] != kNoPos ? rp2[j] : 0u], rp2[j] != kNoPos, qx, qy, qz);
            nb8_sort(s);
            bound_sq = nxt;
        }
        bound_sq = fmin(bound_sq, a.c.knn_max_sq);      // points outside the 3x3x3 block are >= one cell (>= sqrt(max_sq)) away
    }
    __syncthreads();   // ex.u.rows is written next
    *how = hit ? 1 : (miss ? 2 : 0);
    const bool real5 = searched && s.idx[4] != 0xffffffffu;     // five real neighbours present

    // ---- plane through the 5 neighbours (LoamRegister.cpp:29-45): reused from the cache when the five and their order are
    // the ones it was fitted to, since it does not depend on the query ----
    double x[3] = {0, 0, 0};
    bool plane_ok = false;
    bool reuse = real5 && have_seed && (ce_in.flags & 2u);
#pragma unroll
    for (int j = 0; j < 5; ++j) reuse = reuse && (s.idx[j] == ce_in.pidx[j]);
    const bool gate_knn = real5 && (s.d[4] < a.c.knn_max_sq);      // LoamRegister.cpp:59
    double A[5][3];
#pragma unroll
    for (int j = 0; j < 5; ++j) { A[j][0] = (double)s.x[j]; A[j][1] = (double)s.y[j]; A[j][2] = (double)s.z[j]; }
    if (reuse) {
        x[0] = ce_in.x[0]; x[1] = ce_in.x[1]; x[2] = ce_in.x[2];
        plane_ok = (ce_in.flags & 4u) != 0;
    } else if (real5 && !(kAblation && (a.ablate & 2))) {
        double Aq[5][3];
#pragma unroll
        for (int j = 0; j < 5; ++j) { Aq[j][0] = A[j][0]; Aq[j][1] = A[j][1]; Aq[j][2] = A[j][2]; }
        plane_qr_solve(Aq, x);
        const double xn0 = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        plane_ok = true;           // LoamRegister.cpp:38-43
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double dot = x[0] * A[i][0] + x[1] * A[i][1] + x[2] * A[i][2];
            if (fabs(dot + 1.0) > a.c.plane_thresh * xn0) plane_ok = false;
        }
    }
    // ---- remember everything for the next iteration ----
    if (a.nn_cache && in_range && !active) {
        // not handled by this launch (outside this rank's query tile): whatever entry is there -- a previous iteration's, a
        // previous scan's -- must not be trusted when the query comes back
        a.nn_cache[qi].flags = 0;
    }
    if (a.nn_cache && active) {
        float4* const dst = reinterpret_cast<float4*>(a.nn_cache + qi);
        if (miss || moved) {
#pragma unroll
            for (int j = 0; j < kNb; ++j) dst[j] = make_float4(s.x[j], s.y[j], s.z[j], __uint_as_float(s.idx[j]));
        }
        if (miss) {
            // the anchor of the entry: this search's query position and what it proved about everything it did not list
            dst[kNb] = make_float4(px, py, pz, __double2float_rd(sqrt(bound_sq) * (1.0 - 1e-15)));
        }
        if (miss || !reuse) {
            union { struct { double x[3]; uint32_t pidx[5]; uint32_t flags; } t; float4 v[3]; } tail;
            tail.t.x[0] = x[0]; tail.t.x[1] = x[1]; tail.t.x[2] = x[2];
#pragma unroll
            for (int j = 0; j < 5; ++j) tail.t.pidx[j] = s.idx[j];
            const bool x_valid = real5 && !(kAblation && (a.ablate & 2));
            tail.t.flags = (searched ? 1u : 0u) | (x_valid ? 2u : 0u) | (plane_ok ? 4u : 0u);
#pragma unroll
            for (int f = 0; f < 3; ++f) dst[kNb + 1 + f] = tail.v[f];
        }
    }
    if (tl) tl[4] = wall_clock64();
    if (!valid) return 4;
    if (!searched) return 1;
#pragma unroll
    for (int j = 0; j < 5; ++j) nn_idx[j] = s.idx[j];
    if (!gate_knn) return 1;
    if (kAblation && (a.ablate & 2)) return 2;
    if (!plane_ok) return 2;
    const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    const double dist = (qx * x[0] + qy * x[1] + qz * x[2] + 1.0) / xn;   // LoamRegister.hpp:75-77
    const float r2 = sx * sx + sy * sy + sz * sz;                           // LoamRegister.cpp:147-148 (float)
    const float rr = sqrtf(sqrtf(r2));
    const double w = 1 - 0.9 * fabs(dist) / (double)rr;
    if (!(w > a.c.point_thresh)) return 3;
    const double n0 = x[0] / xn, n1 = x[1] / xn, n2 = x[2] / xn;            // LoamRegister.hpp:70-73
    const double s0 = w * n0, s1 = w * n1, s2 = w * n2;
    row[0] = s0; row[1] = s1; row[2] = s2;
    row[3] = s1 * (-qz) + s2 * qy;       // s*n^T * (-skew(p)), manifolds.hpp:63-68, matrix.hpp:13-18
    row[4] = s0 * qz + s2 * (-qx);
    row[5] = s0 * (-qy) + s1 * qx;
    row[6] = w * dist;
    return 0;
}

// ------------------------------------------------------------------------------
// prologue: fold the previous launch's partial sums, solve, update the pose.
// Every block runs it and obtains bit-identical results.  Returns true when this
// launch must not linearise (loop finished).
// ------------------------------------------------------------------------------
static constexpr double kBigStep = 0.12;      // metres at 10 m from the sensor: |rho| + 10 |omega| of the Gauss-Newton step
struct Prologue {
    double pose[16];
    int done;
    int big_step;      // the step just applied moves the scan by more than a neighbour cache entry survives: this launch searches every query afresh
};

// 16 bytes per lane from global memory straight into LDS (lane i lands at lds_wave_base + 16 i); completion is
// signalled on vmcnt like any load.
__device__ __forceinline__ void lds_dma16(const float4* src, float4* lds_wave_base) {
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(base) : "m0", "memory");
}

// pre_src / pre_dst: this lane's 192-byte neighbour-cache entry and the wave's slice of the LDS staging area.
__device__ bool loam_prologue(const LoamArgs& a, int k, double* sh_sum /* 8*32 */, Prologue* sh, unsigned long long* tl = nullptr,
                              const float4* pre_src = nullptr, float4* pre_dst = nullptr) {
    const LoamState* prev = &a.state[(k + 1) & 1];
    LoamState* cur = &a.state[k & 1];
    const int t = threadIdx.x;
    if (k == 0) {
        if (t < 16) sh->pose[t] = a.init_pose[t];
        if (t == 0) { sh->done = 0; sh->big_step = 0; }
        if (blockIdx.x == 0 && t == 0) {
            for (int i = 0; i < 16; ++i) cur->pose[i] = a.init_pose[i];
            cur->done = 0; cur->converged = 0; cur->iters_run = 0; cur->fail = 0;
        }
        __syncthreads();
        return false;
    }
    // Request this thread's share of the previous launch's partial sums together with the state it guards
    // (one memory round trip instead of two); they are simply unused when the loop has already finished.
    const int comp = t & 31, slice = t >> 5;
    const uint32_t n_rows = a.n_prev ? a.n_prev : a.n_partials;      // rows of launch k - 1
    // the state first: loads complete in issue order, so testing `done` then waits for nothing younger
    const int prev_done = prev->done;
    const double prev_pose_t = t < 16 ? prev->pose[t] : 0.0;
    double pv[32];
    {
        const double* part = a.partials + (size_t)((k + 1) & 1) * kMaxPartials * kAccum + comp;
        const uint32_t last = n_rows ? n_rows - 1u : 0u;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const uint32_t b = (uint32_t)slice + 8u * u;
            pv[u] = part[(size_t)(b < last ? b : last) * kAccum];      // unconditional: rows past the end are masked in the fold
        }
    }
    // fixed-order reduction of the partial sums of launch k-1.  Straight-line code (selects, no branches): every
    // conditional region here made the compiler either sink the loads below it or wait for all of them at its join.
    const double* const part0 = a.partials + (size_t)((k + 1) & 1) * kMaxPartials * kAccum + comp;
    const double red = *(a.reduced ? a.reduced + comp : part0);       // sharded mode: sums already all-reduced
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += ((uint32_t)slice + 8u * u < n_rows) ? pv[u] : 0.0;
    // more than 256 blocks (scans beyond 65 536 points): sixteen loads in flight per step, added in the same order as a plain
    // loop (one load per step cost 11 us of every launch on a 131 072-point scan)
    for (uint32_t b0 = (uint32_t)slice + 256u; b0 < n_rows; b0 += 128u) {
        const uint32_t last = n_rows - 1u;
        double w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const uint32_t b = b0 + 8u * u; w[u] = part0[(size_t)(b < last ? b : last) * kAccum]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (b0 + 8u * u < n_rows) ? w[u] : 0.0;
    }
    if (a.reduced) acc = slice == 0 ? red : 0.0;
    sh_sum[slice * 32 + comp] = acc;
    if (t < 16) sh->pose[t] = prev_pose_t;
    __syncthreads();
    if (tl) tl[7] = wall_clock64();
    // The partial sums have arrived: now fetch this lane's neighbour-cache entry, straight into LDS, while the normal
    // equations are solved.  Issued from inline asm on purpose: the compiler's wait-count bookkeeping treats an LDS-DMA
    // in flight as "wait for everything" at every barrier and at every use of any other load; the consumer waits by hand.
    if (pre_dst) {
        const float4* src = pre_src ? pre_src : reinterpret_cast<const float4*>(a.partials) + (size_t)t * kEntryVec;   // harmless address
#pragma unroll
        for (int f = 0; f < kEntryVec; ++f) lds_dma16(src + f, pre_dst + f * 256);
    }
    // (the tests of the previous state come only now: placed before the loads above they made the compiler sink the
    // partial-sum loads below the branch, i.e. two dependent round trips instead of one)
    if (prev_done) {
        if (t == 0) sh->done = 1;
        if (blockIdx.x == 0 && t == 0) { *cur = *prev; __hip_atomic_store(&a.result->progress, (k << 1) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        __syncthreads();
        return true;
    }
    if (kAblation && (a.ablate & 4)) {
        if (t == 0) { sh->done = k >= a.c.iters; sh->big_step = 0; }
        if (blockIdx.x == 0 && t == 0) { *cur = *prev; cur->done = k >= a.c.iters; cur->iters_run = k; }
        __syncthreads();
        return k >= a.c.iters;
    }
    if (t < 32) {
        double v = sh_sum[t];
#pragma unroll
        for (int s = 1; s < 8; ++s) v += sh_sum[s * 32 + t];
        sh_sum[t] = v;
    }
    __syncthreads();
    // ---- x = (JtJ)^-1 (-JtE)  (LoamRegister.cpp:198).  The reference calls Eigen's LDLT; JtJ is symmetric
    // positive definite, so plain Gauss-Jordan elimination gives the same x to rounding.  It runs with one lane
    // per entry of the augmented 6x7 system (42 lanes of wave 0, 6 short steps) instead of ~1.5k dependent
    // instructions on a single lane while 255 threads of every block wait.  When the geometry does NOT constrain all six
    // degrees of freedom (a single plane, a corridor: a pivot collapses) the answer depends on how the factorisation
    // treats the null space, so that case is redone exactly as Eigen does it: pivoted LDLT, zero pivots dropped. ----
    double* const sh_m = sh_sum + 64;      // 42 entries of the augmented matrix
    double* const sh_x = sh_sum + 112;     // 6 entries of the solution
    if (t < 64) {
        const int i = t / 7, j = t - 7 * i;
        const bool in = t < 42;
        double v = 0.0;
        if (in) {
            const int r = i < j ? i : j, c = i < j ? j : i;
            v = j < 6 ? sh_sum[r * 6 - (r * (r - 1)) / 2 + (c - r)] : -sh_sum[21 + i];
        }
        // diagonal entries of the packed upper triangle: 0, 6, 11, 15, 18, 20
        const double maxd = fmax(fmax(fmax(fabs(sh_sum[0]), fabs(sh_sum[6])), fmax(fabs(sh_sum[11]), fabs(sh_sum[15]))),
                                 fmax(fabs(sh_sum[18]), fabs(sh_sum[20])));
        bool weak = false;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
            if (in) sh_m[t] = v;
            __builtin_amdgcn_wave_barrier();
            const double pk = sh_m[kk * 7 + kk], aik = in ? sh_m[i * 7 + kk] : 0.0, akj = in ? sh_m[kk * 7 + j] : 0.0;
            __builtin_amdgcn_wave_barrier();
            weak = weak || !(fabs(pk) > 1e-6 * maxd);        // also catches NaN
            const double rp = 1.0 / pk;                      // one division per elimination step instead of two
            if (in) v = (i == kk) ? akj * rp : v - (aik * rp) * akj;
        }
        if (in && j == 6) sh_x[i] = v;
        if (t == 0) sh_x[6] = weak ? 1.0 : 0.0;              // every lane saw the same pivots
    }
    __syncthreads();
    if (tl) tl[11] = wall_clock64();      // normal equations solved
    if (t == 0) {
        double JtJ[36], rhs[6], x[6];
        int q = 0;
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { JtJ[r * 6 + c] = JtJ[c * 6 + r] = sh_sum[q++]; }
        for (int r = 0; r < 6; ++r) { rhs[r] = -sh_sum[21 + r]; x[r] = sh_x[r]; }
        if (sh_x[6] != 0.0) ldlt6_solve(JtJ, rhs, x);         // rank-deficient or ill-conditioned: Eigen's LDLT, step for step
        const double n = sh_sum[27];
        int done = 0, conv = 0, fail = 0;
        double pose[16];
        for (int i = 0; i < 16; ++i) pose[i] = sh->pose[i];      // previous pose, staged below
        if (a.reduced && sh_sum[kSlotRankFail] != 0.0) { done = 1; fail = 2; for (int r = 0; r < 6; ++r) x[r] = 0.0; }   // a rank has no index: every rank stops here
        else if (sh_sum[kSlotEscapes] != 0.0) { done = 1; fail = 3; for (int r = 0; r < 6; ++r) x[r] = 0.0; }        // the clamped index was too small
        else if (n < 6.0) { done = 1; fail = 1; for (int r = 0; r < 6; ++r) x[r] = 0.0; }   // LoamRegister.cpp:173-176
        else {
            const double np = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            const double nr = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
            if (a.c.early_exit && np <= a.c.pos_conv && nr <= a.c.rot_conv) { done = 1; conv = 1; }   // :202-206
            else {
                double E[16], out[16];
                se3_exp(x, E);                                     // :213-216
                for (int c = 0; c < 4; ++c)
                    for (int r = 0; r < 4; ++r) {
                        double s = 0;
                        for (int kk = 0; kk < 4; ++kk) s += E[kk * 4 + r] * pose[c * 4 + kk];
                        out[c * 4 + r] = s;
                    }
                for (int i = 0; i < 16; ++i) pose[i] = out[i];
            }
        }
        if (k >= a.c.iters) done = 1;
        for (int i = 0; i < 16; ++i) sh->pose[i] = pose[i];
        sh->done = done;
        // A cache entry proves its five neighbours while the query has moved by less than the gap between the 5th and the 9th neighbour
        // distance -- centimetres on a 0.5 m map.  A step that moves a point 10 m from the sensor by more than kBigStep makes four
        // entries in five fail (launch 1 after a 0.3 m / 2 degree guess: 52 682 of 65 536): then nobody reads or tests an entry and
        // every lane searches its own query, without the exchange.  A performance decision only -- a search is exact whatever the cache
        // would have said.  Measured (A/B on one box, profiles/r03_notes.md): launch 1 30.5 -> 28.6 us, and launch 2 searches 7 140
        // instead of 11 682 queries because every entry is now anchored at the pose after the big step; 3 803 -> 3 894 scans/s.
        sh->big_step = (!done && sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) + 10.0 * sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]) > kBigStep) ? 1 : 0;
        if (blockIdx.x == 0) {
            for (int i = 0; i < 16; ++i) cur->pose[i] = pose[i];
            cur->done = done; cur->converged = conv; cur->fail = fail; cur->iters_run = k;
            // (the host may be queueing launches only a few ahead of this word: run_loam)
            __hip_atomic_store(&a.result->progress, (k << 1) | (done ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (a.trace) {
                LoamTrace* tr = &a.trace[k - 1];
                for (int i = 0; i < 36; ++i) tr->JtJ[i] = JtJ[i];
                for (int i = 0; i < 6; ++i) { tr->JtE[i] = -rhs[i]; tr->x[i] = x[i]; }
                tr->n = (int64_t)n;
                tr->cache_hits = (int64_t)sh_sum[28]; tr->searches = (int64_t)sh_sum[29];
            }
        }
    }
    __syncthreads();
    return sh->done != 0;
}

// ------------------------------------------------------------------------------
// the iteration kernel
// ------------------------------------------------------------------------------
// kGroup = candidates in flight per lane of the per-lane search.  <8, 1> is the default (one wave per SIMD -- which is all
// that 65 536 queries give anyway); <4, 2> fits two waves per SIMD so that the blocks of ANOTHER handle's launch can share the
// CUs: a little slower alone, more scans/s with several handles in flight (pcr_params.loam_coresident = 1).
template <int kGroup, int kWavesPerSimd>
__global__ __launch_bounds__(256, kWavesPerSimd) void loam_iterate_kernel(const LoamArgs a, const int k) {
    __shared__ double sh_sum[8 * 32];
    __shared__ Prologue sh_pro;
    // The staging area of the prefetched cache entries (48 KB, dead once every lane has read its entry) shares its
    // LDS with the search scratch and the miss exchange (first written after that), so that two blocks -- e.g. of two scans
    // on two streams -- fit on a CU.
    struct SearchLds { KnnRuns knn; MissExchange ex; };
    constexpr size_t kStage = (size_t)kEntryVec * 256 * sizeof(float4);
    __shared__ __attribute__((aligned(16))) unsigned char sh_ov[sizeof(SearchLds) > kStage ? sizeof(SearchLds) : kStage];
    KnnRuns& sh_knn = reinterpret_cast<SearchLds*>(sh_ov)->knn;
    MissExchange& sh_ex = reinterpret_cast<SearchLds*>(sh_ov)->ex;   // also holds the rows: [component][point] s*J (6), s*d, accepted flag
    double* const sh_rows = sh_ex.u.rows;
    const int tid = threadIdx.x;
    unsigned long long* const tl = (a.timeline && tid == 0) ? a.timeline + ((size_t)k * kMaxPartials + blockIdx.x) * kTimelineSlots : nullptr;
    if (tl) tl[0] = wall_clock64();
    // XCD-aware mapping: consecutive logical blocks (adjacent lidar rings) share an XCD's L2
    uint32_t blk = blockIdx.x;
    if ((gridDim.x & 7u) == 0) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // Everything the first round needs from memory is requested before/inside the prologue, so that the scan point and
    // the 128-byte neighbour-cache entry arrive while the previous iteration's normal equations are being solved.
    // The entry goes straight to LDS ([field][thread], 48 KB): held in registers it was spilled to AGPRs, which made
    // the wave wait for it before the prologue had even started.
    float4* const sh_pre = reinterpret_cast<float4*>(sh_ov);
    const GridHeader h = *a.grid.hdr;      // uniform: scalar loads, in flight during the prologue
    float pre_x = 0.f, pre_y = 0.f, pre_z = 0.f;
    const bool use_cache = k > 0 && a.nn_cache != nullptr;
    const float4* pre_src = nullptr;
    // half blocks (LoamArgs::half): 128 queries per block, owned by the lower half of its threads; the upper half has none of its own and
    // only takes its slice of the searches (loam_point: a block with 128 posted queries searches each with two lanes, in different waves)
    const uint32_t qpb = a.half ? 128u : 256u;
    const bool owner_lane = (uint32_t)tid < qpb;
    {
        const uint32_t q = blk * qpb + (uint32_t)tid;
        // unconditional load of a clamped index: a conditional one makes the wave wait for it at the join
        const float* sp = a.n_src ? a.src + (size_t)(q < a.n_src ? q : a.n_src - 1u) * a.src_stride
                                  : reinterpret_cast<const float*>(a.partials);      // empty scan: any readable address
        pre_x = sp[0]; pre_y = sp[1]; pre_z = sp[2];
        pre_src = (use_cache && owner_lane && q < a.n_src) ? reinterpret_cast<const float4*>(a.nn_cache + q) : nullptr;
    }
    if (loam_prologue(a, k, sh_sum, &sh_pro, tl, pre_src, use_cache ? sh_pre + (tid & ~63) : nullptr)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // never leave with an LDS-DMA in flight
        return;
    }
    if (tl) tl[1] = wall_clock64();
    double pose[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pose[i] = sh_pro.pose[i];

    // This thread accumulates ONE entry of the normal equations over a 32-point chunk:
    // e < 21: JtJ(er,ec) upper triangle; 21..26: JtE(er) = sum row[er]*row[6]; 27: accepted count.
    const int e = tid & 31, ch = tid >> 5;
    int er = 7, ec = 7;
    {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int c = r; c < 6; ++c) { if (q == e) { er = r; ec = c; } ++q; }
        }
        if (e >= 21 && e < 27) { er = e - 21; ec = 6; }
    }
    double acc = 0.0;
    uint32_t n_hit = 0, n_search = 0, n_esc = 0;
    for (uint32_t base = blk * qpb; base < a.n_src; base += gridDim.x * qpb) {
        const uint32_t q = base + tid;
        const bool valid = owner_lane && q < a.n_src;
        double row[7] = {0, 0, 0, 0, 0, 0, 0};
        uint32_t nn[5] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        int how = 0;
        bool esc = false;
        float sx = pre_x, sy = pre_y, sz = pre_z;
        union { NnCacheEntry e; float4 v[kEntryVec]; } ce;
        ce.e.flags = 0;
        if (base == blk * qpb) {
            // EVERY wave waits for its own LDS-DMA, also one without a single valid query (its lanes fetched a dummy address):
            // the staging area is reused right after the barrier, and a transfer still in flight would land in the search
            // scratch of the other waves.
            if (use_cache) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (use_cache && valid && !sh_pro.big_step) {
#pragma unroll
                for (int f = 0; f < kEntryVec; ++f) ce.v[f] = sh_pre[f * 256 + tid];
            }
            __syncthreads();          // the staging area is about to be reused by the search scratch and the exchange
        } else {                      // later rounds of a grid-stride launch
            sx = sy = sz = 0.f;
            if (valid) {
                const float* sp = a.src + (size_t)q * a.src_stride;
                sx = sp[0]; sy = sp[1]; sz = sp[2];
                if (use_cache) ce.e = a.nn_cache[q];
            }
        }
        const bool all_search = sh_pro.big_step != 0 && base == blk * qpb && !a.half;      // (later rounds of a grid-stride launch go the ordinary way)
        const int st = loam_point<kGroup>(a, h, pose, sx, sy, sz, valid, ce.e, use_cache && valid && !all_search, sh_knn, sh_ex, row, nn, q, &how, &esc,
                                          base == blk * qpb ? tl : nullptr, all_search);
        if (valid && (a.dbg_status || a.dbg_nn || a.dbg_rows)) {
            const size_t oi = (size_t)q;
            if (a.dbg_status) a.dbg_status[oi] = (int8_t)st;
            if (a.dbg_nn) { for (int j = 0; j < 5; ++j) a.dbg_nn[oi * 5 + j] = (int32_t)nn[j]; }
            if (a.dbg_rows) { for (int j = 0; j < 7; ++j) a.dbg_rows[oi * 7 + j] = st == 0 ? row[j] : 0.0; }
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) sh_rows[c * kRowStride + tid] = st == 0 ? row[c] : 0.0;
        sh_rows[7 * kRowStride + tid] = st == 0 ? 1.0 : 0.0;
        __syncthreads();
        n_hit += (uint32_t)__popcll(__ballot(how == 1)); n_search += (uint32_t)__popcll(__ballot(how == 2));   // per wave
        if (h.clamped) n_esc += (uint32_t)__popcll(__ballot(esc));
        if (e < 28) {
            const double* ra = sh_rows + er * kRowStride + ch * 32;
            const double* rb = sh_rows + ec * kRowStride + ch * 32;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) acc += ra[i] * rb[i];
        }
        __syncthreads();
    }
    if (tl) tl[5] = wall_clock64();
    sh_sum[ch * 32 + e] = e < 28 ? acc : 0.0;
    // statistics ride in the two spare components (exact small integers in f64)
    __shared__ uint32_t sh_cnt[12];
    if ((tid & 63) == 0) { sh_cnt[(tid >> 6) * 2] = n_hit; sh_cnt[(tid >> 6) * 2 + 1] = n_search; sh_cnt[8 + (tid >> 6)] = n_esc; }
    __syncthreads();
    if (tid < 32) {
        double v = sh_sum[tid];
#pragma unroll
        for (int c = 1; c < 8; ++c) v += sh_sum[c * 32 + tid];
        if (tid == 28) v = (double)(sh_cnt[0] + sh_cnt[2] + sh_cnt[4] + sh_cnt[6]);
        if (tid == 29) v = (double)(sh_cnt[1] + sh_cnt[3] + sh_cnt[5] + sh_cnt[7]);
        if (tid == kSlotEscapes) v = (double)(sh_cnt[8] + sh_cnt[9] + sh_cnt[10] + sh_cnt[11]);
        a.partials[((size_t)(k & 1) * kMaxPartials + blockIdx.x) * kAccum + tid] = v;
    }
    if (tl) tl[6] = wall_clock64();
}

// last launch: prologue only, then T2SE3 and the output pose (LoamRegister.cpp:220)
__global__ __launch_bounds__(256) void loam_finalize_kernel(const LoamArgs a, const int k) {
    __shared__ double sh_sum[8 * 32];
    __shared__ Prologue sh_pro;
    loam_prologue(a, k, sh_sum, &sh_pro);
    if (threadIdx.x == 0) {
        double T[16];
        for (int i = 0; i < 16; ++i) T[i] = sh_pro.pose[i];
        t2se3(T);
        const LoamState* cur = &a.state[k & 1];   // written by this thread in the prologue
        LoamResult* r = a.result;
        for (int i = 0; i < 16; ++i) r->pose[i] = T[i];
        r->converged = cur->converged; r->iters_run = cur->iters_run; r->fail = cur->fail;
        r->grid_overflow = a.grid.hdr->overflow; r->grid_empty = a.grid.hdr->empty; r->grid_cells = a.grid.hdr->n_cells;
        r->grid_stale = a.grid.hdr->stale;
        // completion marker: the host polls this word of host-mapped memory, so everything above has to be visible first
        __threadfence_system();
        __hip_atomic_store(&r->pad, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// This thread's share (component comp, rows slice, slice + 8, ...) of the rows launch k wrote, added in row order.  The state word and the first 32
// rows are requested together and unconditionally (rows past the end repeat the last one and are masked in the sum): a load per loop step behind the
// test of `done` was a chain of up to 33 round trips in a kernel every linearisation of a sharded call waits for.
__device__ __forceinline__ double loam_fold_own_rows(const LoamArgs& a, int k, int comp, int slice) {
    const int done = a.state[k & 1].done;
    const double* const part0 = a.partials + (size_t)(k & 1) * kMaxPartials * kAccum + comp;
    const uint32_t last = a.n_partials ? a.n_partials - 1u : 0u;
    double pv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) { const uint32_t b = (uint32_t)slice + 8u * u; pv[u] = part0[(size_t)(b < last ? b : last) * kAccum]; }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += ((uint32_t)slice + 8u * u < a.n_partials) ? pv[u] : 0.0;
    for (uint32_t b = (uint32_t)slice + 256u; b < a.n_partials; b += 8u) acc += part0[(size_t)b * kAccum];      // (scans beyond 65 536 points)
    return done ? 0.0 : acc;
}

// sharded (multi-GPU) mode: fold this rank's partial sums into kAccum doubles so that
// RCCL can all-reduce them before the next launch's prologue reads a.reduced
__global__ __launch_bounds__(256) void loam_reduce_kernel(const LoamArgs a, const int k, double* __restrict__ out) {
    __shared__ double sh_sum[8 * 32];
    const int t = threadIdx.x, comp = t & 31, slice = t >> 5;
    const double acc = loam_fold_own_rows(a, k, comp, slice);
    sh_sum[slice * 32 + comp] = acc;
    __syncthreads();
    if (t < 32) {
        double v = sh_sum[t];
#pragma unroll
        for (int s = 1; s < 8; ++s) v += sh_sum[s * 32 + t];
        if (t == kSlotRankFail) v = (a.rank_fail || a.grid.hdr->overflow) ? 1.0 : 0.0;
        out[t] = v;
    }
}

// (the exchange itself: peer_exchange.h -- shared with the NDT and VGICP loops)
// the fold of this rank's partial sums (loam_reduce_kernel) + the exchange, one launch per linearisation
__global__ __launch_bounds__(256) void loam_peer_exchange_kernel(const LoamArgs a, const int k, const PeerComm pc, const double seq, double* __restrict__ out) {
    __shared__ double sh_sum[8 * 32];
    const int t = threadIdx.x, comp = t & 31, slice = t >> 5;
    const double acc = loam_fold_own_rows(a, k, comp, slice);
    sh_sum[slice * 32 + comp] = acc;
    __syncthreads();
    double v = 0.0;
    if (t < 32) {
        v = sh_sum[t];
#pragma unroll
        for (int s = 1; s < 8; ++s) v += sh_sum[s * 32 + t];
        if (t == kSlotRankFail) v = (a.rank_fail || a.grid.hdr->overflow) ? 1.0 : 0.0;
    }
    const bool ok = peer_exchange_block(pc, seq, nullptr, v, kAccum, 0, out);
    if (!ok && t == kSlotRankFail) out[t] = 1.0;      // every rank that timed out stops its loop (fail = 2); the sums are not used then
}
// n <= 64 doubles of a device buffer, in place (op 0 = sum, 1 = max): the agreement exchanges around the loops
__global__ __launch_bounds__(256) void peer_allreduce_kernel(double* __restrict__ inout, const int n, const int op, const PeerComm pc, const double seq) {
    const int t = threadIdx.x;
    const double v = t < n ? inout[t] : 0.0;
    peer_exchange_block(pc, seq, nullptr, v, n, op, inout);
}
hipError_t loam_launch_peer_exchange(const LoamArgs& a, int k, const PeerComm& pc, double seq, double* d_out, hipStream_t s) {
    hipLaunchKernelGGL(loam_peer_exchange_kernel, dim3(1), dim3(256), 0, s, a, k, pc, seq, d_out);
    return hipGetLastError();
}
hipError_t peer_launch_allreduce(double* d_inout, int n, int op, const PeerComm& pc, double seq, hipStream_t s) {
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(256), 0, s, d_inout, n, op, pc, seq);
    return hipGetLastError();
}

uint32_t loam_grid_blocks(uint32_t n_src) {
    uint32_t b = (n_src + 255) / 256;
    if (b < 1) b = 1;
    if (b > (uint32_t)kMaxPartials) b = kMaxPartials;
    return b;
}

// start/stop (optional): events that the packet processor stamps at the kernel's own begin and end (hipExtLaunchKernelGGL),
// i.e. what a profiler reports as the kernel's duration -- events recorded around an ordinary launch also contain the
// dispatch latency (~2 us here).
hipError_t loam_launch_iteration(const LoamArgs& a_in, int k, hipStream_t s, hipEvent_t start, hipEvent_t stop, bool allow_half) {
    LoamArgs a = a_in;
    a.half = 0; a.n_prev = 0;
    // (development: launch 0 -- every query searches -- as twice the blocks of 128 queries, two lanes per query, two waves per SIMD)
    static const int half0 = dev_env("PCR_LOAM_HALF0") ? atoi(dev_env("PCR_LOAM_HALF0")) : 0;
    const bool can_half = allow_half && half0 > 0 && !a.reduced && !a.coresident && 2u * a.n_partials <= (uint32_t)kMaxPartials && !(start && stop);
    if (can_half && k == 1) a.n_prev = 2u * a.n_partials;
    if (can_half && k == 0) {
        a.half = 1;
        if (half0 == 1) hipLaunchKernelGGL((loam_iterate_kernel<8, 2>), dim3(2u * a.n_partials), dim3(256), 0, s, a, k);
        else hipLaunchKernelGGL((loam_iterate_kernel<4, 2>), dim3(2u * a.n_partials), dim3(256), 0, s, a, k);
        return hipGetLastError();
    }
    if (start && stop) {
        if (a.coresident) hipExtLaunchKernelGGL((loam_iterate_kernel<4, 2>), dim3(a.n_partials), dim3(256), 0, s, start, stop, 0, a, k);
        else hipExtLaunchKernelGGL((loam_iterate_kernel<8, 1>), dim3(a.n_partials), dim3(256), 0, s, start, stop, 0, a, k);
        return hipGetLastError();
    }
    if (a.coresident) hipLaunchKernelGGL((loam_iterate_kernel<4, 2>), dim3(a.n_partials), dim3(256), 0, s, a, k);
    else hipLaunchKernelGGL((loam_iterate_kernel<8, 1>), dim3(a.n_partials), dim3(256), 0, s, a, k);
    return hipGetLastError();
}
hipError_t loam_launch_finalize(const LoamArgs& a, int k, hipStream_t s) {
    hipLaunchKernelGGL(loam_finalize_kernel, dim3(1), dim3(256), 0, s, a, k);
    return hipGetLastError();
}
hipError_t loam_launch_reduce(const LoamArgs& a, int k, double* d_out, hipStream_t s) {
    hipLaunchKernelGGL(loam_reduce_kernel, dim3(1), dim3(256), 0, s, a, k, d_out);
    return hipGetLastError();
}

}  // namespace pcr
