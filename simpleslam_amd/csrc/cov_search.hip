// cov_search.hip -- the 20 nearest neighbours of every point of a cloud (fast_gicp_impl.hpp:241-262: FLANN k-NN over the cloud itself),
// as TWO classes of queries, and the covariance arithmetic as a kernel of its own (gfx950, hand-written HIP).
//
// vgicp_cov_kernel (vgicp.hip) gives every query a lane and lets that lane walk ring after ring, level after level.  On a lidar scan --
// four orders of magnitude of density -- the lanes of a wave then do very different amounts of work: measured per wave, mean 134 us against
// a maximum of 358 us (profiles/r03_notes.md), the kernel lasting as long as its slowest wave with one wave per SIMD.  Here:
//
//   A  cov_ring1_kernel   lane per query, ring 1 of the fine level ONLY (the 3 x 3 x 3 block: nine x-runs walked as one flat candidate
//                         stream, kGroup candidates in flight).  The list is twenty 32-bit SCREENING keys
//                         (float distance bits, low bits cleared | number of the candidate in the stream), kept by v_min_u32 / v_max_u32,
//                         two instructions per slot and only when some lane of the wave has a candidate below its 20th key.  The
//                         smallest key that is NOT listed is kept too: the twenty are then recomputed exactly (FLANN's float
//                         arithmetic, ties on the lower index) and are the answer iff the 20th is strictly nearer than everything
//                         unlisted AND than the faces of the block.  Every other query -- too sparse a neighbourhood, a tie at
//                         the edge of the list, more candidates than the key numbers -- goes to a queue.
//   B  cov_wave_kernel    WAVE per queued query, lanes = candidates: ring after ring, level after level like the old kernel, but a ring's
//                         rows are fetched 64 at a time and its candidates 64 at a time, keys exact (distance bits << 32 | index); a
//                         chunk's candidates below the current 20th key are merged into the list by rank counting in LDS.
//   C  cov_from_nbr_kernel  lane per query: the covariance arithmetic (f64 sums in neighbour order, Jacobi eigen-decomposition, PLANE
//                         regularisation) from the neighbour lists A and B leave in HBM.
//
// Results are the old kernel's bit for bit (same neighbours in the same order into the same arithmetic: cov_math.h).
#include <string.h>
#include <algorithm>

#include "pcr_internal.h"
#include "cov_math.h"

namespace pcr {

namespace {

constexpr uint32_t kKeyEmpty = 0xffffffffu;
constexpr uint32_t kNbrNone = 0xffffffffu;      // unfilled slot of a neighbour list (a cloud of fewer than K points)
constexpr uint32_t kNbrSkip = 0xfffffffeu;      // nbr[0][j]: the point lies outside the prepared region, no covariance is wanted
constexpr int kMaxIdBits = 13;                  // a lane's stream numbers at most 8 192 candidates (10 mantissa bits left to screen with)

// rows of the 3 x 3 (y, z) neighbourhood, centre first: (dy + 1) | (dz + 1) << 2 per row
constexpr unsigned long long kRowOrder = 0xA82091645ull;
__device__ __forceinline__ constexpr int row_dy(int r) { return (int)((kRowOrder >> (4 * r)) & 3) - 1; }
__device__ __forceinline__ constexpr int row_dz(int r) { return (int)((kRowOrder >> (4 * r + 2)) & 3) - 1; }

constexpr int kRuns = 11;                       // the query's own cell, its two x-neighbours, the eight other rows of the block
struct CovRuns { uint2 run[kRuns][256]; };      // {first position, length} of every lane's runs, in the order they are streamed

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m, 64));
    return v;
}
__device__ __forceinline__ uint32_t lane_rank(unsigned long long m) {      // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// FLANN L2_Simple<float>: ((dx^2 + dy^2) + dz^2), every operation rounded (no contraction)
__device__ __forceinline__ float flann_dist(float qx, float qy, float qz, const float4& p) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    float d = __fmul_rn(dx, dx);
    d = __fadd_rn(d, __fmul_rn(dy, dy));
    d = __fadd_rn(d, __fmul_rn(dz, dz));
    return d;
}

// squared distance (shaved: float candidates) from q to the nearest face of the block [c - r, c + r] that has cells behind it;
// < 0: the block covers the whole grid
__device__ __forceinline__ double block_bound_sq(const GridHeader& h, float qx, float qy, float qz, int cx, int cy, int cz, int r) {
    const double o0 = h.org[0] + h.shift, o1 = h.org[1] + h.shift, o2 = h.org[2] + h.shift;
    double bound = 1e300;
    if (cx - r > 0) bound = fmin(bound, (double)qx - (o0 + (double)(cx - r)) * h.cell);
    if (cx + r < h.dims[0] - 1) bound = fmin(bound, (o0 + (double)(cx + r + 1)) * h.cell - (double)qx);
    if (cy - r > 0) bound = fmin(bound, (double)qy - (o1 + (double)(cy - r)) * h.cell);
    if (cy + r < h.dims[1] - 1) bound = fmin(bound, (o1 + (double)(cy + r + 1)) * h.cell - (double)qy);
    if (cz - r > 0) bound = fmin(bound, (double)qz - (o2 + (double)(cz - r)) * h.cell);
    if (cz + r < h.dims[2] - 1) bound = fmin(bound, (o2 + (double)(cz + r + 1)) * h.cell - (double)qz);
    if (bound >= 1e299) return -1.0;
    return bound > 0 ? bound * bound * (1.0 - 1e-5) : 0.0;
}

template <int kCtrl> __device__ __forceinline__ uint32_t quad_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kCtrl, 0xf, 0xf, false); }
template <int kCtrl> __device__ __forceinline__ unsigned long long quad_dpp64(unsigned long long v) {
    return ((unsigned long long)quad_dpp<kCtrl>((uint32_t)(v >> 32)) << 32) | (unsigned long long)quad_dpp<kCtrl>((uint32_t)v);
}
// maximum over the wave as a scalar: quads, halves of rows and rows by DPP (quad_perm, row_half_mirror, row_mirror), the four rows by readlane
__device__ __forceinline__ uint32_t wave_max_uniform(uint32_t v) {
    v = max(v, quad_dpp<0xB1>(v)); v = max(v, quad_dpp<0x4E>(v)); v = max(v, quad_dpp<0x141>(v)); v = max(v, quad_dpp<0x140>(v));
    return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

#define COV_CSWAP(a, b) { const bool sw_ = (b) < (a); const unsigned long long lo_ = sw_ ? (b) : (a), hi_ = sw_ ? (a) : (b); (a) = lo_; (b) = hi_; any_sw |= sw_; }

// ------------------------------------------------------------------------------
// A: kLpq adjacent lanes per query, ring 1 of the fine level
// ------------------------------------------------------------------------------
// One lane per query means one wave per SIMD for a 65 536-point scan, every group of candidates a memory round trip that nothing hides and
// every vector instruction issued at half rate (measured: 200 us).  With kLpq lanes per query each lane streams its slice of the block's
// candidates; the kLpq lists are then merged in registers: partner's list through DPP (quad_perm), the twenty smallest of the two as
// min(mine[i], theirs[19 - i]) -- a sequence that rises and then falls -- put in order by a bitonic merge network for twenty elements.
template <int kLo, int kN>
struct BitonicDesc {      // sorts a[kLo .. kLo + kN), a sequence that rises and then falls, into descending order (Lang's network for arbitrary n)
    static __device__ __forceinline__ void run(uint32_t* a) {
        if constexpr (kN > 1) {
            constexpr int kM = kN > 16 ? 16 : (kN > 8 ? 8 : (kN > 4 ? 4 : (kN > 2 ? 2 : 1)));      // greatest power of two below kN
#pragma unroll
            for (int i = kLo; i < kLo + kN - kM; ++i) { const uint32_t hi = max(a[i], a[i + kM]), lo = min(a[i], a[i + kM]); a[i] = hi; a[i + kM] = lo; }
            BitonicDesc<kLo, kM>::run(a);
            BitonicDesc<kLo + kM, kN - kM>::run(a);
        }
    }
};
// key[0..kCovK) ascending + key[kCovK] (the smallest key not listed) of this lane and of its partner (kCtrl: quad_perm) -> those of the union
template <int kCtrl>
__device__ __forceinline__ void merge_with_partner(uint32_t key[kCovK + 1]) {
    uint32_t c[kCovK];
    uint32_t rest = min(key[kCovK], quad_dpp<kCtrl>(key[kCovK]));
#pragma unroll
    for (int i = 0; i < kCovK; ++i) {
        const uint32_t theirs = quad_dpp<kCtrl>(key[kCovK - 1 - i]);      // (the partner runs the same code: its key[19 - i])
        c[i] = min(key[i], theirs);
        rest = min(rest, max(key[i], theirs));
    }
    BitonicDesc<0, kCovK>::run(c);
#pragma unroll
    for (int i = 0; i < kCovK; ++i) key[i] = c[kCovK - 1 - i];
    key[kCovK] = rest;
}

template <int kGroup, int kLpq>
__global__ __launch_bounds__(256, kLpq) void cov_ring1_kernel(GridView g, uint32_t n_sorted_max, uint32_t* __restrict__ nbr, uint32_t n_cap,
                                                        uint32_t* __restrict__ queue, unsigned long long* __restrict__ queue_seed,
                                                        uint32_t* __restrict__ queue_count, const RoiView roi, uint32_t dense_limit) {
    static_assert(kLpq == 1 || kLpq == 2 || kLpq == 4, "lanes per query");
    constexpr int kShift = kLpq == 4 ? 2 : (kLpq == 2 ? 1 : 0);
    constexpr int kPer = kCovK / kLpq;      // listed candidates each lane of a query looks up again
    __shared__ CovRuns sh;
    const GridHeader h = *g.hdr;
    if (h.empty || h.overflow || h.stale) return;
    GridHeader lat;
    if (roi.mask) lat = *roi.lat;
    if (roi.mask && (lat.stale || lat.overflow)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t part = (uint32_t)tid & (uint32_t)(kLpq - 1);
    const uint32_t n = min(g.cell_start[h.n_cells], n_sorted_max);
    const uint32_t j = (blockIdx.x * 256u + (uint32_t)tid) >> kShift;
    bool active = j < n;
    const float4 q = g.pts[active ? j : 0u];
    if (active && roi.mask && !roi_holds_point(roi, lat, (double)q.x, (double)q.y, (double)q.z)) { if (part == 0u) nbr[j] = kNbrSkip; active = false; }
    const int d0 = h.dims[0], d1 = h.dims[1], d2 = h.dims[2];
    const double fx = floor((double)q.x / h.cell - h.shift) - h.org[0], fy = floor((double)q.y / h.cell - h.shift) - h.org[1],
                 fz = floor((double)q.z / h.cell - h.shift) - h.org[2];
    const int cx = (int)fmin(fmax(fx, 0.0), (double)(d0 - 1)), cy = (int)fmin(fmax(fy, 0.0), (double)(d1 - 1)), cz = (int)fmin(fmax(fz, 0.0), (double)(d2 - 1));
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, d0 - 1);
    // Runs in the order they are streamed: the query's OWN cell first, then its two x-neighbours, then the other eight rows -- the own cell
    // holds most of the answer, so the bound that lets a candidate be skipped is tight after the first few steps.
    uint32_t ra[kRuns], rb[kRuns];
    {
        const uint32_t row_c = active ? ((uint32_t)cz * (uint32_t)d1 + (uint32_t)cy) * (uint32_t)d0 : 0u;
        const uint32_t a0 = g.cell_start[active ? row_c + (uint32_t)x0 : 0u], a1 = g.cell_start[active ? row_c + (uint32_t)cx : 0u],
                       a2 = g.cell_start[active ? row_c + (uint32_t)cx + 1u : 0u], a3 = g.cell_start[active ? row_c + (uint32_t)x1 + 1u : 0u];
        ra[0] = a1; rb[0] = a2; ra[1] = a0; rb[1] = a1; ra[2] = a2; rb[2] = a3;
    }
#pragma unroll
    for (int r = 1; r < 9; ++r) {      // (the lanes of a query ask for the same words)
        const int z = cz + row_dz(r), y = cy + row_dy(r);
        const bool in = active && z >= 0 && z < d2 && y >= 0 && y < d1;
        const uint32_t row = in ? ((uint32_t)z * (uint32_t)d1 + (uint32_t)y) * (uint32_t)d0 : 0u;
        ra[r + 2] = g.cell_start[in ? row + (uint32_t)x0 : 0u]; rb[r + 2] = g.cell_start[in ? row + (uint32_t)x1 + 1u : 0u];
    }
    uint32_t total = 0;
#pragma unroll
    for (int r = 0; r < kRuns; ++r) { const uint32_t len = rb[r] > ra[r] ? rb[r] - ra[r] : 0u; sh.run[r][tid] = make_uint2(ra[r], len); total += len; }
    // (each lane reads back only what it wrote itself: no barrier)
    const uint32_t max_total = wave_max_uniform(total);
    const uint32_t bits = min(32u - (uint32_t)__clz((int)(max_total | 1u)), (uint32_t)kMaxIdBits);      // wave-uniform: 2^bits > every lane's count, or the cap
    const uint32_t idmask = (1u << bits) - 1u;
    // (dense_limit: a query with more candidates in its block than that goes to the queue at once -- a wave per query, lanes = candidates -- and
    //  its lanes lengthen nobody's walk here: next to the sensor a block holds 886 candidates against 277 on average, and a wave walks as far as its
    //  longest run)
    const bool too_many = total > idmask || total > dense_limit;
    uint32_t key[kCovK + 1];      // key[kCovK]: the smallest key that is NOT in the list
#pragma unroll
    for (int i = 0; i <= kCovK; ++i) key[i] = kKeyEmpty;
    // The stream is walked by the WAVE, not by its lanes: run after run, a run in steps of kLpq x kGroup positions -- the lanes of a query
    // take the kGroup-position pieces of a step in turn -- as many steps as the longest run of the wave needs (queries that are neighbours
    // in cell order have the same or similar runs).  What a lane does per candidate is then an add and a compare; a per-lane cursor over
    // the runs cost as much as the distance and the key together (measured: 52 instructions per candidate besides the list).
    struct Step { uint32_t addr0, seq0, nv; };      // first position in the sorted array, its number in the stream, valid positions (0 .. kGroup)
    int r_cur = -1;                 // wave-uniform iterator: run, step inside it, steps it has
    uint32_t i_cur = 0, steps_cur = 0;
    uint2 run_cur = make_uint2(0u, 0u);
    uint32_t prefix = 0, prefix_next = 0;      // numbers of the stream before the current run / after it
    auto advance = [&](Step& st) -> bool {
        ++i_cur;
        while (i_cur >= steps_cur) {
            ++r_cur;
            if (r_cur >= kRuns) return false;
            run_cur = sh.run[r_cur][tid];
            prefix = prefix_next; prefix_next += run_cur.y;
            steps_cur = (wave_max_uniform(too_many ? 0u : run_cur.y) + (uint32_t)(kLpq * kGroup) - 1u) / (uint32_t)(kLpq * kGroup);
            i_cur = 0;
        }
        const uint32_t off0 = (i_cur * (uint32_t)kLpq + part) * (uint32_t)kGroup;
        const uint32_t len = too_many ? 0u : run_cur.y;
        st.addr0 = run_cur.x + off0; st.seq0 = prefix + off0; st.nv = len > off0 ? len - off0 : 0u;
        return true;
    };
    Step st_next;
    i_cur = 0; steps_cur = 0;      // (the first advance() enters run 0: ++i_cur makes 1 >= 0)
    bool have = advance(st_next);
    float4 nx[kGroup];
#pragma unroll
    for (int u = 0; u < kGroup; ++u) nx[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) nx[u] = g.pts[(uint32_t)u < st_next.nv ? st_next.addr0 + (uint32_t)u : 0u];
    }
    uint32_t bound = kKeyEmpty;      // no key >= this can be among the query's twenty (see below)
    while (have) {
        float4 c[kGroup];
#pragma unroll
        for (int u = 0; u < kGroup; ++u) c[u] = nx[u];
        const Step st = st_next;
        have = advance(st_next);
        if (have) {
#pragma unroll
            for (int u = 0; u < kGroup; ++u) nx[u] = g.pts[(uint32_t)u < st_next.nv ? st_next.addr0 + (uint32_t)u : 0u];
        }
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
            // a SCREEN (contracted arithmetic, low bits cleared): within 1e-6 relative of FLANN's value, see the proof below
            const float dx = q.x - c[u].x, dy = q.y - c[u].y, dz = q.z - c[u].z;
            float d = dx * dx;
            d = __builtin_fmaf(dy, dy, d);
            d = __builtin_fmaf(dz, dz, d);
            uint32_t t = (__float_as_uint(d) & ~idmask) | (st.seq0 + (uint32_t)u);
            t = (uint32_t)u < st.nv ? t : kKeyEmpty;
            if (__any(t < bound)) {      // wave-uniform: some lane lists this candidate
#pragma unroll
                for (int k = 0; k < kCovK; ++k) { const uint32_t lo = min(key[k], t), hi = max(key[k], t); key[k] = lo; t = hi; }
            }
            key[kCovK] = min(key[kCovK], t);      // what is not listed: a candidate that never entered, or the one that fell off
        }
        // What the lanes of a query have seen together bounds its answer: the 20th key of the union is at most each lane's own 20th, and at
        // most the largest of their (20 / kLpq)-th keys (that many keys of each lane lie at or below it).  A later candidate at or above the
        // bound cannot be among the twenty (it still counts as "not listed" above).
        bound = key[kCovK - 1];
        if (kLpq == 2) { bound = min(bound, quad_dpp<0xB1>(bound)); const uint32_t m = key[kCovK / 2 - 1]; bound = min(bound, max(m, quad_dpp<0xB1>(m))); }
        if (kLpq == 4) {
            bound = min(bound, quad_dpp<0xB1>(bound)); bound = min(bound, quad_dpp<0x4E>(bound));
            uint32_t m = key[kCovK / 4 - 1];
            m = max(m, quad_dpp<0xB1>(m)); m = max(m, quad_dpp<0x4E>(m));
            bound = min(bound, m);
        }
    }
    // the lists of a query's lanes -> one (every lane of the query then holds it)
    if (kLpq >= 2) merge_with_partner<0xB1>(key);      // quad_perm [1, 0, 3, 2]
    if (kLpq >= 4) merge_with_partner<0x4E>(key);      // quad_perm [2, 3, 0, 1]
    // The listed candidates again, exactly: position in the cell-sorted array from the number in the stream (run t holds the numbers
    // [P_t, P_t + len_t): position = number + (first_t - P_t) of the last run with P_t <= number), FLANN's distance, the original index.
    // Each lane of a query looks up kPer of the twenty; the others come over by DPP.
    unsigned long long K[kCovK];
    {
        uint32_t P[kRuns], off[kRuns];
        uint32_t acc = 0;
#pragma unroll
        for (int t = 0; t < kRuns; ++t) { const uint2 r = sh.run[t][tid]; P[t] = acc; off[t] = r.x - acc; acc += r.y; }
        unsigned long long mine[kPer];
        uint32_t km[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            uint32_t k = key[i];
            if (kLpq >= 2) k = (part & 1u) ? key[kPer + i] : k;
            if (kLpq >= 4) k = (part & 2u) ? ((part & 1u) ? key[3 * kPer + i] : key[2 * kPer + i]) : k;
            km[i] = k;
        }
#pragma unroll
        for (int b = 0; b < kPer; b += 5) {
            float4 p[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const uint32_t seq = km[b + i] & idmask;
                uint32_t o = off[0];
#pragma unroll
                for (int t = 1; t < kRuns; ++t) o = seq >= P[t] ? off[t] : o;      // (the empty runs past the last real one have P_t = total > seq)
                p[i] = g.pts[km[b + i] == kKeyEmpty ? 0u : seq + o];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float d = flann_dist(q.x, q.y, q.z, p[i]);
                mine[b + i] = km[b + i] == kKeyEmpty ? ~0ull : (((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(p[i].w));
            }
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            if (kLpq == 1) K[i] = mine[i];
            if (kLpq == 2) { K[i] = quad_dpp64<0xA0>(mine[i]); K[kPer + i] = quad_dpp64<0xF5>(mine[i]); }      // quad_perm [0, 0, 2, 2] / [1, 1, 3, 3]
            if (kLpq == 4) { K[i] = quad_dpp64<0x00>(mine[i]); K[kPer + i] = quad_dpp64<0x55>(mine[i]); K[2 * kPer + i] = quad_dpp64<0xAA>(mine[i]); K[3 * kPer + i] = quad_dpp64<0xFF>(mine[i]); }
        }
    }
    // (distance, index) order: the screening order is already that but for candidates that share a screening bucket -- odd-even
    // transposition passes until no lane of the wave swaps (usually the first pass finds nothing)
    for (;;) {
        bool any_sw = false;
#pragma unroll
        for (int i = 0; i + 1 < kCovK; i += 2) COV_CSWAP(K[i], K[i + 1])
#pragma unroll
        for (int i = 1; i + 1 < kCovK; i += 2) COV_CSWAP(K[i], K[i + 1])
        if (!__any(any_sw)) break;
    }
    // Proof.  An unlisted candidate has a screening key >= key[K], hence a screened distance >= that key with its low bits cleared, hence
    // FLANN's distance >= LB = that value x (1 - 1e-6) (the two float evaluations of a sum of three squares differ by a few ulp).  The twenty
    // are the answer iff the 20th is STRICTLY nearer than LB (nothing unlisted ties with it or beats it) and than every face of the block
    // with cells behind it (nothing outside the block does).
    const bool full = key[kCovK - 1] != kKeyEmpty;
    const double d20 = (double)__uint_as_float((uint32_t)(K[kCovK - 1] >> 32));
    const double lb = key[kCovK] == kKeyEmpty ? 1e300 : (double)__uint_as_float(key[kCovK] & ~idmask) * (1.0 - 1e-6);
    const double b2 = block_bound_sq(h, q.x, q.y, q.z, cx, cy, cz, 1);
    const bool covers = b2 < 0.0;
    const bool listed_exact = !too_many && (full ? d20 < lb : true);      // (not full: every candidate of the block is listed)
#ifdef COV_DEBUG_ALLQ
    const bool ok = false;
#else
    const bool ok = listed_exact && (covers || (full && d20 < b2));
#endif
    if (active && ok) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            unsigned long long k = K[i];
            if (kLpq >= 2) k = (part & 1u) ? K[kPer + i] : k;
            if (kLpq >= 4) k = (part & 2u) ? ((part & 1u) ? K[3 * kPer + i] : K[2 * kPer + i]) : k;
            nbr[(size_t)(part * (uint32_t)kPer + (uint32_t)i) * n_cap + j] = k == ~0ull ? kNbrNone : (uint32_t)k;
        }
    }
    const bool fail = active && !ok && part == 0u;
    const unsigned long long fm = __ballot(fail);
    if (fm) {
        const int leader = __ffsll((long long)fm) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(queue_count, (uint32_t)__popcll(fm));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (fail) {
            const uint32_t slot = base + lane_rank(fm);
            // bit 31: ring 1 is known exactly and does not suffice -- the wave starts at ring 2; the seed is the 20th key of twenty REAL
            // points, an upper bound of the answer's 20th key whatever else failed
#ifdef COV_DEBUG_ALLQ
            queue[slot] = j;
            queue_seed[slot] = ~0ull;
#else
            queue[slot] = j | (listed_exact ? 0x80000000u : 0u);
            queue_seed[slot] = full ? K[kCovK - 1] : ~0ull;
#endif
        }
    }
}

// ------------------------------------------------------------------------------
// B: wave per queued query, lanes = candidates
// ------------------------------------------------------------------------------
// The lanes of ONE wave hand data to each other through LDS: its DS operations complete in order, so no barrier is needed -- but the compiler
// reasons per thread and, across a RELEASE fence, forwards a lane's own earlier store to its later load of the same address (measured: the
// position -> row table came back as the zeros each lane had cleared it with).  A sequentially consistent fence is the compiler barrier.
__device__ __forceinline__ void lds_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

struct CovWaveSh {
    uint32_t slot[64];                    // chunk position -> the row that starts there
    unsigned long long buf[2][64];        // the list (first entries, ascending) + the chunk's candidates that may enter it; double-buffered
};

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    return ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)v);
}
__device__ __forceinline__ float uniform_f32(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

__global__ __launch_bounds__(256) void cov_wave_kernel(GridView g0, GridView g1, GridView g2, int n_levels, const uint32_t* __restrict__ queue,
                                                       const unsigned long long* __restrict__ queue_seed, const uint32_t* __restrict__ queue_count,
                                                       uint32_t queue_cap, uint32_t* __restrict__ nbr, uint32_t n_cap) {
    __shared__ CovWaveSh shw[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    CovWaveSh& sh = shw[wv];
    if (g0.hdr->empty || g0.hdr->overflow || g0.hdr->stale) return;
    int nl = n_levels;
    if (n_levels > 1 && (g1.hdr->overflow || g1.hdr->empty || g1.hdr->stale)) nl = 1;
    if (n_levels > 2 && (g2.hdr->overflow || g2.hdr->empty || g2.hdr->stale)) nl = min(nl, 2);
    const uint32_t cnt = min(*queue_count, queue_cap);
    const uint32_t nw = gridDim.x * 4u;
    for (uint32_t e = blockIdx.x * 4u + (uint32_t)wv; e < cnt; e += nw) {
        const uint32_t ent = __builtin_amdgcn_readfirstlane(queue[e]);
        const uint32_t j = ent & 0x7fffffffu;
        const bool skip1 = (ent >> 31) != 0u;
        unsigned long long seed = uniform_u64(queue_seed[e]);      // keys <= seed can be among the twenty, nothing else
        const float4 qv = g0.pts[j];
        const float qx = uniform_f32(qv.x), qy = uniform_f32(qv.y), qz = uniform_f32(qv.z);
        int cur = 0;
        uint32_t n_real = 0;                 // real entries of the list (wave-uniform)
        unsigned long long thr = ~0ull;      // the 20th key once the list is full
        bool done = false;
        for (int l = 0; l < nl && !done; ++l) {
            const GridHeader* hp = l == 0 ? g0.hdr : (l == 1 ? g1.hdr : g2.hdr);
            const float4* __restrict__ pts = l == 0 ? g0.pts : (l == 1 ? g1.pts : g2.pts);
            const uint32_t* __restrict__ cs = l == 0 ? g0.cell_start : (l == 1 ? g1.cell_start : g2.cell_start);
            const GridHeader& h = *hp;
            const int d0 = h.dims[0], d1 = h.dims[1], d2 = h.dims[2];
            const double fx = floor((double)qx / h.cell - h.shift) - h.org[0], fy = floor((double)qy / h.cell - h.shift) - h.org[1],
                         fz = floor((double)qz / h.cell - h.shift) - h.org[2];
            const int cx = (int)fmin(fmax(fx, 0.0), (double)(d0 - 1)), cy = (int)fmin(fmax(fy, 0.0), (double)(d1 - 1)),
                      cz = (int)fmin(fmax(fz, 0.0), (double)(d2 - 1));
            const int rmax = max(max(max(cx, d0 - 1 - cx), max(cy, d1 - 1 - cy)), max(cz, d2 - 1 - cz));
            const int last_ring = l + 1 < nl ? 2 : 0x7fffffff;
            for (int r = (l == 0 && skip1) ? 2 : 1; ; ++r) {
                if (r > max(rmax, 1)) { done = true; break; }      // the previous block covered the whole grid
                const int z0 = max(cz - r, 0), z1 = min(cz + r, d2 - 1), y0 = max(cy - r, 0), y1 = min(cy + r, d1 - 1);
                const int x0 = max(cx - r, 0), x1 = min(cx + r, d0 - 1);
                const uint32_t ny = (uint32_t)(y1 - y0 + 1), nrows = ny * (uint32_t)(z1 - z0 + 1);
                // A new block holds the old one's points again: the list starts empty, and what it had learnt stays as the seed.
#ifndef COV_DEBUG_NOSEED
                if (n_real == (uint32_t)kCovK) seed = seed < thr ? seed : thr;
#endif
                n_real = 0; thr = ~0ull;
                const unsigned long long lim_seed = seed == ~0ull ? ~0ull : seed + 1ull;
                for (uint32_t rb = 0; rb < nrows; rb += 64u) {
                    const uint32_t row = rb + (uint32_t)lane;
                    const bool rv = row < nrows;
                    const uint32_t zz = row / ny, yy = row - zz * ny;
                    const uint32_t base = rv ? ((uint32_t)(z0 + (int)zz) * (uint32_t)d1 + (uint32_t)(y0 + (int)yy)) * (uint32_t)d0 : 0u;
                    const uint32_t rs = cs[rv ? base + (uint32_t)x0 : 0u], re = cs[rv ? base + (uint32_t)x1 + 1u : 0u];
                    const uint32_t len = rv ? re - rs : 0u;
                    uint32_t incl = len;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += t; }
                    const uint32_t excl = incl - len;
                    const uint32_t tb = __builtin_amdgcn_readfirstlane((uint32_t)__shfl((int)incl, 63, 64));
                    for (uint32_t p0 = 0; p0 < tb; p0 += 64u) {
                        // the row of every position of the chunk: rows that START inside the chunk leave their number at that position,
                        // a running maximum spreads it; the row the chunk begins in = the rows that end at or before p0
                        const uint32_t r_first = (uint32_t)__popcll(__ballot(incl <= p0));
                        sh.slot[lane] = 0u;
                        lds_wave_sync();
                        if (len > 0u && excl >= p0 && excl < p0 + 64u) sh.slot[excl - p0] = (uint32_t)lane;
                        lds_wave_sync();
                        uint32_t rid = sh.slot[lane];
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)rid, d, 64); if (lane >= d) rid = max(rid, t); }
                        rid = max(rid, r_first);
                        const uint32_t pos = p0 + (uint32_t)lane;
                        const bool cv = pos < tb;
                        const uint32_t my_rs = (uint32_t)__shfl((int)rs, (int)(rid & 63u), 64), my_excl = (uint32_t)__shfl((int)excl, (int)(rid & 63u), 64);
                        const float4 p = pts[cv ? my_rs + (pos - my_excl) : 0u];
                        const float d = flann_dist(qx, qy, qz, p);
                        const unsigned long long key = cv ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(p.w)) : ~0ull;
                        const unsigned long long lim = thr < lim_seed ? thr : lim_seed;
                        bool pass = key < lim;
                        for (;;) {
                            const unsigned long long m = __ballot(pass);
                            if (!m) break;
                            // up to (64 - n_real) of them join the list's entries in the buffer; every entry's rank = the entries below it
                            const uint32_t room = 64u - n_real, c = lane_rank(m);
                            const bool take = pass && c < room;
                            if (take) sh.buf[cur][n_real + c] = key;
                            const uint32_t nn = n_real + min((uint32_t)__popcll(m), room);
                            lds_wave_sync();
                            const unsigned long long me = (uint32_t)lane < nn ? sh.buf[cur][lane] : ~0ull;
                            uint32_t rank = 0;
#pragma unroll 4
                            for (uint32_t t = 0; t < nn; ++t) rank += sh.buf[cur][t] < me ? 1u : 0u;
                            sh.buf[cur ^ 1][lane] = ~0ull;
                            lds_wave_sync();
                            if ((uint32_t)lane < nn && rank < (uint32_t)kCovK) sh.buf[cur ^ 1][rank] = me;
                            lds_wave_sync();
                            cur ^= 1;
                            n_real = min(nn, (uint32_t)kCovK);
                            thr = n_real == (uint32_t)kCovK ? uniform_u64(sh.buf[cur][kCovK - 1]) : ~0ull;
                            pass = pass && !take && key < thr;
                        }
                    }
                }
                const double b2 = block_bound_sq(h, qx, qy, qz, cx, cy, cz, r);
                if (b2 < 0.0) { done = true; break; }
                if (n_real == (uint32_t)kCovK && (double)__uint_as_float((uint32_t)(thr >> 32)) < b2) { done = true; break; }
                if (r >= last_ring) break;
            }
        }
        if (lane < kCovK) {
            const unsigned long long k = (uint32_t)lane < n_real ? sh.buf[cur][lane] : ~0ull;
            nbr[(size_t)lane * n_cap + j] = k == ~0ull ? kNbrNone : (uint32_t)k;
        }
        lds_wave_sync();      // (the next query's first merge writes the buffers again)
    }
}

// ------------------------------------------------------------------------------
// C: covariance of every point from its neighbour list (lane per cell-sorted point)
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cov_from_nbr_kernel(GridView g, const float* __restrict__ orig, uint32_t stride, uint32_t n_sorted_max,
                                                           const uint32_t* __restrict__ nbr, uint32_t n_cap, double* __restrict__ cov6,
                                                           const int use_check, const CovCheck chk) {
    const GridHeader& h = *g.hdr;
    if (h.empty || h.overflow || h.stale) return;
    const uint32_t n = min(g.cell_start[h.n_cells], n_sorted_max);
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    uint32_t nb_idx[kCovK];
    nb_idx[0] = nbr[j];
    if (nb_idx[0] == kNbrSkip) return;
#pragma unroll
    for (int i = 1; i < kCovK; ++i) nb_idx[i] = nbr[(size_t)i * n_cap + j];
    const float4 q = g.pts[j];
    const int found = cov_from_neighbours(nb_idx, orig, stride, cov6 + (size_t)__float_as_uint(q.w) * 6);
    if (use_check) {
        // sharded target: this rank holds every map point inside [ext_lo, ext_hi) only.  The neighbourhood of a point that
        // can enter a voxel of the tile is the map's own iff its 20th neighbour is nearer than every face of that region.
        const double qd[3] = {(double)q.x, (double)q.y, (double)q.z};
        bool in = true;
        double margin = 1e300;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            in = in && qd[d] >= chk.chk_lo[d] && qd[d] < chk.chk_hi[d];
            margin = fmin(margin, fmin(qd[d] - chk.ext_lo[d], chk.ext_hi[d] - qd[d]));
        }
        if (in && margin < 1e29) {
            double r20 = 1e300;
            if (found == kCovK) {
                const float* p = orig + (size_t)nb_idx[kCovK - 1] * stride;
                r20 = sqrt((double)flann_dist(q.x, q.y, q.z, make_float4(p[0], p[1], p[2], 0.f))) * (1.0 + 1e-6);
            }
            if (!(r20 < margin)) atomicAdd(chk.violations, 1u);
        }
    }
}

}  // namespace

hipError_t CovScratch::reserve(size_t n) {
    hipError_t e;
    if ((e = nbr.reserve((n + 64) * kCovK * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = queue.reserve((n + 64) * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = seed.reserve((n + 64) * sizeof(unsigned long long))) != hipSuccess) return e;
    return count.reserve(64);
}
hipError_t CovScratch::reserve_region(size_t n, hipStream_t s) {
    hipError_t e;
    if ((e = region_list.reserve((n + 64) * sizeof(uint32_t))) != hipSuccess) return e;
    if (!region_count.p) {
        if ((e = region_count.reserve(256)) != hipSuccess) return e;
        return hipMemsetAsync(region_count.p, 0, 256, s);
    }
    return hipSuccess;
}
void CovScratch::release() { nbr.release(); queue.release(); seed.release(); count.release(); region_list.release(); region_count.release(); }

// The three kernels over one cloud (scratch reserved for >= n points by the caller, before anything was queued).
hipError_t cov_search_launch(const GridIndex& grid, const GridIndex* coarse1, const GridIndex* coarse2, const float* d_orig, size_t stride_floats,
                             size_t n, double* d_cov6, hipStream_t s, const CovCheck* check, const RoiView* roi, CovScratch& sc, hipEvent_t* ev) {
    const uint32_t blocks = (uint32_t)((n + 255) / 256 ? (n + 255) / 256 : 1);
    const int levels = coarse1 ? (coarse2 ? 3 : 2) : 1;
    CovCheck chk;
    memset(&chk, 0, sizeof chk);
    if (check) chk = *check;
    RoiView rv;
    memset(&rv, 0, sizeof rv);
    if (roi) rv = *roi;
    const uint32_t n_cap = (uint32_t)std::min(sc.queue.cap / sizeof(uint32_t), sc.nbr.cap / (kCovK * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(sc.count.p, 0, 4, s);
    if (e != hipSuccess) return e;
    // (development builds: PCR_COV_LPQ = lanes per query of the first kernel, PCR_COV_WAVE_BLOCKS = grid of the second)
    static const int lpq_small = dev_env("PCR_COV_LPQ") ? atoi(dev_env("PCR_COV_LPQ")) : 4, lpq_big = dev_env("PCR_COV_LPQ_BIG") ? atoi(dev_env("PCR_COV_LPQ_BIG")) : 1;
    const int lpq = n <= 300000 ? lpq_small : lpq_big;
    static const int wb = dev_env("PCR_COV_WAVE_BLOCKS") ? atoi(dev_env("PCR_COV_WAVE_BLOCKS")) : 2048;
    const uint32_t qblocks = (uint32_t)((n * (size_t)lpq + 255) / 256);
    // (ev: events stamped at each kernel's own begin and end -- profiling passes)
#define COV_LAUNCH_A(G, L) hipExtLaunchKernelGGL((cov_ring1_kernel<G, L>), dim3(qblocks), dim3(256), 0, s, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr, 0, grid.view(), (uint32_t)n, \
                                                 sc.nbr.as<uint32_t>(), n_cap, sc.queue.as<uint32_t>(), sc.seed.as<unsigned long long>(), sc.count.as<uint32_t>(), rv, dense_limit)
    static const uint32_t dense_limit = dev_env("PCR_COV_DENSE") ? (uint32_t)atoi(dev_env("PCR_COV_DENSE")) : 0xffffffffu;      // (development: candidates in a query's block from which it is queued at once)
    static const int grp = dev_env("PCR_COV_GROUP") ? atoi(dev_env("PCR_COV_GROUP")) : 4;
    if (lpq == 1) COV_LAUNCH_A(8, 1); else if (lpq == 2) { if (grp == 4) COV_LAUNCH_A(4, 2); else COV_LAUNCH_A(8, 2); } else { if (grp == 4) COV_LAUNCH_A(4, 4); else COV_LAUNCH_A(8, 4); }
#undef COV_LAUNCH_A
    const uint32_t wave_blocks = (uint32_t)std::max(1, wb);
    hipExtLaunchKernelGGL(cov_wave_kernel, dim3(wave_blocks), dim3(256), 0, s, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr, 0, grid.view(), coarse1 ? coarse1->view() : grid.view(),
                       coarse2 ? coarse2->view() : grid.view(), levels, sc.queue.as<uint32_t>(), sc.seed.as<unsigned long long>(), sc.count.as<uint32_t>(),
                       n_cap, sc.nbr.as<uint32_t>(), n_cap);
    hipExtLaunchKernelGGL(cov_from_nbr_kernel, dim3(blocks), dim3(256), 0, s, ev ? ev[4] : nullptr, ev ? ev[5] : nullptr, 0, grid.view(), d_orig, (uint32_t)stride_floats, (uint32_t)n, sc.nbr.as<uint32_t>(), n_cap,
                       d_cov6, check ? 1 : 0, chk);
    return hipGetLastError();
}

}  // namespace pcr
