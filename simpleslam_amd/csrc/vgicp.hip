// vgicp.hip -- voxelised GICP scan-to-map on gfx950 (hand-written HIP).
//
// Kernels for the reference's PCR::VgicpRegister::scan2Map (PCR/src/VgicpRegister.cpp:30-45),
// i.e. fast_gicp::FastVGICP under PCL's align():
//   V2 per-point covariances      fast_gicp_impl.hpp:241-297  (serial there: 20-NN over a FLANN
//      kd-tree, 4x20 f64 neighbours, cov/20, JacobiSVD, PLANE regularisation U diag(1,1,1e-3) V^T)
//      -> vgicp_cov_kernel: exact ring search on the uniform grid, float distances like FLANN
//   V3 Gaussian voxel map         pclomp/fast_vgicp_voxel.hpp:105-174 (serial unordered_map, ADDITIVE)
//      -> vgicp_voxel_kernel: one thread per voxel, fixed-point sums (order independent)
//   V4/V5 correspondences, Mahalanobis, linearize, compute_error   fast_vgicp_impl.hpp:73-204
//      -> vgicp_linearize_kernel<false> (linearize) and <true> (compute_error of an LM trial pose + the linearisation at
//         that pose in the same pass), fixed-order reductions
//   V6 LM driver                  lsq_registration_impl.hpp:53-171 -> host code in capi.hip
// Also the PCL fitness score (pcl::Registration::getFitnessScore, VgicpRegister.cpp:42-45).
#include <string.h>

#include "pcr_internal.h"
#include "peer_exchange.h"
#include "small_math.h"
#include "vgicp_opt.h"
#include "cov_math.h"

namespace pcr {

// ------------------------------------------------------------------------------
// exact K nearest neighbours by ring search; float squared distances (x,y,z order, no FMA:
// FLANN L2_Simple<float>), ties on the lower original index.  key = dist_bits << 32 | index.
// ------------------------------------------------------------------------------
template <int K>
struct KeyList {
    unsigned long long k[K];
};

template <int K, bool DEDUPE>
__device__ __forceinline__ void keylist_insert(KeyList<K>& L, unsigned long long key) {
    bool c[K];
    bool dup = false;
#pragma unroll
    for (int i = 0; i < K; ++i) { c[i] = key < L.k[i]; if (DEDUPE) dup |= key == L.k[i]; }
    if (DEDUPE && dup) return;      // a coarser level meets the points of the finer ones again
#pragma unroll
    for (int i = K - 1; i >= 1; --i) L.k[i] = c[i - 1] ? L.k[i - 1] : (c[i] ? key : L.k[i]);
    L.k[0] = c[0] ? key : L.k[0];
}

template <int K, bool DEDUPE>
__device__ __forceinline__ void ring_scan_run(const float4* __restrict__ pts, uint32_t s, uint32_t e, float qx, float qy, float qz,
                                              KeyList<K>& L) {
    // four candidates per step, their loads issued together: with one load per iteration the branchy insertion kept the
    // compiler from overlapping them, and every candidate cost a full memory round trip
    for (uint32_t j = s; j < e; j += 4) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[j + u < e ? j + u : j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
            float d = dx * dx;
            d += dy * dy;
            d += dz * dz;
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(p[u].w);
            if (j + u < e && key < L.k[K - 1]) keylist_insert<K, DEDUPE>(L, key);
        }
    }
}

// One cloud indexed at up to three cell sizes (capi.hip: cov_levels -- a scan gets two, cell and 6 cell).  A lidar scan spans four
// orders of magnitude of density; on a single grid the K-neighbourhood of a far point is dozens of rings wide and one such
// lane holds its whole wave.  Rings 1 and 2 of each level guarantee radii of one and two of its cells; only the last level
// keeps growing.
struct GridLevels {
    const GridHeader* hdr[3];
    const float4* pts[3];
    const uint32_t* cell_start[3];
    int n;
};

// Rings first..last of one level.  Returns true when the K-th distance is final (or nothing can lie beyond).
static constexpr bool kBatchRing1 = true;      // (taken only where the caller hands an LDS table over)
// rows: 9 x 256 uint2 of LDS (this block's), or nullptr (callers outside a 256-thread block layout)
template <int K, bool DEDUPE>
__device__ __forceinline__ bool ring_level(const GridHeader& h, const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start,
                                           float qx, float qy, float qz, float max_sq, int last_ring, KeyList<K>& L, uint2* rows = nullptr) {
    const int d0 = h.dims[0], d1 = h.dims[1], d2 = h.dims[2];
    double fx = floor((double)qx / h.cell - h.shift) - h.org[0], fy = floor((double)qy / h.cell - h.shift) - h.org[1],
           fz = floor((double)qz / h.cell - h.shift) - h.org[2];
    // centre cell, clamped into the grid (queries of the fitness score may lie outside)
    const int cx = (int)fmin(fmax(fx, 0.0), (double)(d0 - 1)), cy = (int)fmin(fmax(fy, 0.0), (double)(d1 - 1)),
              cz = (int)fmin(fmax(fz, 0.0), (double)(d2 - 1));
    const int rmax = max(max(max(cx, d0 - 1 - cx), max(cy, d1 - 1 - cy)), max(cz, d2 - 1 - cz));
    const float cellf = (float)h.cell;
    const double o0 = h.org[0] + h.shift, o1 = h.org[1] + h.shift, o2 = h.org[2] + h.shift;   // cell i spans [(o + i) cell, (o + i + 1) cell)
    for (int r = 1; r <= max(rmax, 1); ++r) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, d2 - 1), y0 = max(cy - r, 0), y1 = min(cy + r, d1 - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, d0 - 1);
        const float worst = __uint_as_float((uint32_t)(L.k[K - 1] >> 32));
        if (kBatchRing1 && r == 1 && rows) {
            // Ring 1 -- the nine rows of the 3 x 3 x 3 block, for most points the whole search of a level -- with ALL its row ranges
            // requested at once (18 independent loads) instead of row by row: a scan point's search is a chain of dependent round
            // trips at one wave per SIMD, and the nine pairs were nine of them.  The ranges wait in this lane's slots of the block's
            // LDS table; rows are then walked in the same order, each tested against the K-th distance as it stands.
            uint32_t ra[9], rb[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int z = cz + i / 3 - 1, y = cy + i % 3 - 1;
                const bool in = z >= 0 && z < d2 && y >= 0 && y < d1;
                const uint32_t row = in ? ((uint32_t)z * (uint32_t)d1 + (uint32_t)y) * (uint32_t)d0 : 0u;
                ra[i] = cell_start[in ? row + (uint32_t)x0 : 0u]; rb[i] = cell_start[in ? row + (uint32_t)x1 + 1u : 0u];
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) rows[i * 256 + threadIdx.x] = make_uint2(ra[i], rb[i]);
            for (int i = 0; i < 9; ++i) {
                const uint2 rg = rows[i * 256 + threadIdx.x];      // (each lane reads back what it wrote: no barrier)
                if (rg.y <= rg.x) continue;
                const int z = cz + i / 3 - 1, y = cy + i % 3 - 1;
                const float zlo = (float)((o2 + z) * h.cell), gz = fmaxf(fmaxf(zlo - qz, qz - (zlo + cellf)), 0.f) * 0.99999f;
                const float ylo = (float)((o1 + y) * h.cell), gy = fmaxf(fmaxf(ylo - qy, qy - (ylo + cellf)), 0.f) * 0.99999f;
                if (gy * gy + gz * gz > __uint_as_float((uint32_t)(L.k[K - 1] >> 32))) continue;
                ring_scan_run<K, DEDUPE>(pts, rg.x, rg.y, qx, qy, qz, L);
            }
        } else
        for (int z = z0; z <= z1; ++z) {
            // distance from the query to the slab of cells z (0 inside it); float, shaved so that it never exceeds the true gap
            const float zlo = (float)((o2 + z) * h.cell), gz = fmaxf(fmaxf(zlo - qz, qz - (zlo + cellf)), 0.f) * 0.99999f;
            if (gz * gz > worst) continue;
            // rows y0..y1 of one z layer are contiguous in key order: one subtraction tells whether the whole band
            // (all x) is empty
            if (cell_start[((uint32_t)z * (uint32_t)d1 + (uint32_t)y0) * (uint32_t)d0] ==
                cell_start[((uint32_t)z * (uint32_t)d1 + (uint32_t)y1 + 1u) * (uint32_t)d0]) continue;
            for (int y = y0; y <= y1; ++y) {
                const float ylo = (float)((o1 + y) * h.cell), gy = fmaxf(fmaxf(ylo - qy, qy - (ylo + cellf)), 0.f) * 0.99999f;
                if (gy * gy + gz * gz > worst) continue;      // the whole row is farther than the current K-th distance
                const uint32_t row = ((uint32_t)z * (uint32_t)d1 + (uint32_t)y) * (uint32_t)d0;
                const bool shell_row = r == 1 || z == cz - r || z == cz + r || y == cy - r || y == cy + r;
                if (shell_row) {
                    ring_scan_run<K, DEDUPE>(pts, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, L);
                } else {
                    if (cx - r >= 0) ring_scan_run<K, DEDUPE>(pts, cell_start[row + cx - r], cell_start[row + cx - r + 1], qx, qy, qz, L);
                    if (cx + r <= d0 - 1) ring_scan_run<K, DEDUPE>(pts, cell_start[row + cx + r], cell_start[row + cx + r + 1], qx, qy, qz, L);
                }
            }
        }
        // every point not yet visited lies beyond a face of the block [c-r, c+r]; faces on the
        // grid boundary have nothing behind them
        double bound = 1e300;
        if (cx - r > 0) bound = fmin(bound, (double)qx - (o0 + (double)(cx - r)) * h.cell);
        if (cx + r < d0 - 1) bound = fmin(bound, (o0 + (double)(cx + r + 1)) * h.cell - (double)qx);
        if (cy - r > 0) bound = fmin(bound, (double)qy - (o1 + (double)(cy - r)) * h.cell);
        if (cy + r < d1 - 1) bound = fmin(bound, (o1 + (double)(cy + r + 1)) * h.cell - (double)qy);
        if (cz - r > 0) bound = fmin(bound, (double)qz - (o2 + (double)(cz - r)) * h.cell);
        if (cz + r < d2 - 1) bound = fmin(bound, (o2 + (double)(cz + r + 1)) * h.cell - (double)qz);
        if (bound >= 1e299) return true;   // the block covers the whole grid
        const double b2 = bound > 0 ? bound * bound * (1.0 - 1e-5) : 0.0;   // margin: float distances
        if (b2 > (double)max_sq) return true;
        if ((double)__uint_as_float((uint32_t)(L.k[K - 1] >> 32)) < b2) return true;
        if (r >= last_ring) return false;
    }
    return true;
}

// Exact K nearest neighbours.  max_sq: neighbours farther than this are not needed (FLT_MAX for none).
// seed_sq: a radius^2 expected to hold at least K points (from the local density); the list starts with
// sentinels at that radius so that, in a crowded cell, the thousands of farther candidates are rejected by one
// compare instead of being inserted and displaced again.
template <int K>
__device__ __forceinline__ void ring_knn_pass(const GridLevels& lv, float qx, float qy, float qz, float max_sq, float seed_sq, KeyList<K>& L, uint2* rows) {
    const unsigned long long sentinel = ((unsigned long long)__float_as_uint(seed_sq) << 32) | 0xffffffffull;
#pragma unroll
    for (int i = 0; i < K; ++i) L.k[i] = sentinel;
    bool done = ring_level<K, false>(*lv.hdr[0], lv.pts[0], lv.cell_start[0], qx, qy, qz, max_sq, lv.n > 1 ? 2 : 0x7fffffff, L, rows);
    for (int l = 1; l < lv.n; ++l) {
        if (done) break;
        done = ring_level<K, true>(*lv.hdr[l], lv.pts[l], lv.cell_start[l], qx, qy, qz, max_sq, l + 1 < lv.n ? 2 : 0x7fffffff, L, rows);
    }
}

template <int K>
__device__ __forceinline__ void ring_knn(const GridLevels& lv, float qx, float qy, float qz, float max_sq, KeyList<K>& L, uint2* rows = nullptr) {
    const GridHeader& h = *lv.hdr[0];
    if (h.empty || h.overflow) {
#pragma unroll
        for (int i = 0; i < K; ++i) L.k[i] = ~0ull;
        return;
    }
    float seed = 3.0e38f;
    if (K > 1) {
        // points of the query's own cell, taken as a surface patch of area cell^2: radius holding ~2K of them
        const double fx = floor((double)qx / h.cell - h.shift) - h.org[0], fy = floor((double)qy / h.cell - h.shift) - h.org[1],
                     fz = floor((double)qz / h.cell - h.shift) - h.org[2];
        if (fx >= 0 && fx < h.dims[0] && fy >= 0 && fy < h.dims[1] && fz >= 0 && fz < h.dims[2]) {
            const uint32_t key = ((uint32_t)fz * (uint32_t)h.dims[1] + (uint32_t)fy) * (uint32_t)h.dims[0] + (uint32_t)fx;
            const uint32_t nc = lv.cell_start[0][key + 1] - lv.cell_start[0][key];
            if (nc >= 4u * (uint32_t)K) seed = (float)(h.cell * h.cell) * (2.0f * (float)K / (3.14159265f * (float)nc));
        }
    }
    ring_knn_pass<K>(lv, qx, qy, qz, max_sq, seed, L, rows);
    if (seed < 3.0e38f && (uint32_t)L.k[K - 1] == 0xffffffffu)      // the seed radius held fewer than K points: exact redo
        ring_knn_pass<K>(lv, qx, qy, qz, max_sq, 3.0e38f, L, rows);
#pragma unroll
    for (int i = 0; i < K; ++i) if ((uint32_t)L.k[i] == 0xffffffffu) L.k[i] = ~0ull;   // unfilled slots
}

__device__ __forceinline__ GridLevels one_level(const GridView& g) {
    GridLevels lv;
    lv.hdr[0] = lv.hdr[1] = lv.hdr[2] = g.hdr; lv.pts[0] = lv.pts[1] = lv.pts[2] = g.pts;
    lv.cell_start[0] = lv.cell_start[1] = lv.cell_start[2] = g.cell_start; lv.n = 1;
    return lv;
}

// kBatch: ring 1 of every level with its nine row ranges requested at once (ring_level).  For a SCAN-sized cloud, whose search is a chain
// of dependent round trips at one wave per SIMD: A/B on one box, the scan's 65 k covariances no longer hold the optimiser up (align 0.19 ->
// 0.11 ms).  A map-sized cloud runs the same kernel at four waves per SIMD and is bound by instruction issue: there the batched form
// costs 0.83 -> 1.28 ms per million points, so it keeps the row-by-row walk.
// A map-sized target prepared for one scan: the sorted positions of the points whose voxel the scan can reach, compacted (in blocks of 1 024
// positions, each block's share in order; one atomic per block).  The covariance kernel below then runs FULL waves over the region's ~100 k points
// instead of 16 000 thin ones over the million, most of whose lanes left at the region test while the others searched.
__global__ __launch_bounds__(256) void vgicp_region_list_kernel(GridView g, uint32_t n_sorted_max, const RoiView roi, uint32_t* __restrict__ list,
                                                                uint32_t* __restrict__ count, uint32_t* __restrict__ count_next) {
    __shared__ uint32_t sh_cnt[16], sh_base;
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_next = 0u;      // (the other of two counters, for the next call: before anything can return)
    const GridHeader h = *g.hdr;
    if (h.empty || h.overflow || h.stale) return;
    const GridHeader lat = *roi.lat;
    if (lat.stale || lat.overflow) return;
    const uint32_t n = min(g.cell_start[h.n_cells], n_sorted_max);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t c0 = blockIdx.x * 1024u; c0 < n; c0 += gridDim.x * 1024u) {
        unsigned long long m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j = c0 + (uint32_t)u * 256u + threadIdx.x;
            bool in = false;
            if (j < n) { const float4 q = g.pts[j]; in = roi_holds_point(roi, lat, (double)q.x, (double)q.y, (double)q.z); }
            m[u] = __ballot(in);
            if (lane == 0) sh_cnt[u * 4 + wave] = (uint32_t)__popcll(m[u]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int k = 0; k < 16; ++k) tot += sh_cnt[k];
            sh_base = tot ? atomicAdd(count, tot) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t off = sh_base;
            for (int k = 0; k < u * 4 + wave; ++k) off += sh_cnt[k];
            if ((m[u] >> lane) & 1ull) list[off + (uint32_t)__popcll(m[u] & ((1ull << lane) - 1ull))] = c0 + (uint32_t)u * 256u + threadIdx.x;
        }
        __syncthreads();      // sh_cnt is rewritten by the next chunk
    }
}

// list / list_count (optional): the sorted positions to process (vgicp_region_list_kernel) instead of every position with the region test
template <bool kBatch>
__global__ __launch_bounds__(256, kBatch ? 1 : 4) void vgicp_cov_kernel(GridView g, GridView g1, GridView g2, int n_levels, const float* __restrict__ orig,
                                                        uint32_t stride, uint32_t n_sorted_max, double* __restrict__ cov6, const int use_check,
                                                        const CovCheck chk, const RoiView roi, const uint32_t* __restrict__ list,
                                                        const uint32_t* __restrict__ list_count) {
    __shared__ uint2 sh_rows[kBatch ? 9 * 256 : 1];      // row ranges of ring 1 (ring_level)
    const GridHeader h = *g.hdr;
    if (h.empty || h.overflow || h.stale) return;      // (stale: queued ahead of the host's look at the header, capi.hip: settle_cov_levels; the caller builds afresh)
    GridHeader lat;
    if (roi.mask) lat = *roi.lat;
    if (roi.mask && (lat.stale || lat.overflow)) return;
    GridLevels lv;
    lv.hdr[0] = g.hdr; lv.pts[0] = g.pts; lv.cell_start[0] = g.cell_start;
    lv.hdr[1] = g1.hdr; lv.pts[1] = g1.pts; lv.cell_start[1] = g1.cell_start;
    lv.hdr[2] = g2.hdr; lv.pts[2] = g2.pts; lv.cell_start[2] = g2.cell_start;
    lv.n = n_levels;
    if (n_levels > 1 && (g1.hdr->overflow || g1.hdr->empty)) lv.n = 1;
    if (n_levels > 2 && (g2.hdr->overflow || g2.hdr->empty)) lv.n = min(lv.n, 2);
    const uint32_t n = list ? *list_count : min(g.cell_start[h.n_cells], n_sorted_max);   // points actually indexed (finite ones), or the listed ones
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t j = list ? list[i] : i;
        const float4 q = g.pts[j];
        // a target prepared for one scan: only the points whose voxel the scan can reach (the search itself always sees the whole cloud)
        if (!list && roi.mask && !roi_holds_point(roi, lat, (double)q.x, (double)q.y, (double)q.z)) continue;
        if (roi.count) atomicAdd(roi.count, 1u);      // (profiling passes only)
        KeyList<kCovK> L;
        ring_knn<kCovK>(lv, q.x, q.y, q.z, 3.0e38f, L, kBatch ? sh_rows : nullptr);
        uint32_t nb_idx[kCovK];
#pragma unroll
        for (int i = 0; i < kCovK; ++i) nb_idx[i] = L.k[i] != ~0ull ? (uint32_t)L.k[i] : 0xffffffffu;
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
                const double r20 = found == kCovK ? sqrt((double)__uint_as_float((uint32_t)(L.k[kCovK - 1] >> 32))) * (1.0 + 1e-6) : 1e300;
                if (!(r20 < margin)) atomicAdd(chk.violations, 1u);
            }
        }
    }
}

// ------------------------------------------------------------------------------
// V3: Gaussian voxel map.  Reference voxel coordinate c = floor(x/res - 0.5)
// (fast_vgicp_voxel.hpp:158-160): the target index is built on exactly that lattice
// (GridHeader.shift = 0.5), so a voxel IS a cell and its points are one contiguous run.
// One thread per sorted point; the thread of a cell's first point folds the run and stores the
// voxel at that position (no slot table, no counter).
// Sums are fixed point (2^44 per unit), hence independent of the order of the points.
// ------------------------------------------------------------------------------
static constexpr double kFix = 17592186044416.0;   // 2^44

__device__ __forceinline__ bool lattice_key(const GridHeader& h, double x, double y, double z, uint32_t* key, double c[3]) {
    c[0] = floor(x / h.cell - h.shift); c[1] = floor(y / h.cell - h.shift); c[2] = floor(z / h.cell - h.shift);
    const double vx = c[0] - h.org[0], vy = c[1] - h.org[1], vz = c[2] - h.org[2];
    if (!(vx >= 0.0 && vx < (double)h.dims[0] && vy >= 0.0 && vy < (double)h.dims[1] && vz >= 0.0 && vz < (double)h.dims[2])) return false;
    *key = ((uint32_t)vz * (uint32_t)h.dims[1] + (uint32_t)vy) * (uint32_t)h.dims[0] + (uint32_t)vx;
    return true;
}

__global__ __launch_bounds__(256) void vgicp_voxel_kernel(GridView g, const double* __restrict__ cov6, VgicpVoxel* __restrict__ vox, const RoiView roi) {
    const GridHeader h = *g.hdr;
    if (h.overflow || h.empty || h.stale) return;
    const uint32_t n = g.cell_start[h.n_cells];
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const float4 q = g.pts[j];
        uint32_t key;
        double c[3];
        if (!lattice_key(h, (double)q.x, (double)q.y, (double)q.z, &key, c)) continue;
        if (g.cell_start[key] != j) continue;                 // not the first point of its voxel
        if (roi.mask && roi.mask[roi_macro(h, roi.mshift, (int)(c[0] - h.org[0]), (int)(c[1] - h.org[1]), (int)(c[2] - h.org[2]))] == 0) continue;      // out of the scan's reach
        if (roi.count) atomicAdd(roi.count + 16, 1u);      // (profiling passes only)
        const uint32_t e = g.cell_start[key + 1];
        const double ox = (c[0] + h.shift) * h.cell, oy = (c[1] + h.shift) * h.cell, oz = (c[2] + h.shift) * h.cell;   // lower corner
        const uint32_t cnt = e - j;
        VgicpVoxel v;
        const double inv = 1.0 / (double)cnt;
        // Every term is rounded to a multiple of 2^-44 first, so the sums are exact and independent of the order.  While they provably
        // stay below 2^53 (offsets < cell, covariance entries <= 1 in magnitude after the regularisation) the integers are added
        // up as doubles -- one v_rndne_f64 and one v_add_f64 per term, where the f64 -> i64 conversion alone costs a dozen instructions.
        if ((double)cnt * (h.cell > 1.0 ? h.cell : 1.0) <= 256.0) {
            double dm[3] = {0, 0, 0}, dc[6] = {0, 0, 0, 0, 0, 0};
            for (uint32_t i = j; i < e; ++i) {
                const float4 p = g.pts[i];
                dm[0] += rint(((double)p.x - ox) * kFix); dm[1] += rint(((double)p.y - oy) * kFix); dm[2] += rint(((double)p.z - oz) * kFix);
                const double* cc = cov6 + (size_t)__float_as_uint(p.w) * 6;
#pragma unroll
                for (int k = 0; k < 6; ++k) dc[k] += rint(cc[k] * kFix);
            }
            v.mean[0] = ox + dm[0] / kFix * inv; v.mean[1] = oy + dm[1] / kFix * inv; v.mean[2] = oz + dm[2] / kFix * inv;
#pragma unroll
            for (int k = 0; k < 6; ++k) v.cov[k] = dc[k] / kFix * inv;
        } else {
            long long sm[3] = {0, 0, 0}, sc[6] = {0, 0, 0, 0, 0, 0};
            for (uint32_t i = j; i < e; ++i) {
                const float4 p = g.pts[i];
                sm[0] += llrint(((double)p.x - ox) * kFix); sm[1] += llrint(((double)p.y - oy) * kFix); sm[2] += llrint(((double)p.z - oz) * kFix);
                const double* cc = cov6 + (size_t)__float_as_uint(p.w) * 6;
#pragma unroll
                for (int k = 0; k < 6; ++k) sc[k] += llrint(cc[k] * kFix);
            }
            v.mean[0] = ox + (double)sm[0] / kFix * inv; v.mean[1] = oy + (double)sm[1] / kFix * inv; v.mean[2] = oz + (double)sm[2] / kFix * inv;
#pragma unroll
            for (int k = 0; k < 6; ++k) v.cov[k] = (double)sc[k] / kFix * inv;
        }
        v.w = sqrt((double)cnt);   // fast_vgicp_impl.hpp:149
        v.n = cnt; v.pad = 0;
        vox[j] = v;
    }
}

// ------------------------------------------------------------------------------
// V4/V5: one linearisation (update_correspondences + linearize)
// ------------------------------------------------------------------------------
__device__ __forceinline__ void inv3_sym(const double S[6], double M[6]) {
    const double a = S[0], b = S[1], c = S[2], d = S[3], e = S[4], f = S[5];
    const double A = d * f - e * e, B = c * e - b * f, Cc = b * e - c * d;
    const double det = a * A + b * B + c * Cc;
    const double id = 1.0 / det;
    M[0] = A * id; M[1] = B * id; M[2] = Cc * id;
    M[3] = (a * f - c * c) * id; M[4] = (b * c - a * e) * id; M[5] = (a * d - b * b) * id;
}

// roi: a voxel that exists but lies outside the prepared region is an ESCAPE (counted; the host prepares the whole target and repeats)
__device__ __forceinline__ uint32_t vgicp_lookup(const GridHeader& h, const uint32_t* __restrict__ cell_start, const double tp[3], const RoiView& roi) {
    if (h.overflow || h.empty) return 0;
    uint32_t key;
    double c[3];
    if (!lattice_key(h, tp[0], tp[1], tp[2], &key, c)) return 0;
    // (an index of the region's points only -- roi.filtered -- holds nothing outside the mask, not even the cell table's entries: the mask is tested
    //  FIRST, and a lookup outside it is an escape whether or not a voxel is there -- as NDT treats its region-only index)
    const bool outside = roi.mask && roi.mask[roi_macro(h, roi.mshift, (int)(c[0] - h.org[0]), (int)(c[1] - h.org[1]), (int)(c[2] - h.org[2]))] == 0;
    if (outside && roi.filtered) { atomicAdd(roi.escapes, 1u); return 0; }
    const uint32_t s = cell_start[key], e = cell_start[key + 1];
    if (e > s && outside) {
        atomicAdd(roi.escapes, 1u);
        return 0;
    }
    return e > s ? s + 1 : 0;
}

static constexpr int kLinStride = 258;

// one source point of update_correspondences + linearize at pose T: v[0..20] H (upper triangle), v[21..26] b, v[27] error;
// the correspondence (voxel slot, Mahalanobis matrix) goes to slot_out / M_out
__device__ __forceinline__ void vgicp_lin_point(const VgicpArgs& a, const GridHeader& h, const Pose16& T, uint32_t i, double v[28],
                                                uint32_t* __restrict__ slot_out, double* __restrict__ M_out) {
    const float* sp = a.src + (size_t)i * a.src_stride;
    const double p[3] = {(double)sp[0], (double)sp[1], (double)sp[2]};
    double tp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) tp[r] = T.m[r] * p[0] + T.m[4 + r] * p[1] + T.m[8 + r] * p[2] + T.m[12 + r] * 1.0;
    uint32_t slot = vgicp_lookup(h, a.cell_start, tp, a.roi);
    if (a.escapes && h.clamped) {
        // The index covers only the bulk of the target (a stray point made its box too large for dense tables).  A source point that
        // lands near -- or beyond -- a face behind which target points were left out would meet voxels whose covariances lack
        // neighbours, or miss a voxel altogether: counted, and the host fails the call instead of returning a pose of the cut map.
        const double c[3] = {floor(tp[0] / h.cell - h.shift) - h.org[0], floor(tp[1] / h.cell - h.shift) - h.org[1], floor(tp[2] / h.cell - h.shift) - h.org[2]};
        const double g = (double)(kPad + a.guard_cells);
        bool esc = false;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            esc = esc || ((h.cut_mask & (1 << d)) && c[d] < g) || ((h.cut_mask & (8 << d)) && c[d] >= (double)h.dims[d] - g);
        if (esc) atomicAdd(a.escapes, 1u);
    }
    if (a.use_tile && !(tp[0] >= a.tile_lo[0] && tp[0] < a.tile_hi[0] && tp[1] >= a.tile_lo[1] && tp[1] < a.tile_hi[1] &&
                        tp[2] >= a.tile_lo[2] && tp[2] < a.tile_hi[2])) slot = 0;      // sharded target: another rank's query
    slot_out[i] = slot;
    if (!slot) return;
    const VgicpVoxel vx = a.vox[slot - 1];
    const double* ca = a.src_cov6 + (size_t)i * 6;
    const double CA[3][3] = {{ca[0], ca[1], ca[2]}, {ca[1], ca[3], ca[4]}, {ca[2], ca[4], ca[5]}};
    double RC[3][3], S[6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) RC[r][c] = T.m[r] * CA[0][c] + T.m[4 + r] * CA[1][c] + T.m[8 + r] * CA[2][c];
    int o = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = r; c < 3; ++c) { S[o] = vx.cov[o] + (RC[r][0] * T.m[c] + RC[r][1] * T.m[4 + c] + RC[r][2] * T.m[8 + c]); ++o; }
    double M6[6];
    inv3_sym(S, M6);   // (C_B + T C_A T^T)^-1, fast_vgicp_impl.hpp:104-115
#pragma unroll
    for (int k = 0; k < 6; ++k) M_out[(size_t)i * 6 + k] = M6[k];
    const double M[3][3] = {{M6[0], M6[1], M6[2]}, {M6[1], M6[3], M6[4]}, {M6[2], M6[4], M6[5]}};
    const double er[3] = {vx.mean[0] - tp[0], vx.mean[1] - tp[1], vx.mean[2] - tp[2]};
    double Me[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) Me[r] = M[r][0] * er[0] + M[r][1] * er[1] + M[r][2] * er[2];
    const double w = vx.w;
    // J = [skew(Tp) | -I]   fast_vgicp_impl.hpp:156-158, so3.hpp:21-31
    const double J[3][6] = {{0, -tp[2], tp[1], -1, 0, 0}, {tp[2], 0, -tp[0], 0, -1, 0}, {-tp[1], tp[0], 0, 0, 0, -1}};
    double MJ[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) MJ[r][c] = M[r][0] * J[0][c] + M[r][1] * J[1][c] + M[r][2] * J[2][c];
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) v[q++] = w * (J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c]);
#pragma unroll
    for (int r = 0; r < 6; ++r) v[21 + r] = w * (J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2]);
    v[27] = w * (er[0] * Me[0] + er[1] * Me[1] + er[2] * Me[2]);
}

// compute_error (fast_vgicp_impl.hpp:183-204) of one point: the correspondences and Mahalanobis matrices of the LAST
// linearisation, new pose
__device__ __forceinline__ double vgicp_err_point(const VgicpArgs& a, const Pose16& T, uint32_t i) {
    const uint32_t slot = a.corr_slot[i];
    if (!slot) return 0.0;
    const float* sp = a.src + (size_t)i * a.src_stride;
    const double p[3] = {(double)sp[0], (double)sp[1], (double)sp[2]};
    const VgicpVoxel vx = a.vox[slot - 1];
    double er[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) er[r] = vx.mean[r] - (T.m[r] * p[0] + T.m[4 + r] * p[1] + T.m[8 + r] * p[2] + T.m[12 + r] * 1.0);
    const double* M6 = a.corr_M + (size_t)i * 6;
    const double Me0 = M6[0] * er[0] + M6[1] * er[1] + M6[2] * er[2], Me1 = M6[1] * er[0] + M6[3] * er[1] + M6[4] * er[2],
                 Me2 = M6[2] * er[0] + M6[4] * er[1] + M6[5] * er[2];
    return vx.w * (er[0] * Me0 + er[1] * Me1 + er[2] * Me2);
}

// kWithError = false: linearize(T) -- correspondences to a.corr_slot / a.corr_M, partial sums [0..27].
// kWithError = true: one pass for an LM trial pose T: [28] = compute_error(T) on the correspondences of the last linearisation
// (a.corr_slot / a.corr_M, read only) AND, speculatively, the linearisation AT T (correspondences to a.corr_slot_next /
// a.corr_M_next, sums [0..27]).  When the trial is accepted -- the usual case -- T is the next linearisation point and the
// host swaps the buffers instead of paying another launch and round trip; when it is rejected the sums are dropped.
// Both sums go through the same fixed-order block reduction as a stand-alone linearisation: bit-identical values.
template <bool kWithError>
__device__ __forceinline__ void vgicp_lin_body(const VgicpArgs& a, const Pose16& T, double* sh /* [29][kLinStride] */, double* sh_sum /* [8][32] */) {
    constexpr int kRows = kWithError ? 29 : 28;
    const GridHeader h = *a.hdr;
    const int tid = threadIdx.x, e = tid & 31, ch = tid >> 5;
    double acc = 0.0;
    for (uint32_t base = blockIdx.x * 256; base < a.n_src; base += gridDim.x * 256) {
        const uint32_t i = base + tid;
        double v[28];
#pragma unroll
        for (int k = 0; k < 28; ++k) v[k] = 0.0;
        double err = 0.0;
        if (i < a.n_src) {
            if (kWithError) {
                err = vgicp_err_point(a, T, i);
                vgicp_lin_point(a, h, T, i, v, a.corr_slot_next, a.corr_M_next);
            } else {
                vgicp_lin_point(a, h, T, i, v, a.corr_slot, a.corr_M);
            }
        }
#pragma unroll
        for (int k = 0; k < 28; ++k) sh[k * kLinStride + tid] = v[k];
        if (kWithError) sh[28 * kLinStride + tid] = err;
        __syncthreads();
        if (e < kRows) {
            const double* row = sh + e * kLinStride + ch * 32;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) acc += row[k];
        }
        __syncthreads();
    }
    sh_sum[ch * 32 + e] = e < kRows ? acc : 0.0;
    __syncthreads();
    if (tid < 32) {
        double s = sh_sum[tid];
#pragma unroll
        for (int c = 1; c < 8; ++c) s += sh_sum[c * 32 + tid];
        a.partials[(size_t)blockIdx.x * 32 + tid] = s;
    }
}

template <bool kWithError>
__global__ __launch_bounds__(256) void vgicp_linearize_kernel(const VgicpArgs a, const Pose16 T) {
    __shared__ double sh[(kWithError ? 29 : 28) * kLinStride];
    __shared__ double sh_sum[8 * 32];
    vgicp_lin_body<kWithError>(a, T, sh, sh_sum);
}

// ------------------------------------------------------------------------------
// Device-resident Levenberg-Marquardt loop (unsharded targets): one launch per pass.  The prologue of a launch folds the 29 sums
// of the previous one and takes the optimiser's step (vgicp_opt.h: vg_ctl_step) -- in every block, the same instructions on the
// same numbers, as ndt_pass_pro_kernel and loam_iterate_kernel do; block 0 writes the new state and the progress word.  State and
// rows are double-buffered by launch parity; the two correspondence buffers are chosen by the state's own parity (an accepted
// trial makes the buffer it wrote the current one).  The loop ends in the prologue of the launch after its last pass.
// ------------------------------------------------------------------------------
struct VgProArgs {
    const double* rows_prev;     // [rows_prev_n][32]
    const VgCtl* ctl_prev;
    VgCtl* ctl_next;
    VgOut* out;
    double seq;
    uint32_t rows_prev_n;
    int32_t first;
    const double* reduced;       // sharded over the peer exchange: the previous launch's 32 sums, already folded over the rows AND the ranks (vgicp_peer_exchange_kernel)
};
static constexpr int kVgCtlWords = (int)((sizeof(VgCtl) + 3) / 4);
static_assert(sizeof(VgCtl) % 4 == 0, "VgCtl is copied word by word");

__global__ __launch_bounds__(256) void vgicp_pass_pro_kernel(const VgicpArgs a_in, const VgProArgs pa) {
    __shared__ double sh[29 * kLinStride];
    __shared__ double sh_sum[8 * 32];
    __shared__ __attribute__((aligned(16))) uint32_t sh_ctl[kVgCtlWords];
    __shared__ double sh_sums[32];
    const unsigned long long t_in = wall_clock64();
    const int t = threadIdx.x;
    VgCtl* const c = reinterpret_cast<VgCtl*>(sh_ctl);
    // one round trip: the state and the rows of the previous launch ([8 slices][32 components], 32 rows a thread for <= 256 rows)
    const int comp = t & 31, slice = t >> 5;
    double acc = 0.0;
    if (pa.reduced) {      // (block-uniform) the sums arrive folded: slice 0 carries them, the others zeros
        if (!pa.first && slice == 0) acc = pa.reduced[comp];
        for (int w = t; w < kVgCtlWords; w += 256) sh_ctl[w] = reinterpret_cast<const uint32_t*>(pa.ctl_prev)[w];
    } else {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const uint32_t row = (uint32_t)(slice + 8 * u);
            v[u] = (!pa.first && row < pa.rows_prev_n) ? pa.rows_prev[(size_t)row * 32 + comp] : 0.0;
        }
        for (int w = t; w < kVgCtlWords; w += 256) sh_ctl[w] = reinterpret_cast<const uint32_t*>(pa.ctl_prev)[w];
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += v[u];
    }
    if (!pa.first && !pa.reduced)
        for (uint32_t r0 = 256; r0 < pa.rows_prev_n; r0 += 256) {      // (more than 65 536 source points: 512 rows)
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const uint32_t row = r0 + (uint32_t)(slice + 8 * u);
                v[u] = row < pa.rows_prev_n ? pa.rows_prev[(size_t)row * 32 + comp] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += v[u];
        }
    sh_sum[slice * 32 + comp] = acc;
    __syncthreads();
    if (c->done) {      // finished in an earlier launch: hand the state on to whatever is queued behind
        if (blockIdx.x == 0) for (int w = t; w < kVgCtlWords; w += 256) reinterpret_cast<uint32_t*>(pa.ctl_next)[w] = sh_ctl[w];
        return;
    }
    if (!pa.first) {
        if (t < 32) {
            double s = sh_sum[t];
#pragma unroll
            for (int k = 1; k < 8; ++k) s += sh_sum[k * 32 + t];
            sh_sums[t] = s;
        }
        __syncthreads();
        if (t == 0) {
            const unsigned long long t_a = wall_clock64();
            vg_opt::ctl_step(c, sh_sums);
            c->ticks[0] += (uint32_t)(t_a - t_in); c->ticks[1] += (uint32_t)(wall_clock64() - t_a);
        }
        __syncthreads();
        if (blockIdx.x == 0) {
            for (int w = t; w < kVgCtlWords; w += 256) reinterpret_cast<uint32_t*>(pa.ctl_next)[w] = sh_ctl[w];
            if (t == 0) {
                VgOut* const out = pa.out;
                if (c->done) {
                    out->x0 = c->x0;
                    out->conv = c->conv; out->outer = c->outer; out->n_lin = c->n_lin; out->n_err = c->n_err; out->passes = c->passes;
                    // (every pass ran in an earlier launch of this stream: the count is final)
                    out->roi_escapes = a_in.roi.mask ? (int32_t)min(*a_in.roi.escapes, 0x7fffffffu) : 0;
                    __threadfence_system();
                    __hip_atomic_store(&out->seq, pa.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    __hip_atomic_store(&out->progress, pa.seq * kProgressWindow + (double)c->passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        if (c->done) return;
    }
    // the pose of this pass and the correspondence buffers, as scalars
    Pose16 T;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double v = c->xi.m[i];
        T.m[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
    const int kind = __builtin_amdgcn_readfirstlane(c->kind), parity = __builtin_amdgcn_readfirstlane(c->parity);
    VgicpArgs a = a_in;
    if (parity) {
        a.corr_slot = a_in.corr_slot_next; a.corr_M = a_in.corr_M_next;
        a.corr_slot_next = a_in.corr_slot; a.corr_M_next = a_in.corr_M;
    }
    __syncthreads();      // (sh_sum is reused by the body)
    if (kind == kVgPassLinearize) vgicp_lin_body<false>(a, T, sh, sh_sum);
    else vgicp_lin_body<true>(a, T, sh, sh_sum);
}

// Sharded targets over the peer exchange (pcr_comm_init_peer): between two passes ONE launch folds this rank's rows of the pass that has just run --
// the order of vgicp_pass_pro_kernel's own fold -- pushes the 32 sums into every peer's receive buffer and folds what arrived in rank order
// (peer_exchange.h); the next pass's prologue takes the result (VgProArgs::reduced) and the optimiser's step as ever, in every block, on every rank,
// on the same bits.  Two launches per pass and no host round trip, where the host-driven loop of a sharded target has one per pass.  Once the loop
// has finished (the state the pass before left says so, on every rank in the same launch) nothing is exchanged.
__global__ __launch_bounds__(256) void vgicp_peer_exchange_kernel(const double* __restrict__ rows, uint32_t n_rows, const VgCtl* __restrict__ ctl, const PeerComm pc,
                                                                  const double xseq, double* __restrict__ reduced) {
    __shared__ double sh_sum[8 * 32];
    const int t = threadIdx.x, comp = t & 31, slice = t >> 5;
    if (ctl->done) return;
    double acc = 0.0;
    for (uint32_t r0 = 0; r0 < n_rows; r0 += 256) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const uint32_t row = r0 + (uint32_t)(slice + 8 * u);
            v[u] = row < n_rows ? rows[(size_t)row * 32 + comp] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += v[u];
    }
    sh_sum[slice * 32 + comp] = acc;
    __syncthreads();
    double mine = 0.0;
    if (t < 32) {
        mine = sh_sum[t];
#pragma unroll
        for (int k = 1; k < 8; ++k) mine += sh_sum[k * 32 + t];
    }
    peer_exchange_block(pc, xseq, nullptr, mine, 32, 0, reduced);      // (a rank that never arrives: the status word says so to the host, which ends the call and the session)
}

struct VgCtlArg { uint32_t w[kVgCtlWords]; };
__global__ __launch_bounds__(256) void vgicp_ctl_store_kernel(VgCtl* __restrict__ ctl, const VgCtlArg init, uint32_t* __restrict__ roi_escapes) {
    for (int t = threadIdx.x; t < kVgCtlWords; t += 256) reinterpret_cast<uint32_t*>(ctl)[t] = init.w[t];
    if (roi_escapes && threadIdx.x == 0) *roi_escapes = 0u;
}

// fold per-block partials (32 doubles each) into 32 doubles, fixed order
// out lives in host-mapped memory: out[31] receives `seq` LAST (system-scope release), the word the host spins on
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partials, uint32_t nblocks, double* __restrict__ out, double seq) {
    __shared__ double sh[8 * 32];
    const int t = threadIdx.x, comp = t & 31, slice = t >> 5;
    double acc = 0.0;
    // eight loads in flight per step, added in the order of a plain loop (same sums, a quarter of the round trips)
    for (uint32_t b0 = slice; b0 < nblocks; b0 += 64) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint32_t b = b0 + 8u * u; v[u] = b < nblocks ? partials[(size_t)b * 32 + comp] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (b0 + 8u * u < nblocks) acc += v[u];
    }
    sh[slice * 32 + comp] = acc;
    __syncthreads();
    if (t < 31) {
        double v = sh[t];
#pragma unroll
        for (int s = 1; s < 8; ++s) v += sh[s * 32 + t];
        out[t] = v;
        __threadfence_system();
    }
    __syncthreads();
    if (t == 0) __hip_atomic_store(&out[31], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------
// PCL fitness score: mean squared 1-NN distance of the transformed source
// (pcl::transformPointCloud in float, float kd-tree distances)
// ------------------------------------------------------------------------------
struct PoseF16 { float m[16]; };

__global__ __launch_bounds__(256) void fitness_kernel(GridView g, const float* __restrict__ src, uint32_t n_src, uint32_t stride,
                                                      const PoseF16 T, float max_range, double* __restrict__ partials, const FitTile tile) {
    __shared__ double sh[256];
    __shared__ double shc[256];
    __shared__ double shv[256];
    double acc = 0.0, cnt = 0.0, viol = 0.0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_src; i += gridDim.x * 256) {
        const float* p = src + (size_t)i * stride;
        const float qx = T.m[0] * p[0] + T.m[4] * p[1] + T.m[8] * p[2] + T.m[12];
        const float qy = T.m[1] * p[0] + T.m[5] * p[1] + T.m[9] * p[2] + T.m[13];
        const float qz = T.m[2] * p[0] + T.m[6] * p[1] + T.m[10] * p[2] + T.m[14];
        double margin = 1e300;
        if (tile.use) {         // sharded target: the rank whose tile holds the transformed point scores it
            const double qd[3] = {(double)qx, (double)qy, (double)qz};
            bool in = true;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                in = in && qd[d] >= tile.lo[d] && qd[d] < tile.hi[d];
                margin = fmin(margin, fmin(qd[d] - tile.ext_lo[d], tile.ext_hi[d] - qd[d]));
            }
            if (!in) continue;
        }
        KeyList<1> L;
        ring_knn<1>(one_level(g), qx, qy, qz, max_range, L);
        float d = 3.0e38f;
        if (L.k[0] != ~0ull) {
            d = __uint_as_float((uint32_t)(L.k[0] >> 32));
            if (d <= max_range) { acc += (double)d; cnt += 1.0; }
        }
        // the nearest point of the rank's cloud is the map's nearest only if it is nearer than the faces of the region the
        // cloud is complete in (or than the gate: farther points do not count anyway)
        if (tile.use && margin < 1e29 && !(sqrt((double)d) * (1.0 + 1e-6) < margin) && !(sqrt((double)max_range) < margin)) viol += 1.0;
    }
    sh[threadIdx.x] = acc; shc[threadIdx.x] = cnt; shv[threadIdx.x] = viol;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[threadIdx.x] += sh[threadIdx.x + s]; shc[threadIdx.x] += shc[threadIdx.x + s]; shv[threadIdx.x] += shv[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * 32] = sh[0]; partials[(size_t)blockIdx.x * 32 + 1] = shc[0]; partials[(size_t)blockIdx.x * 32 + 2] = shv[0];
        for (int k = 3; k < 32; ++k) partials[(size_t)blockIdx.x * 32 + k] = 0.0;
    }
}

// ---- host launchers ---------------------------------------------------------------
hipError_t vgicp_launch_cov(const GridIndex& grid, const GridIndex* coarse1, const GridIndex* coarse2, const float* d_orig, size_t stride_floats,
                            size_t n, double* d_cov6, hipStream_t s, const CovCheck* check, const RoiView* roi, CovScratch* scratch, hipEvent_t* ev) {
    static const bool old_kernel = dev_env("PCR_COV_OLD") != nullptr;      // (development builds: A/B against the lane-per-query kernel)
    static const bool all_sizes = dev_env("PCR_COV_NEW_ALL") != nullptr;      // (development builds: map-sized clouds through cov_search.hip too)
    if (scratch && (n <= 300000 || all_sizes) && n > 0 && !old_kernel) {      // scan-sized: two classes of queries (cov_search.hip)
        const hipError_t e = scratch->reserve(n);
        return e != hipSuccess ? e : cov_search_launch(grid, coarse1, coarse2, d_orig, stride_floats, n, d_cov6, s, check, roi, *scratch, ev);
    }
    const int blocks = (int)std::min<size_t>(65535, (n + 255) / 256 ? (n + 255) / 256 : 1);
    const int levels = coarse1 ? (coarse2 ? 3 : 2) : 1;
    CovCheck chk;
    memset(&chk, 0, sizeof chk);
    if (check) chk = *check;
    RoiView rv;
    memset(&rv, 0, sizeof rv);
    if (roi) rv = *roi;
    // (ev: events stamped at the kernel's own begin and end -- profiling passes)
    // a map-sized target prepared for one scan: the region's points are listed first, and the search runs over the list
    const uint32_t* d_list = nullptr;
    const uint32_t* d_list_count = nullptr;
    int cov_blocks = blocks;
    static const bool no_list = dev_env("PCR_COV_NO_LIST") != nullptr;      // (development builds: A/B)
    if (n > 300000 && roi && roi->mask && scratch && !check && !no_list) {
        const hipError_t e = scratch->reserve_region(n, s);
        if (e != hipSuccess) return e;
        scratch->region_idx ^= 1;
        uint32_t* const cnt = scratch->region_count.as<uint32_t>() + 32 * scratch->region_idx;
        uint32_t* const cnt_next = scratch->region_count.as<uint32_t>() + 32 * (scratch->region_idx ^ 1);
        const int lb = (int)std::min<size_t>(2048, (n + 1023) / 1024);
        hipLaunchKernelGGL(vgicp_region_list_kernel, dim3(lb), dim3(256), 0, s, grid.view(), (uint32_t)n, rv, scratch->region_list.as<uint32_t>(), cnt, cnt_next);
        d_list = scratch->region_list.as<uint32_t>(); d_list_count = cnt;
        cov_blocks = std::min(blocks, 1024);      // (four blocks per CU resident at once; the kernel strides over the list)
    }
#define PCR_COV_ARGS grid.view(), coarse1 ? coarse1->view() : grid.view(), coarse2 ? coarse2->view() : grid.view(), levels, d_orig, (uint32_t)stride_floats, (uint32_t)n, d_cov6, check ? 1 : 0, chk, rv, d_list, d_list_count
    if (n <= 300000) {      // scan-sized (the same threshold as the choice of search levels, capi.hip: cov_levels)
        if (ev) hipExtLaunchKernelGGL(vgicp_cov_kernel<true>, dim3(blocks), dim3(256), 0, s, ev[0], ev[1], 0, PCR_COV_ARGS);
        else hipLaunchKernelGGL(vgicp_cov_kernel<true>, dim3(blocks), dim3(256), 0, s, PCR_COV_ARGS);
    } else {
        if (ev) hipExtLaunchKernelGGL(vgicp_cov_kernel<false>, dim3(cov_blocks), dim3(256), 0, s, ev[0], ev[1], 0, PCR_COV_ARGS);
        else hipLaunchKernelGGL(vgicp_cov_kernel<false>, dim3(cov_blocks), dim3(256), 0, s, PCR_COV_ARGS);
    }
#undef PCR_COV_ARGS
    return hipGetLastError();
}

hipError_t vgicp_launch_voxels(const GridIndex& grid, const double* d_cov6, VgicpVoxel* d_vox, hipStream_t s, const RoiView* roi) {
    const size_t n = grid.n_points;
    const int blocks = (int)std::min<size_t>(65535, (n + 255) / 256 ? (n + 255) / 256 : 1);
    RoiView rv;
    memset(&rv, 0, sizeof rv);
    if (roi) rv = *roi;
    hipLaunchKernelGGL(vgicp_voxel_kernel, dim3(blocks), dim3(256), 0, s, grid.view(), d_cov6, d_vox, rv);
    return hipGetLastError();
}

uint32_t vgicp_blocks(uint32_t n_src) {
    uint32_t b = (n_src + 255) / 256;
    return b < 1 ? 1 : (b > 512 ? 512 : b);
}

hipError_t vgicp_launch_linearize(const VgicpArgs& a, const Pose16& T, double* d_out32, hipStream_t s, double seq) {
    const uint32_t nb = vgicp_blocks(a.n_src);
    hipLaunchKernelGGL(vgicp_linearize_kernel<false>, dim3(nb), dim3(256), 0, s, a, T);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, a.partials, nb, d_out32, seq);
    return hipGetLastError();
}

hipError_t vgicp_launch_ctl_init(VgCtl* d_ctl2, const Pose16& guess, int max_iters, int lm_inner, double lm_init_scale, double rot_eps, double trans_eps, hipStream_t s,
                                 uint32_t* d_roi_escapes) {
    VgCtl c;
    memset(&c, 0, sizeof c);
    vg_opt::ctl_init(&c, guess, max_iters, lm_inner, lm_init_scale, rot_eps, trans_eps);
    VgCtlArg arg;
    memcpy(arg.w, &c, sizeof c);
    hipLaunchKernelGGL(vgicp_ctl_store_kernel, dim3(1), dim3(256), 0, s, d_ctl2, arg, d_roi_escapes);
    return hipGetLastError();
}
// launch `index` of the device-resident loop: d_ctl2 = two VgCtl, d_rows2 = two buffers of 512 * 32 doubles
// pc / xseq / d_reduced (sharded over the peer exchange, launches after the first): the exchange launch in front of the pass
hipError_t vgicp_launch_pass_pro(const VgicpArgs& a_in, VgCtl* d_ctl2, double* d_rows2, VgOut* d_out, hipStream_t s, double seq, int index,
                                 const PeerComm* pc, double xseq, double* d_reduced) {
    const uint32_t nb = vgicp_blocks(a_in.n_src);
    VgicpArgs a = a_in;
    a.partials = d_rows2 + (size_t)(index & 1) * 512 * 32;
    VgProArgs pa;
    pa.rows_prev = d_rows2 + (size_t)((index + 1) & 1) * 512 * 32;
    pa.ctl_prev = index == 0 ? d_ctl2 : d_ctl2 + ((index + 1) & 1);
    pa.ctl_next = d_ctl2 + (index & 1);
    pa.out = d_out; pa.seq = seq; pa.rows_prev_n = nb; pa.first = index == 0 ? 1 : 0;
    pa.reduced = pc ? d_reduced : nullptr;
    if (pc && index > 0) hipLaunchKernelGGL(vgicp_peer_exchange_kernel, dim3(1), dim3(256), 0, s, pa.rows_prev, nb, pa.ctl_prev, *pc, xseq, d_reduced);
    hipLaunchKernelGGL(vgicp_pass_pro_kernel, dim3(nb), dim3(256), 0, s, a, pa);
    return hipGetLastError();
}

hipError_t vgicp_launch_error(const VgicpArgs& a, const Pose16& T, double* d_out32, hipStream_t s, double seq) {
    const uint32_t nb = vgicp_blocks(a.n_src);
    hipLaunchKernelGGL(vgicp_linearize_kernel<true>, dim3(nb), dim3(256), 0, s, a, T);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, a.partials, nb, d_out32, seq);
    return hipGetLastError();
}

hipError_t fitness_launch(const GridIndex& grid, const float* d_src, size_t n_src, size_t stride_floats, const double pose[16], double max_range,
                          double* d_partials, double* d_out32, hipStream_t s, double seq, const FitTile* tile) {
    PoseF16 T;
    for (int i = 0; i < 16; ++i) T.m[i] = (float)pose[i];
    const uint32_t nb = vgicp_blocks((uint32_t)n_src);
    const float mr = max_range >= 3.0e38 ? 3.0e38f : (float)max_range;
    FitTile ft;
    memset(&ft, 0, sizeof ft);
    if (tile) ft = *tile;
    hipLaunchKernelGGL(fitness_kernel, dim3(nb), dim3(256), 0, s, grid.view(), d_src, (uint32_t)n_src, (uint32_t)stride_floats, T, mr, d_partials, ft);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, d_partials, nb, d_out32, seq);
    return hipGetLastError();
}

}  // namespace pcr
