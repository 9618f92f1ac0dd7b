// grid_index.hip -- uniform-grid spatial index over the target cloud (gfx950).
//
// Replaces the reference's per-call kd-tree build, Kdtree::setInputCloud ->
// nanoflann buildIndex (reference PCR/src/LoamRegister.cpp:110,
// third_parties/nanoflann/include/nanoflann/nanoflann.hpp:1542-1564): a serial
// O(N log N) recursive split there, five launches here:
//   bbox partials (+ header, by the last block to finish) -> histogram with ranks -> exclusive scan of the cell
//   counters in two launches (the first also re-zeroes the counters for the next build) -> scatter.
// Algorithmic traffic: 16 B read + 16 B written per target point.
// Everything is enqueued on one stream with no host synchronisation; the grid
// geometry lives in a device-side GridHeader that the later kernels read.
#include "pcr_internal.h"
#include <stdlib.h>
#include <algorithm>
#include <string.h>

namespace pcr {

static constexpr int kScanPerThread = 8;
static constexpr int kScanBlock = 256;
static constexpr int kScanTile = kScanPerThread * kScanBlock;  // 2048 cells per block

// ---- wave64 helpers --------------------------------------------------------
__device__ inline float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// ---- 1. per-block bounding boxes; the last block to finish turns them into the header ----------------
// (one launch instead of two; the cell counters are NOT cleared here: every build leaves them zeroed, see the scan)
__global__ __launch_bounds__(256) void grid_bbox_header_kernel(const float* __restrict__ pts, uint32_t n, uint32_t stride,
                                                               float* __restrict__ partials, uint32_t* __restrict__ ticket,
                                                               GridHeader* __restrict__ hdr, uint64_t capacity, double cell, double shift, int pcl_mode,
                                                               const ClampBox clamp, int margin_xy, GridHeader* __restrict__ mirror, const HeaderTwin twin, int margin_z_pcl) {
    // mirror: a host-mapped copy of the header (or nullptr), written here so that no copy has to be queued behind the build when nothing
    // later in it can change the header (a build without hints).  twin: a second index over the SAME cloud at another cell size gets its
    // header from the same box (the coarse level of a scan's covariance search: one pass over the cloud instead of two).
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    // eight independent loads in flight per lane (a 1 M-point cloud is 16-32 MB: this pass should run at HBM speed)
    constexpr int kU = 8;
    const uint32_t step = gridDim.x * 256;
    for (uint32_t i0 = blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += kU * step) {
        float v[kU][3];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + u * step;
            const float* p = pts + (size_t)(i < n ? i : i0) * stride;      // out-of-range slots repeat a valid point
            v[u][0] = p[0]; v[u][1] = p[1]; v[u][2] = p[2];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const float x = v[u][0], y = v[u][1], z = v[u][2];
            if (isfinite(x) && isfinite(y) && isfinite(z)) {
                mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
                mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
                mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
            }
        }
    }
    __shared__ float sh[4][6];
    __shared__ uint32_t sh_last;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float a = wave_min(mn[d]), b = wave_max(mx[d]);
        if (lane == 0) { sh[wave][d] = a; sh[wave][3 + d] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sh[w][threadIdx.x]) : fmaxf(v, sh[w][threadIdx.x]);
        __hip_atomic_store(&partials[blockIdx.x * 6 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        // The six stores above come from this wave and go to device scope (sc1, past this XCD's L2), as do the loads of the
        // folding block: waiting for their acknowledgement orders them before the ticket.  (A release fence here writes
        // back and invalidates the whole L2 of the XCD -- once per block.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!sh_last) return;
    // ---- last block: fold the partial boxes (read past this XCD's L2) and write the header ----
    float fmn[3] = {INFINITY, INFINITY, INFINITY}, fmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += 256) {
        float t[6];      // all six loads in flight before the first use
#pragma unroll
        for (int d = 0; d < 6; ++d) t[d] = __hip_atomic_load(&partials[b * 6 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int d = 0; d < 3; ++d) { fmn[d] = fminf(fmn[d], t[d]); fmx[d] = fmaxf(fmx[d], t[3 + d]); }
    }
    __syncthreads();     // sh is reused
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float a = wave_min(fmn[d]), b = wave_max(fmx[d]);
        if (lane == 0) { sh[wave][d] = a; sh[wave][3 + d] = b; }
    }
    __syncthreads();
    if (threadIdx.x == 0) *ticket = 0;                         // ready for the next build
    if (threadIdx.x < 2 && (threadIdx.x == 0 || twin.hdr)) {  // (lane 0: this index's header, lane 1: the twin's)
        const bool second = threadIdx.x == 1;
        if (second) { cell = twin.cell; capacity = twin.capacity; hdr = twin.hdr; mirror = twin.mirror; }
        GridHeader h;
        h.cell = cell; h.inv_cell = 1.0 / cell; h.n_points = n; h.shift = shift;
        h.empty = 0; h.overflow = 0;
        h.pcl_mode = pcl_mode; h.inv_leaf_f = 1.0f / (float)cell; h.too_fine = 0; h.sum_sq = 0.f; h.sum_sq_u = 0ull;
        h.min_b[0] = h.min_b[1] = h.min_b[2] = 0;
        h.clamped = (clamp.use && !pcl_mode) ? 1 : 0; h.cut_mask = 0; h.stale = 0; h.pad2_ = 0;
        double nc = 1.0, nc_tight = 1.0;      // (nc_tight: cells of the box WITHOUT the room left for the next cloud -- pcl's own count)
        for (int d = 0; d < 3; ++d) {
            float lo = fminf(fminf(sh[0][d], sh[1][d]), fminf(sh[2][d], sh[3][d]));
            float hi = fmaxf(fmaxf(sh[0][3 + d], sh[1][3 + d]), fmaxf(sh[2][3 + d], sh[3][3 + d]));
            if (!(lo <= hi)) { h.empty = 1; lo = hi = 0.f; }
            if (h.clamped) {      // keep the part of the box inside the region of interest (points outside get no key)
                if (lo < (float)clamp.lo[d]) h.cut_mask |= 1 << d;          // target points lie beyond this face: a query next to it
                if (hi > (float)clamp.hi[d]) h.cut_mask |= 8 << d;          // would miss neighbours (loam.hip counts such queries)
                lo = fmaxf(lo, (float)clamp.lo[d]); hi = fminf(hi, (float)clamp.hi[d]);
                if (!(lo <= hi)) { h.empty = 1; lo = hi = 0.f; }
            }
            if (pcl_mode) {
                // pcl::VoxelGrid::applyFilter: min_b = floor(min_p * inverse_leaf_size), float arithmetic throughout
                float flo = floorf(lo * h.inv_leaf_f), fhi = floorf(hi * h.inv_leaf_f);
                nc_tight *= (double)fhi - (double)flo + 1.0;
                // (after a hint has failed: voxel membership does not depend on where the lattice starts, and neither does the ORDER of the voxels -- idx sorts
                //  by (z, y, x) whatever the box: NDT pads x and y, the voxel filter z as well)
                const int mg = d < 2 ? margin_xy : margin_z_pcl;
                if (!h.empty && mg) { flo -= (float)mg; fhi += (float)mg; }
                const double dim = (double)fhi - (double)flo + 1.0;
                h.min_b[d] = fabsf(flo) < 2.0e9f ? (int32_t)flo : 0;
                h.org[d] = (double)flo; h.origin[d] = (double)flo * cell;
                h.dims[d] = dim < 2.0e9 ? (int32_t)dim : 0x7fffffff;
                nc *= dim;
                continue;
            }
            double clo = floor((double)lo / cell - shift), chi = floor((double)hi / cell - shift);
            if (!h.empty) {      // room for the next cloud (see GridIndex::hint_ok): a sub-map grows sideways, a scan's box breathes on every axis
                const int mg = d < 2 ? margin_xy : (margin_xy < 2 ? margin_xy : 2);
                clo -= mg; chi += mg;
            }
            h.org[d] = clo - kPad;
            h.origin[d] = (clo - kPad + shift) * cell;
            double dim = chi - clo + 1.0 + 2.0 * kPad;
            h.dims[d] = dim < 2.0e9 ? (int32_t)dim : 0x7fffffff;
            nc *= dim;
        }
        if (pcl_mode && nc_tight > 2147483647.0) h.too_fine = 1;      // (dx*dy*dz) > INT_MAX
        // keys are uint32 and the table holds n_cells + 1 starts
        if (h.too_fine) { h.overflow = 0; h.empty = 1; h.n_cells = 1; }      // nothing is indexed; the caller copies its input
        else if (nc + 1.0 > (double)capacity || nc > 4.0e9) { h.overflow = 1; h.n_cells = nc < 1.8e19 ? (uint64_t)nc : ~0ull; }
        else h.n_cells = (uint64_t)h.dims[0] * (uint64_t)h.dims[1] * (uint64_t)h.dims[2];
        *hdr = h;
        if (mirror) *mirror = h;
    }
}

// (cxyz: the cell's coordinates in the lattice as well, for a caller that looks the cell up in a region mask)
__device__ inline bool point_key(const GridHeader& h, float x, float y, float z, uint32_t* key, bool* outside = nullptr, uint32_t* cxyz = nullptr) {
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) return false;
    if (h.pcl_mode) {
        // ijk = static_cast<int>(std::floor(p * inverse_leaf_size) - static_cast<float>(min_b))   (voxel_grid.hpp)
        const int ix = (int)(floorf(x * h.inv_leaf_f) - (float)h.min_b[0]), iy = (int)(floorf(y * h.inv_leaf_f) - (float)h.min_b[1]),
                  iz = (int)(floorf(z * h.inv_leaf_f) - (float)h.min_b[2]);
        if ((uint32_t)ix >= (uint32_t)h.dims[0] || (uint32_t)iy >= (uint32_t)h.dims[1] || (uint32_t)iz >= (uint32_t)h.dims[2]) {
            if (outside) *outside = true;       // only possible with a box taken over from the previous build
            return false;
        }
        *key = ((uint32_t)iz * (uint32_t)h.dims[1] + (uint32_t)iy) * (uint32_t)h.dims[0] + (uint32_t)ix;
        if (cxyz) { cxyz[0] = (uint32_t)ix; cxyz[1] = (uint32_t)iy; cxyz[2] = (uint32_t)iz; }
        return true;
    }
    // cell index = floor(x / cell) - org.  For a power-of-two cell (LOAM) x / cell is exact and this
    // equals floor((x - origin) / cell); for any other edge (VGICP/NDT resolutions) it is the single
    // definition every kernel uses, so a point and its queries always agree on the cell.
    double fx, fy, fz;
    if (h.inv_cell * h.cell == 1.0 && (__double_as_longlong(h.cell) & 0x000fffffffffffffll) == 0) {
        // a power-of-two edge (every LOAM index): x * (1 / cell) IS x / cell, without three double divisions per point
        fx = floor((double)x * h.inv_cell - h.shift) - h.org[0];
        fy = floor((double)y * h.inv_cell - h.shift) - h.org[1];
        fz = floor((double)z * h.inv_cell - h.shift) - h.org[2];
    } else {
        fx = floor((double)x / h.cell - h.shift) - h.org[0];
        fy = floor((double)y / h.cell - h.shift) - h.org[1];
        fz = floor((double)z / h.cell - h.shift) - h.org[2];
    }
    if (!(fx >= (double)kPad && fx < (double)(h.dims[0] - kPad) && fy >= (double)kPad && fy < (double)(h.dims[1] - kPad) &&
          fz >= (double)kPad && fz < (double)(h.dims[2] - kPad))) {
        // outside the region of interest of a clamped index -- or outside a box taken over from the previous build
        if (outside) *outside = true;
        return false;
    }
    const uint32_t cx = (uint32_t)fx, cy = (uint32_t)fy, cz = (uint32_t)fz;
    *key = (cz * (uint32_t)h.dims[1] + cy) * (uint32_t)h.dims[0] + cx;
    if (cxyz) { cxyz[0] = cx; cxyz[1] = cy; cxyz[2] = cz; }
    return true;
}

// ---- 3. histogram + rank of every point inside its cell -------------------------------
// One atomic per RUN of equal keys in consecutive lanes (clouds that come out of a voxel filter or a
// lidar driver are spatially ordered, so neighbouring lanes often share a cell): the run's first lane adds the
// run length and hands the base to the others.  key/rank are kept so that the scatter pass needs neither the
// f64 key arithmetic nor atomics again.
__global__ __launch_bounds__(256) void grid_count_kernel(const float* __restrict__ pts, uint32_t n, uint32_t stride,
                                                         const GridHeader* __restrict__ hdr, uint32_t* __restrict__ cell_count,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ ranks) {
    const GridHeader h = *hdr;
    if (h.overflow || h.empty) return;
    const int lane = threadIdx.x & 63;
    const uint32_t n_round = (n + 255u) & ~255u;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_round; i += gridDim.x * 256) {
        uint32_t key = 0xffffffffu;
        if (i < n) {
            const float* p = pts + (size_t)i * stride;
            uint32_t k;
            if (point_key(h, p[0], p[1], p[2], &k)) key = k;
        }
        const uint32_t prev = __shfl_up(key, 1, 64);
        const bool leader = lane == 0 || prev != key;
        const unsigned long long lead = __ballot(leader);
        // my run starts at the highest leader bit at or below my lane and ends before the next leader bit
        const unsigned long long below = lead & (~0ull >> (63 - lane));
        const int start = 63 - __clzll(below);
        const unsigned long long above = lane == 63 ? 0ull : (lead >> (lane + 1)) << (lane + 1);
        const int end = above ? __ffsll((long long)above) - 1 : 64;
        uint32_t base = 0;
        if (leader && key != 0xffffffffu) base = atomicAdd(&cell_count[key], (uint32_t)(end - start));
        base = __shfl(base, start, 64);
        if (i < n) { keys[i] = key; ranks[i] = base + (uint32_t)(lane - start); }
    }
}

// ---- 4. exclusive scan of the counters (two launches) -----------------------------------------
// Tile = 2048 cells per block: thread t owns cells 4t..4t+3 of each 1024-cell half (one 16-byte access per half,
// fully coalesced).  Launch A: tile-local exclusive scan -> cell_start, tile total -> block_sums, and the counters
// are written back as ZERO, which is the state the next build's histogram expects (no separate clear pass).
// Launch B: every block sums the totals of the tiles before it (a few thousand at most) and adds that offset.
template <int kThreads>
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* total, uint32_t* sh /* >= kThreads / 64 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) { const uint32_t x = sh[w]; if (w < wave) off += x; tot += x; }
    *total = tot;
    return off + inc - v;
}
__device__ inline uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* total, uint32_t* sh /* >= 4 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) off += sh[w]; tot += sh[w]; }
    *total = tot;
    return off + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void grid_scan_local_kernel(uint32_t* __restrict__ cell_count,
                                                                    uint32_t* __restrict__ cell_start,
                                                                    uint32_t* __restrict__ block_sums,
                                                                    const GridHeader* __restrict__ hdr) {
    __shared__ uint32_t sh[8];
    if (hdr->overflow) return;
    const uint64_t total = hdr->n_cells + 1;
    const uint64_t tile = (uint64_t)blockIdx.x * kScanTile;
    if (tile >= total) return;
    const uint64_t i0 = tile + (uint64_t)threadIdx.x * 4, i1 = i0 + kScanTile / 2;
    uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
    // (the tables are padded to a whole tile, so the 16-byte accesses never leave the allocation)
    if (i0 < total) { a = *reinterpret_cast<const uint4*>(cell_count + i0); *reinterpret_cast<uint4*>(cell_count + i0) = make_uint4(0, 0, 0, 0); }
    if (i1 < total) { b = *reinterpret_cast<const uint4*>(cell_count + i1); *reinterpret_cast<uint4*>(cell_count + i1) = make_uint4(0, 0, 0, 0); }
    uint32_t ta, tb;
    const uint32_t oa = block_exclusive_scan_256(a.x + a.y + a.z + a.w, &ta, sh);
    const uint32_t ob = block_exclusive_scan_256(b.x + b.y + b.z + b.w, &tb, sh + 4) + ta;
    if (i0 < total) *reinterpret_cast<uint4*>(cell_start + i0) = make_uint4(oa, oa + a.x, oa + a.x + a.y, oa + a.x + a.y + a.z);
    if (i1 < total) *reinterpret_cast<uint4*>(cell_start + i1) = make_uint4(ob, ob + b.x, ob + b.x + b.y, ob + b.x + b.y + b.z);
    // sum of count^2 of the tile (a density estimate, float is plenty): second half of block_sums, as float bits
    float sq = (float)a.x * (float)a.x + (float)a.y * (float)a.y + (float)a.z * (float)a.z + (float)a.w * (float)a.w +
               (float)b.x * (float)b.x + (float)b.y * (float)b.y + (float)b.z * (float)b.z + (float)b.w * (float)b.w;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
    __shared__ float sh_sq[4];
    if ((threadIdx.x & 63) == 0) sh_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = ta + tb;
        block_sums[gridDim.x + blockIdx.x] = __float_as_uint(sh_sq[0] + sh_sq[1] + sh_sq[2] + sh_sq[3]);
    }
}

__global__ __launch_bounds__(kScanBlock) void grid_scan_add_kernel(uint32_t* __restrict__ cell_start,
                                                                  const uint32_t* __restrict__ block_sums,
                                                                  GridHeader* __restrict__ hdr) {
    __shared__ uint32_t sh[4];
    __shared__ float shf[4];
    if (hdr->overflow) return;
    const uint64_t total = hdr->n_cells + 1;
    const uint64_t tile = (uint64_t)blockIdx.x * kScanTile;
    if (tile >= total) return;
    if (blockIdx.x == 0) {                       // density estimate: sum of the tiles' count^2
        const uint32_t tiles = (uint32_t)((total + kScanTile - 1) / kScanTile);
        float sq = 0.f;
        for (uint32_t t = threadIdx.x; t < tiles; t += kScanBlock) sq += __uint_as_float(block_sums[gridDim.x + t]);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
        if ((threadIdx.x & 63) == 0) shf[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) hdr->sum_sq = shf[0] + shf[1] + shf[2] + shf[3];
    }
    uint32_t part = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kScanBlock) part += block_sums[t];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
    __syncthreads();
    const uint32_t add = sh[0] + sh[1] + sh[2] + sh[3];
    const uint64_t i0 = tile + (uint64_t)threadIdx.x * 4, i1 = i0 + kScanTile / 2;
    if (i0 < total) { uint4 v = *reinterpret_cast<uint4*>(cell_start + i0); v.x += add; v.y += add; v.z += add; v.w += add; *reinterpret_cast<uint4*>(cell_start + i0) = v; }
    if (i1 < total) { uint4 v = *reinterpret_cast<uint4*>(cell_start + i1); v.x += add; v.y += add; v.z += add; v.w += add; *reinterpret_cast<uint4*>(cell_start + i1) = v; }
}

// ---- 5. scatter into cell order (no atomics: position = cell start + rank) ---------------
__global__ __launch_bounds__(256) void grid_scatter_kernel(const float* __restrict__ pts, uint32_t n, uint32_t stride,
                                                           const GridHeader* __restrict__ hdr, const uint32_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ cell_start,
                                                           float4* __restrict__ sorted) {
    if (hdr->overflow || hdr->empty) return;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t key = keys[i];
        if (key != 0xffffffffu) {
            const float* p = pts + (size_t)i * stride;
            // the order inside a cell is immaterial (the searches break distance ties on the original index in .w)
            sorted[cell_start[key] + ranks[i]] = make_float4(p[0], p[1], p[2], __uint_as_float(i));
        }
    }
}


// ======================================================================================================================
// Tiled build path (cell tables up to kMaxBins << kMaxTileShift cells): a two-level counting sort with NO global atomic
// per point.  The one-level path above spends half its time in a million returning atomics on scattered counters, which
// this multi-XCD part executes at the memory side at ~25 G/s whatever their scope.  Here:
//   bin    (grid_bin_kernel)    points -> tile = key >> shift (<= kMaxBins tiles of 2^shift consecutive cells); a block
//                               histograms a chunk of 2048 points at a time in LDS (one LDS atomic per run of equal tiles in
//                               consecutive lanes; it returns the rank inside the chunk) and claims room in each tile the chunk
//                               touched with ONE global atomic (counters 64 bytes apart: memory-side atomics on one line
//                               serialise).  The last block to finish scans the counters into bin_start.
//   place  (grid_place_kernel)  moves each point to its slot of its tile: the cloud grouped by tile, 16 B read + 16 B written
//                               per point.
//   tile   (grid_tile_kernel)   one block per tile: histogram of the tile's cells in LDS, exclusive scan -> cell_start of
//                               those cells (every cell of the table is written exactly once: no scan pass over the table, no
//                               clearing), points to their final position.  16 B read + 16 B written per point.
// The order of the points inside a cell depends on the arrival order of LDS atomics, as it depended on the order of the
// global ones before: nothing downstream may depend on it, and nothing does (searches break distance ties on the
// original index in .w; voxel statistics are fixed-point sums).
// ======================================================================================================================
#ifdef PCR_DEV_SWITCHES
// development builds: s_memrealtime stamps of the bin and tile kernels ([kernel 0/1][block < 8192][8]), read back by pcr_dev_read_stamps
__device__ unsigned long long* g_dev_stamps = nullptr;
#define DEV_STAMP(kernel, slot) do { if (g_dev_stamps && threadIdx.x == 0 && blockIdx.x < 8192) g_dev_stamps[((size_t)(kernel) * 8192 + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define DEV_STAMP(kernel, slot) do { } while (0)
#endif
static constexpr int kBinStride = 16;                  // counters 64 bytes apart: memory-side atomics on one line serialise
// (Measured, round 5: the same counters 16 to a line -- bin b at word (b & 15) * 512 + (b >> 4), so that the last block reads them back in 16 coalesced
//  rows instead of one line per counter at ~2.3 ns a line -- shorten that block's tail from 3.6 to 3 us at 1 M points and make the claims come back after
//  41 us instead of 16: a chunk of a map in generator order touches every ground tile, ~300 claims per counter, and sixteen counters' claims then queue up
//  on one line.  What bounds the claim phase is the serial handling of the claims on its hottest LINE.)
static constexpr int kBinPerDefault = 8;              // points per thread and chunk of the bin kernel

// ---- BINS: the units of the two passes.  A build without a layout hint bins by TILE (2^shift consecutive cells: bin b = tile b).  A ground tile of a
// 1 m grid over a 0.5 m map holds eight times the points of the median tile, and the tile pass lasted as long as its heaviest tile's ONE block (24 us of
// chunked two-pass work against a median block of 5 us at 1 M points; 151-165 us against 11.5 at 10 M: profiles/r04_notes.md).  So the LAYOUT a build
// leaves for the next one (one block of its tile pass plans it beside the others: plan_next_layout) cuts the tiles that were heavy into 2^k equal
// SLABS of cells -- k per tile, one step up or down per build -- and the bin pass of the next build, which places by that layout anyway, takes a
// point's bin from a per-tile word in LDS: base + (key >> (shift - k)) & (2^k - 1).  A bin is a run of whole cells in cell order, so the sorted cloud
// and the cell table are the same whatever the cut.  Layout buffer (uint32 words; two, alternating):
//   [0 .. nb]              where bin b's points go in `tiled` (room: what the bin held + an eighth + 32; the children of a tile cut one step further
//                          get their parent's room each, a tile merged one step its children's sum)   [nb] = total room
//   [kLaySub + t]          tile t: first bin | k << 13
//   [kLayMap + b]          bin b: tile | slab << 13 | k << 26
//   [kLayMeta]             nb (0: no usable layout), [kLayMeta + 1] tiles of the header the layout was made for
static constexpr uint32_t kLaySub = kMaxBins + 8, kLayMap = 2 * kMaxBins + 16, kLayMeta = 3 * kMaxBins + 24, kLayWords = 3 * kMaxBins + 32;
static constexpr int kMaxSplit = 6;                    // at most 64 slabs per tile (and never fewer than 4 cells per slab)

template <int kThreads>
__device__ inline unsigned long long block_exclusive_scan_u64(unsigned long long v, unsigned long long* total, unsigned long long* sh /* >= kThreads / 64 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    unsigned long long off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) { const unsigned long long x = sh[w]; if (w < wave) off += x; tot += x; }
    *total = tot;
    return off + inc - v;
}

template <bool kVec>
__device__ __forceinline__ void load_xyz(const float* __restrict__ pts, size_t i, uint32_t stride, float& x, float& y, float& z) {
    if (kVec) { const float4 v = *reinterpret_cast<const float4*>(pts + i * stride); x = v.x; y = v.y; z = v.z; }
    else { const float* p = pts + i * stride; x = p[0]; y = p[1]; z = p[2]; }
}

// dynamic LDS: nb_max counters (nb_max = bins this build can have: host-known; without a layout = max_tiles, the tiles the cell table's
// capacity can make) + (kPlace) as many + 1 places + (kSub) the tiles' words, 16 bits each
// kPlace: the build has a LAYOUT HINT -- lay_cur[b] = where bin b's points go in `tiled`, with room for an eighth more than the bin held in the
// previous build (+32), planned by the previous build's tile pass -- and moves every point to its bin right here:
// the separate placing pass (a second read of the cloud, 15 us at 1 M points) disappears.  A sub-map changes by a key frame at a
// time, so the room nearly always suffices; a bin that outgrows it raises header.stale, nothing is stored out of bounds (every store is
// checked against the room and the buffer), and the caller rebuilds without hints (the same protocol as the bounding-box hint that this path
// requires anyway).  kSub: the layout may hold tiles that are cut (see BINS): a point's bin comes from its tile's word; without it bin = tile, the
// kernel of round 4 (the words cost the 10 M-point map's bin pass its third block per CU, and clouds of that size are never cut).
// kPacked (region-only builds, see grid_keep_kernel below): `pts` is the list that pass left -- float4 (x, y, z, original index), *n_packed of them, all
// inside the box and the region -- so this pass pays its chunk protocol (four barriers and a returning claim per 2 048 points) for the points that are
// kept only.
template <bool kVec, int kBinPer, bool kPlace, bool kSub, bool kPacked = false>
__global__ __launch_bounds__(256) void grid_bin_kernel(const float* __restrict__ pts, uint32_t n_in, uint32_t stride, GridHeader* __restrict__ hdr,
                                                       uint32_t* __restrict__ bin_count, uint32_t* __restrict__ slot, int shift, uint32_t max_tiles, uint32_t nb_max,
                                                       uint32_t* __restrict__ ticket, uint32_t* __restrict__ bin_start,
                                                       const uint32_t* __restrict__ lay_cur, float4* __restrict__ tiled, uint32_t tiled_cap,
                                                       const uint8_t* __restrict__ keep_mask, int keep_mshift, uint32_t* __restrict__ n_packed = nullptr) {
    static_assert(!kPacked || (kVec && kPlace), "a packed list is float4 and is placed by a layout");
    // keep_mask (kPlace only; pcr_internal.h: BuildFilter): points in cells whose macro cell is not marked are left out of the index, as
    // non-finite points are; the layout is handed on unchanged (its rooms are the full cloud's)
    static_assert(kPlace || !kSub, "tiles are cut by a layout only");
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    uint32_t* const hist = dyn_lds;
    uint32_t* const sh_lay = dyn_lds + nb_max;      // kPlace: nb_max + 1 entries
    uint16_t* const sh_sub = reinterpret_cast<uint16_t*>(dyn_lds + 2 * (size_t)nb_max + 2);      // kSub: max_tiles entries (first bin | k << 13)
    DEV_STAMP(0, 0);
    constexpr uint32_t kBinChunk = 256u * kBinPer;      // points a block histograms at a time
    // (requesting a block's first chunk of points before the header has arrived -- two independent round trips -- was measured, round 5: nothing at
    //  1 M points, 6-7 us WORSE at 10 M, where blocks queue up three to a CU)
    const uint32_t c_first = blockIdx.x * kBinChunk;
    uint32_t nb_lay = 0, nt_lay = 0;
    if (kPlace) { nb_lay = lay_cur[kLayMeta]; nt_lay = lay_cur[kLayMeta + 1]; }      // (requested beside the header)
    uint32_t n = n_in;
    if (kPacked) n = min(*n_packed, n_in);
    const GridHeader h = *hdr;
    if (h.overflow || h.empty) { if (kPacked && blockIdx.x == 0 && threadIdx.x == 0) *n_packed = 0u; return; }      // (the counter is left at zero: the state the next build's keep pass expects)
    const uint32_t ntiles = (uint32_t)(h.n_cells >> shift) + 1u;
    const uint32_t nb = kPlace ? nb_lay : ntiles;
    if (kPlace) {
        // (a layout made for another lattice, one that holds cuts this kernel does not read, or none: nothing is binned, the caller rebuilds without hints)
        if (nb == 0u || nb > nb_max || nt_lay != ntiles || ntiles > max_tiles || (!kSub && nb != ntiles)) { if (threadIdx.x == 0) hdr->stale = 3; return; }      // (stale: 1 = a point outside the box, 2 = a bin outgrew its room, 3 = no usable layout; callers test != 0)
        for (uint32_t b = threadIdx.x; b <= nb; b += 256) sh_lay[b] = lay_cur[b];
        if (kSub) for (uint32_t t = threadIdx.x; t < ntiles; t += 256) sh_sub[t] = (uint16_t)lay_cur[kLaySub + t];
    }
    for (uint32_t b = threadIdx.x; b < nb_max; b += 256) hist[b] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (uint32_t c0 = c_first; c0 < n; c0 += gridDim.x * kBinChunk) {
        uint32_t bin[kBinPer], loc[kBinPer];
        bool first[kBinPer];
        float px[kBinPer], py[kBinPer], pz[kBinPer];
        uint32_t pw[kPacked ? kBinPer : 1];      // kPacked: the original index the point carries
#pragma unroll
        for (int u = 0; u < kBinPer; ++u) {      // all loads of the chunk in flight
            const uint32_t i = c0 + u * 256 + threadIdx.x;
            if (kPacked) { const float4 v = reinterpret_cast<const float4*>(pts)[i < n ? i : c0]; px[u] = v.x; py[u] = v.y; pz[u] = v.z; pw[u] = __float_as_uint(v.w); }
            else load_xyz<kVec>(pts, i < n ? i : c0, stride, px[u], py[u], pz[u]);
        }
#pragma unroll
        for (int u = 0; u < kBinPer; ++u) {
            const uint32_t i = c0 + u * 256 + threadIdx.x;
            uint32_t key;
            bin[u] = 0xffffffffu; loc[u] = 0u;
            bool outside = false;
            if (i < n && point_key(h, px[u], py[u], pz[u], &key, &outside) && (!kPlace || !keep_mask || roi_mask_holds_cell(h, keep_mask, keep_mshift, key))) {
                bin[u] = key >> shift;
                if (kSub) {      // the tile's slabs (see BINS above)
                    const uint32_t e = sh_sub[bin[u]], k = e >> 13;
                    bin[u] = (e & 0x1fffu) + ((key >> (shift - (int)k)) & ((1u << k) - 1u));
                }
            }
            if (outside && !h.clamped) hdr->stale = 1;      // (only a box reused from the previous build can be too small)
            // one LDS atomic per RUN of equal bins in consecutive lanes (a cloud stored in a spatially coherent order puts
            // whole waves into one bin: 64 same-address atomics would serialise)
            const uint32_t prev = __shfl_up(bin[u], 1, 64);
            const bool leader = lane == 0 || prev != bin[u];
            const unsigned long long lead = __ballot(leader);
            const unsigned long long below = lead & (~0ull >> (63 - lane));
            const int start = 63 - __clzll(below);
            const unsigned long long above = lane == 63 ? 0ull : (lead >> (lane + 1)) << (lane + 1);
            const int end = above ? __ffsll((long long)above) - 1 : 64;
            uint32_t base = 0;
            if (leader && bin[u] != 0xffffffffu) base = atomicAdd(&hist[bin[u]], (uint32_t)(end - start));
            first[u] = leader && bin[u] != 0xffffffffu && base == 0u;      // this lane opened the bin in this chunk
            base = __shfl(base, start, 64);
            loc[u] = base + (uint32_t)(lane - start);
        }
        DEV_STAMP(0, 1);
        __syncthreads();
        DEV_STAMP(0, 2);
        // the lane that opened a bin claims room for all the chunk's points of that bin: one global atomic per (chunk, bin)
        uint32_t cnt[kBinPer], got[kBinPer];
#pragma unroll
        for (int u = 0; u < kBinPer; ++u) cnt[u] = first[u] ? hist[bin[u]] : 0u;
#pragma unroll
        for (int u = 0; u < kBinPer; ++u) got[u] = first[u] ? atomicAdd(&bin_count[(size_t)bin[u] * kBinStride], cnt[u]) : 0u;      // (all in flight together)
        __syncthreads();      // every count has been read
#pragma unroll
        for (int u = 0; u < kBinPer; ++u) if (first[u]) hist[bin[u]] = got[u];
        __syncthreads();
        DEV_STAMP(0, 3);
        if (kPlace) {
            bool over = false;
#pragma unroll
            for (int u = 0; u < kBinPer; ++u) {
                const uint32_t i = c0 + u * 256 + threadIdx.x;
                if (i < n && bin[u] != 0xffffffffu) {
                    const uint32_t r = hist[bin[u]] + loc[u], lo = sh_lay[bin[u]], room = sh_lay[bin[u] + 1u] - lo;
                    if (r < room && lo + r < tiled_cap) tiled[lo + r] = make_float4(px[u], py[u], pz[u], __uint_as_float(kPacked ? pw[u] : i));
                    else over = true;
                }
            }
            if (over) hdr->stale = 2;      // the bin has outgrown the room the previous build left it
        } else {
#pragma unroll
            for (int u = 0; u < kBinPer; ++u) {
                const uint32_t i = c0 + u * 256 + threadIdx.x;
                if (i < n) slot[i] = bin[u] != 0xffffffffu ? hist[bin[u]] + loc[u] : 0xffffffffu;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kBinPer; ++u) if (first[u]) hist[bin[u]] = 0u;      // ready for the next chunk
        __syncthreads();
    }
    DEV_STAMP(0, 4);
    // ---- the last block to finish turns the bin counters (64 bytes apart, written by device-scope atomics) into the compact
    //      exclusive scan the next two kernels read: bin_start[0 .. nb]  (the layout of the NEXT build is planned by a block of the tile
    //      pass, beside the others: tile_block0 -- here it was another block scan on the one chain every block of the next kernel waits for).
    //      Measured and not kept, round 5 (profiles/r05_notes.md): the scan by block 0 of the tile pass, published to the other blocks as tagged words
    //      they poll (its 8 K counter loads queue behind the 16 MB the other blocks ask for at the same moment: published after 12 us); a second, compact
    //      set of counters added up by atomics nobody waits for, summed by every block of the tile pass for itself (sixteen counters to a line: the
    //      claims beside them came back after 25 us instead of 16) ----
    __shared__ uint32_t sh_last, sh4[4];
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's claims have been acknowledged (they returned values)
        sh_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    DEV_STAMP(0, 5);
    if (!sh_last) return;
    const uint32_t per = (nb + 255u) / 256u;      // <= kMaxBins / 256 = 32
    const uint32_t b0 = threadIdx.x * per;
    uint32_t c[kMaxBins / 256];
    // all of a thread's counters requested before the first is used: UNCONDITIONAL loads of clamped indices, as many as the (block-uniform) length of a
    // thread's run rounded up to four (a load under a per-element condition waits for the one before it: 0.7 us apiece, measured in round 5 -- and a
    // counter too many is a line too many: they come back at ~2.3 ns a line)
#pragma unroll
    for (uint32_t j = 0; j < kMaxBins / 256; ++j) c[j] = 0u;
    switch ((per + 3u) >> 2) {
#define PCR_LOAD_COUNTERS(N) case (N) / 4: { _Pragma("unroll") for (uint32_t j = 0; j < (N); ++j) c[j] = __hip_atomic_load(&bin_count[(size_t)min(b0 + j, (uint32_t)kMaxBins) * kBinStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } break;
        PCR_LOAD_COUNTERS(4) PCR_LOAD_COUNTERS(8) PCR_LOAD_COUNTERS(12) PCR_LOAD_COUNTERS(16) PCR_LOAD_COUNTERS(20) PCR_LOAD_COUNTERS(24) PCR_LOAD_COUNTERS(28)
#undef PCR_LOAD_COUNTERS
        default: {
#pragma unroll
            for (uint32_t j = 0; j < kMaxBins / 256; ++j) c[j] = __hip_atomic_load(&bin_count[(size_t)min(b0 + j, (uint32_t)kMaxBins) * kBinStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } break;
    }
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < kMaxBins / 256; ++j) { if (!(j < per && b0 + j < nb)) c[j] = 0u; sum += c[j]; }
    uint32_t total;
    uint32_t off = block_exclusive_scan_256(sum, &total, sh4);
#pragma unroll
    for (uint32_t j = 0; j < kMaxBins / 256; ++j)
        if (j < per && b0 + j < nb) { bin_start[b0 + j] = off; off += c[j]; }
    if (threadIdx.x == 255) bin_start[nb] = total;
    if (threadIdx.x == 0) *ticket = 0u;                        // ready for the next build
    if (kPacked && threadIdx.x == 0) *n_packed = 0u;           // (every block has read it: they all took the ticket after their loops)
    DEV_STAMP(0, 6);
}

// ---- region-only builds: the points of the region, listed ----
// A build that indexes a scan's region only (BuildFilter: NDT's 5 M-point map, VGICP's lattice) drops nine points in ten -- and the bin pass charged every
// chunk of 2 048 points its whole protocol all the same: 60 us for the 5 M-point map, 80 MB read at 1.3 TB/s (chunk sizes of 1 024 and 4 096: slower).
// This pass only streams: eight points per thread, cell, mask byte, a block scan of the kept counts, ONE global claim per block and chunk, the kept points
// stored as (x, y, z, original index).  The bin pass then reads that list (grid_bin_kernel<.., kPacked>).  A point outside the reused box raises
// header.stale here, as the bin pass did.  The order of the list is the order the claims arrive in -- as the order of a bin's points already was.
template <bool kVec>
__global__ __launch_bounds__(256) void grid_keep_kernel(const float* __restrict__ pts, uint32_t n, uint32_t stride, GridHeader* __restrict__ hdr,
                                                        const uint8_t* __restrict__ mask, int mshift, float4* __restrict__ out, uint32_t out_cap,
                                                        uint32_t* __restrict__ counter) {
    __shared__ uint32_t sh4[4];
    __shared__ uint32_t sh_base;
    constexpr int kPer = 8;
    const GridHeader h = *hdr;
    if (h.overflow || h.empty) return;
    for (uint32_t c0 = blockIdx.x * (256u * kPer); c0 < n; c0 += gridDim.x * (256u * kPer)) {
        float px[kPer], py[kPer], pz[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t i = c0 + u * 256 + threadIdx.x;
            load_xyz<kVec>(pts, i < n ? i : c0, stride, px[u], py[u], pz[u]);
        }
        uint32_t mi[kPer];      // the mask byte of the point's cell, or none
        bool outside = false;
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t i = c0 + u * 256 + threadIdx.x;
            uint32_t key, cc[3];
            bool out_u = false;
            mi[u] = 0xffffffffu;
            if (i < n && point_key(h, px[u], py[u], pz[u], &key, &out_u, cc)) mi[u] = roi_macro(h, mshift, (int)cc[0], (int)cc[1], (int)cc[2]);
            outside = outside || (i < n && out_u);
        }
        if (outside && !h.clamped) hdr->stale = 1;      // (only a box reused from the previous build can be too small)
        uint8_t mb[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) mb[u] = mask[mi[u] != 0xffffffffu ? mi[u] : 0u];      // (unconditional loads: all in flight)
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < kPer; ++u) { mb[u] = (mi[u] != 0xffffffffu && mb[u] != 0) ? 1 : 0; mine += mb[u]; }
        uint32_t total;
        uint32_t off = block_exclusive_scan_256(mine, &total, sh4);
        if (threadIdx.x == 0) sh_base = total ? atomicAdd(counter, total) : 0u;
        __syncthreads();
        off += sh_base;
#pragma unroll
        for (int u = 0; u < kPer; ++u)
            if (mb[u]) { if (off < out_cap) out[off] = make_float4(px[u], py[u], pz[u], __uint_as_float(c0 + u * 256 + threadIdx.x)); ++off; }
        __syncthreads();      // sh_base and sh4 are written again in the next chunk
    }
}

template <bool kVec>
__global__ __launch_bounds__(256) void grid_place_kernel(const float* __restrict__ pts, uint32_t n, uint32_t stride, const GridHeader* __restrict__ hdr,
                                                         const uint32_t* __restrict__ bin_count, const uint32_t* __restrict__ slot,
                                                         const uint32_t* __restrict__ bin_start, float4* __restrict__ tiled, int shift) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    uint32_t* const sh_start = dyn_lds;      // max_bins + 1 entries
    const GridHeader h = *hdr;
    if (h.overflow || h.empty) return;
    const uint32_t nbins = (uint32_t)(h.n_cells >> shift) + 1u;
    constexpr int kPer = 4;
    float px[kPer], py[kPer], pz[kPer];
    uint32_t sl[kPer];
    // the first chunk's points are requested BEFORE the scan of the tile counters: two independent memory round trips overlap
    uint32_t c0 = blockIdx.x * (256 * kPer);
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const uint32_t i = c0 + u * 256 + threadIdx.x;
        load_xyz<kVec>(pts, i < n ? i : 0u, stride, px[u], py[u], pz[u]);
        sl[u] = i < n ? slot[i] : 0xffffffffu;
    }
    for (uint32_t b = threadIdx.x; b <= nbins; b += 256) sh_start[b] = bin_start[b];      // (scanned by the last block of the bin kernel)
    __syncthreads();
    for (; c0 < n; c0 += gridDim.x * (256 * kPer)) {
        if (c0 != blockIdx.x * (256 * kPer)) {
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const uint32_t i = c0 + u * 256 + threadIdx.x;
                load_xyz<kVec>(pts, i < n ? i : c0, stride, px[u], py[u], pz[u]);
                sl[u] = i < n ? slot[i] : 0xffffffffu;
            }
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t i = c0 + u * 256 + threadIdx.x;
            uint32_t key;
            if (sl[u] != 0xffffffffu && point_key(h, px[u], py[u], pz[u], &key))
                tiled[sh_start[key >> shift] + sl[u]] = make_float4(px[u], py[u], pz[u], __uint_as_float(i));
        }
    }
}

// rank of this lane's point among the points of cell c counted so far; one atomic per run of equal cells in consecutive lanes (every lane of the wave calls)
__device__ inline uint32_t hist_claim_runs(uint32_t* hist, uint32_t c, bool valid) {
    const int lane = threadIdx.x & 63;
    const uint32_t cc = valid ? c : 0xffffffffu;
    const uint32_t prev = __shfl_up(cc, 1, 64);
    const bool leader = lane == 0 || prev != cc;
    const unsigned long long lead = __ballot(leader);
    const unsigned long long below = lead & (~0ull >> (63 - lane));
    const int start = 63 - __clzll(below);
    const unsigned long long above = lane == 63 ? 0ull : (lead >> (lane + 1)) << (lane + 1);
    const int end = above ? __ffsll((long long)above) - 1 : 64;
    uint32_t base = 0;
    if (leader && valid) base = atomicAdd(&hist[c], (uint32_t)(end - start));
    base = __shfl(base, start, 64);
    return base + (uint32_t)(lane - start);
}

// ---- block 0 of the tile pass: the layout of the NEXT build, planned beside the blocks that sort ----
// In: this build's bin_start (points per bin) and the layout the cloud was binned by (lay_cur; nullptr: by tile).  Out: lay_next.  Per tile, with k the
// current number of cuts: one more when its fullest slab holds more than `hs` points (and a slab would keep >= 4 cells), one fewer when every pair of
// sibling slabs together holds <= hs / 2 (hysteresis: a tile does not flap), else unchanged -- never more than nb_max bins in all (then nothing is
// cut further).  A thread owns a run of consecutive tiles, hence of consecutive bins: one block scan of (bins, room) gives every thread its first bin
// and its first position in `tiled`.  copy_only (a build of a region: its counts are not the cloud's): the layout is handed on as it came.
// (Measured and not kept, round 5: the scan of the bin counters here too, its results published to the other blocks of the launch as tagged 64-bit
//  words they polled -- it takes the ticket and the last block's scan off the bin pass, but block 0's 8 K counter loads queue behind the 16 MB the
//  other blocks ask for at the same moment: published after 12 us at 1 M points, 28 us at 10 M, every block waiting; profiles/r05_notes.md.)
struct TilePlan { uint32_t* lay_next; uint32_t hs; uint32_t nb_max; int32_t enabled; int32_t copy_only; int32_t cuts; int32_t room_shift; uint32_t room_add; uint32_t pad_; };      // room_shift: a bin's room = what it held + that >> room_shift + 32 (3: an eighth)      // cuts: tiles may be cut (hs finite) -- else bin = tile, always

template <int kThreads>
__device__ void tile_block0(const GridHeader* __restrict__ hdr, int shift, const uint32_t* __restrict__ bin_start,
                            const uint32_t* __restrict__ lay_cur, const TilePlan plan, uint32_t* lds) {
    __shared__ unsigned long long sh_scan[kThreads / 64];
    uint32_t nb_lay = 0;
    if (lay_cur) nb_lay = lay_cur[kLayMeta];
    const GridHeader h = *hdr;
    uint32_t* const out = plan.lay_next;
    const int rs = plan.room_shift;
    const uint32_t ra = plan.room_add;
    const uint32_t ntiles = (uint32_t)(h.n_cells >> shift) + 1u;
    const uint32_t nb = lay_cur ? nb_lay : ntiles;
    const bool unusable = h.overflow || h.empty || h.stale || nb == 0u || nb > plan.nb_max || ntiles > (uint32_t)kMaxBins;
    if (unusable) {      // nothing to go by: no layout for the next build (the other blocks leave before they poll)
        if (threadIdx.x == 0) { out[kLayMeta] = 0u; out[kLayMeta + 1] = 0u; }
        return;
    }
    if (!plan.cuts && nb == ntiles && !plan.copy_only) {
        // bin = tile now and next time (the layouts of clouds that are never cut: NDT's and VGICP's lattices, maps of millions of points): a thread's run
        // of tiles straight from bin_start -- UNCONDITIONAL loads of clamped indices, 9 / 17 / 33 by the block-uniform length of the run -- and one block
        // scan of the rooms; no LDS beyond the scan's (this launch's other blocks keep the occupancy their histogram allows)
        __shared__ uint32_t sh_scan32[kThreads / 64];
        const uint32_t per = (nb + kThreads - 1) / kThreads, b0 = threadIdx.x * per;      // <= kMaxBins / 256 = 32
        // (eight tiles at a time -- nine loads in flight -- so that this block's registers stay below what the sorting blocks of the same kernel need)
        uint32_t room = 0;
        for (uint32_t j0 = 0; j0 < per; j0 += 8u) {
            uint32_t c[9];
#pragma unroll
            for (uint32_t j = 0; j <= 8; ++j) c[j] = bin_start[min(b0 + j0 + j, nb)];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) if (j0 + j < per && b0 + j0 + j < nb) { const uint32_t e = c[j + 1] - c[j]; room += e + (e >> rs) + ra; }
        }
        uint32_t room_total;
        uint32_t lo = block_exclusive_scan<kThreads>(room, &room_total, sh_scan32);
        for (uint32_t j0 = 0; j0 < per; j0 += 8u) {
            uint32_t c[9];
#pragma unroll
            for (uint32_t j = 0; j <= 8; ++j) c[j] = bin_start[min(b0 + j0 + j, nb)];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                if (j0 + j < per && b0 + j0 + j < nb) {
                    const uint32_t e = c[j + 1] - c[j], b = b0 + j0 + j;
                    out[b] = lo; out[kLaySub + b] = b; out[kLayMap + b] = b;
                    lo += e + (e >> rs) + ra;
                }
        }
        if (threadIdx.x == 0) { out[nb] = room_total; out[kLayMeta] = nb; out[kLayMeta + 1] = ntiles; }
        return;
    }
    uint32_t* const start = lds;                                                            // nb + 1 (<= nb_max + 1)
    uint16_t* const sub = reinterpret_cast<uint16_t*>(lds + plan.nb_max + 2);              // ntiles: first bin | k << 13
    for (uint32_t i = threadIdx.x; i <= nb; i += kThreads) start[i] = bin_start[i];
    if (plan.copy_only) {
        if (!lay_cur) { if (threadIdx.x == 0) { out[kLayMeta] = 0u; out[kLayMeta + 1] = 0u; } return; }
        for (uint32_t i = threadIdx.x; i <= nb; i += kThreads) out[i] = lay_cur[i];
        for (uint32_t i = threadIdx.x; i < ntiles; i += kThreads) out[kLaySub + i] = lay_cur[kLaySub + i];
        for (uint32_t i = threadIdx.x; i < nb; i += kThreads) out[kLayMap + i] = lay_cur[kLayMap + i];
        if (threadIdx.x == 0) { out[kLayMeta] = nb; out[kLayMeta + 1] = ntiles; }
        return;
    }
    int deep = 0;      // some tile is cut four times or more
    for (uint32_t t = threadIdx.x; t < ntiles; t += kThreads) { const uint16_t e = lay_cur ? (uint16_t)lay_cur[kLaySub + t] : (uint16_t)t; sub[t] = e; deep |= (e >> 13) >= 4u ? 1 : 0; }
    const bool by_bins = __syncthreads_or(deep) != 0;
    DEV_STAMP(1, 5);
    const int kmax = plan.cuts ? min(kMaxSplit, shift - 2) : 0;      // (no cuts: a layout that holds some -- planned when the cloud was smaller -- is merged back step by step)
    // a thread's run of tiles: an equal share of the tiles -- or, once some tile is cut deep, those whose first bin lies in its share of the BINS (the work
    // below is per bin: with equal shares of the tiles the thread that owned the eight tiles under the vehicle's path, cut into 64 slabs each, walked 512
    // bins three times over while the others had left -- the launch took 166 us, round 5; the first bins of the tiles ascend, so a share's first tile is
    // found by bisection.  Shallow layouts keep the equal shares: this block is the tile pass's longest at 1 M points, and the two bisections -- 22
    // dependent LDS reads -- made that pass 1.7 us longer)
    auto first_tile = [&](uint32_t i) -> uint32_t {
        const uint32_t want = (uint32_t)(((unsigned long long)i * nb) / kThreads);
        uint32_t lo = 0, hi = ntiles;      // smallest t with first_bin(t) >= want (ntiles: none)
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((sub[mid] & 0x1fffu) < want) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const uint32_t per = (ntiles + kThreads - 1) / kThreads;
    const uint32_t t0 = by_bins ? first_tile(threadIdx.x) : min(threadIdx.x * per, ntiles);
    const uint32_t t1 = by_bins ? (threadIdx.x + 1 == kThreads ? ntiles : first_tile(threadIdx.x + 1)) : min(t0 + per, ntiles);
    // what becomes of tile t: new k in the low bits of the result, its room << 8
    auto decide = [&](uint32_t t, bool may_cut) -> unsigned long long {
        const uint32_t e = sub[t], base = e & 0x1fffu;
        const int k = (int)(e >> 13);
        const uint32_t nc = 1u << k;
        uint32_t maxc = 0, maxpair = 0, prev = 0;
        unsigned long long same = 0, merged = 0;
        for (uint32_t i = 0; i < nc; ++i) {
            const uint32_t c = start[base + i + 1] - start[base + i];
            maxc = max(maxc, c);
            same += (unsigned long long)c + (c >> rs) + ra;
            if (i & 1u) { const uint32_t s2 = prev + c; maxpair = max(maxpair, s2); merged += (unsigned long long)s2 + (s2 >> rs) + ra; }
            prev = c;
        }
        if (may_cut && maxc > plan.hs && k < kmax) return (unsigned long long)(k + 1) | ((2ull * same) << 8);      // (both children get their parent's room)
        if (k > 0 && maxpair <= plan.hs / 2u) return (unsigned long long)(k - 1) | (merged << 8);
        return (unsigned long long)k | (same << 8);
    };
    bool may_cut = true;
    unsigned long long mine = 0, total = 0, off = 0;      // bins << 40 | room
    for (int attempt = 0; attempt < 2; ++attempt) {
        mine = 0;
        for (uint32_t t = t0; t < t1; ++t) { const unsigned long long d = decide(t, may_cut); mine += (1ull << (40 + (d & 0xffu))) + (d >> 8); }
        off = block_exclusive_scan_u64<kThreads>(mine, &total, sh_scan);
        __syncthreads();      // sh_scan is reused
        if ((total >> 40) <= (unsigned long long)plan.nb_max) break;      // (block-uniform)
        may_cut = false;      // too many bins: nothing is cut further (merges only: never more bins than now)
    }
    if ((total >> 40) > (unsigned long long)plan.nb_max || (total & 0xffffffffffull) > 0xfffffff0ull) {      // (cannot happen with nb <= nb_max; a layout nobody can use says so)
        if (threadIdx.x == 0) { out[kLayMeta] = 0u; out[kLayMeta + 1] = 0u; }
        return;
    }
    DEV_STAMP(1, 6);
    uint32_t nbin = (uint32_t)(off >> 40), pos = (uint32_t)(off & 0xffffffffffull);
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t e = sub[t], base = e & 0x1fffu;
        const int k = (int)(e >> 13), kn = (int)(decide(t, may_cut) & 0xffu);
        out[kLaySub + t] = nbin | ((uint32_t)kn << 13);
        for (uint32_t i = 0; i < (1u << kn); ++i) {
            uint32_t est;
            if (kn == k) est = start[base + i + 1] - start[base + i];
            else if (kn > k) est = start[base + (i >> 1) + 1] - start[base + (i >> 1)];
            else est = start[base + 2 * i + 2] - start[base + 2 * i];
            out[kLayMap + nbin] = t | (i << 13) | ((uint32_t)kn << 26);
            out[nbin] = pos;
            pos += est + (est >> rs) + ra;
            ++nbin;
        }
    }
    if (threadIdx.x == kThreads - 1) { out[nbin] = pos; out[kLayMeta] = nbin; out[kLayMeta + 1] = ntiles; }      // (the last thread ends at the totals)
    DEV_STAMP(1, 7);
}

// The other blocks: a bin each (a tile, or one of the slabs a heavy tile was cut into: see BINS above), bin after bin -- the grid is sized to what the
// device holds at once, and what a block needs to know of its NEXT bin (points, place, cells) is requested while it works on the current one: a block
// launched per bin paid that round trip, and the header's, in front of every bin (round 5: median block 5.5 us, of which 2 before its points were asked for).
// Dynamic LDS: the bin's cell histogram (<= 2^shift counters; block 0 keeps the bins' positions and the tiles' words there).  A bin of up to
// 256 * kTilePer points is held in registers between the two passes (all its loads in flight at once); a larger one goes through the generic loop
// (ranks parked in global scratch).
// kTilePer = 16 (64 VGPRs of points; the kernel then needs 223 VGPRs: TWO waves per SIMD, 512 blocks in flight).  That is right for a
// grid whose tiles all hold several hundred points (LOAM's 1 m cells over a 0.5 m map), and wrong for a fine lattice over the same map:
// the 0.5 m voxel lattice makes 3 350 tiles of which most hold nothing or a handful of points, they go through the kernel in 6.5 rounds of
// 512 blocks -- 123 us.  A sparse grid is therefore served by TWO instantiations over the same tiles: kMode 1 takes the tiles of up to 256
// points with one point per thread (few VGPRs, many blocks in flight), kMode 2 the others with sixteen; kMode 0 = every tile (dense grids).
// kTail (NDT's region-only targets, pcr_internal.h: TileTail): while a tile's cell counts are in registers the cells that will carry a voxel
// are listed and every cell's slot is written -- ndt_candidates_kernel's work without its launch and without reading the table again.
// (Measured and not kept, round 5: for VGICP's region-only lattice -- a sparse grid, 16.7 M cells of 0.5 m of which the scan's region is a fraction -- a
//  variant that stores nothing for the cells outside the region's mask: a bin none of whose cells lies in the mask left at once, of the others only the
//  starts of the four-cell groups that touch the mask written, the lookups testing the mask first.  The table's 67.6 MB of writes went away and the two
//  launches got SLOWER, 22.1 -> 30.8 us and 29.5 -> 41.9 us: a block of this pass is a chain of latencies, not a stream of stores, and the mask test put
//  one more dependent round trip -- header, row decode, mask bytes, a block-wide OR -- in front of every one of its 3 350 blocks.  The region-only
//  index itself (the bin pass drops the points outside the mask) is kept: profiles/r05_notes.md.)
// kRuns (GridIndex::coherent_input -- the voxel filter's clouds: scans and concatenations of scans, stored ring by ring): one LDS atomic per RUN of equal
// cells in consecutive lanes, as the bin pass does for bins.  Consecutive returns of a ring fall into the same 0.5 m voxel eight to forty at a time, and
// 64 same-address LDS atomics serialise: the slab of 32 cells under the vehicle's path -- 12 000 points -- took its block 170 us (round 5).
template <int kTilePer, int kMode, int kThreads, bool kTail, bool kPlan, bool kRuns = false>
// kPlan: the launch that carries block 0 (tile_block0; plan.enabled says whether it does).  src_start: where bin b's points lie in `tiled` -- bin_start after the placing pass, the layout hint when the bin kernel placed them.
// lay_cur: the layout the points were binned by (nullptr: by tile).  plan.enabled: block 0 is tile_block0 and sorts nothing.
// (forcing the 8-per-thread instantiation to 128 registers -- four blocks per CU instead of three -- changed nothing: 46.7 / 47.2 us either way, round 5)
__global__ __launch_bounds__(kThreads) void grid_tile_kernel(const GridHeader* __restrict__ hdr_in, unsigned long long* __restrict__ tile_sq, const uint32_t* __restrict__ bin_start,
                                                        uint32_t* __restrict__ bin_count, const float4* __restrict__ tiled, uint32_t* __restrict__ cell_start,
                                                        float4* __restrict__ sorted, uint32_t* __restrict__ scratch_rank, int shift,
                                                        const uint32_t* __restrict__ src_start, const uint32_t* __restrict__ lay_cur, const TilePlan plan, const TileTail tail) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    uint32_t* const hist = dyn_lds;                    // <= 1 << shift
    __shared__ uint32_t sh4[kThreads / 64];
    __shared__ unsigned long long sh_sq[kThreads / 64];
    __shared__ uint32_t sh_tail_base;
    DEV_STAMP(1, 0);
    const uint32_t first_blk = kPlan && plan.enabled ? 1u : 0u;
    const uint32_t blk = blockIdx.x - first_blk, nblk = gridDim.x - first_blk;      // (block 0 of a planning launch never uses them)
    // what this block needs to know of its first bin, requested beside the header (the indices are in bounds for every block of the grid: <= kMaxBins blocks)
    const bool worker = !(kPlan && plan.enabled && blockIdx.x == 0);
    uint32_t i_p0 = 0, i_p1 = 0, i_q0 = 0, i_map = 0;
    auto fetch = [&](uint32_t b) {
        i_p0 = bin_start[b]; i_p1 = bin_start[b + 1];
        i_q0 = src_start[b];
        if (lay_cur) i_map = lay_cur[kLayMap + b];
    };
    if (worker) fetch(min(blk, (uint32_t)kMaxBins));
    uint32_t nb_lay = 0;
    if (lay_cur) nb_lay = lay_cur[kLayMeta];
    if (kTail && blockIdx.x == first_blk && threadIdx.x == 0) *tail.count_next = 0u;      // (the other of two counters, for the next call: before anything can return)
    if (kPlan && !worker) { tile_block0<kThreads>(hdr_in, shift, bin_start, lay_cur, plan, dyn_lds); return; }
    const GridHeader h = *hdr_in;
    if (kTail && h.empty && !h.overflow)      // nothing indexed: no cell carries a voxel
        for (uint64_t t = (uint64_t)blk * kThreads + threadIdx.x; t < h.n_cells; t += (uint64_t)nblk * kThreads) tail.vox_slot[t] = 0u;
    if (h.overflow || h.empty) return;
    const uint32_t ntiles = (uint32_t)(h.n_cells >> shift) + 1u;
    uint32_t nb = lay_cur ? nb_lay : ntiles;
    if (h.stale) {
        // a hint did not hold (points outside the box, or a bin without room: not every point was stored): nothing here can be
        // trusted and the caller rebuilds.  Only the counters are put back to zero, the state every build expects.
        if (nb == 0u || nb > (uint32_t)kMaxBins) nb = (uint32_t)kMaxBins;
        for (uint32_t b = blk; b < nb; b += nblk) if (threadIdx.x == 0) bin_count[(size_t)b * kBinStride] = 0u;
        return;
    }
    if (nb > plan.nb_max || ntiles > (uint32_t)kMaxBins) return;      // (block 0 has left too: see tile_block0)
    for (uint32_t bin = blk; bin < nb; bin += nblk) {
        // (taking the heavy tiles first -- an order written by the bin kernel's last block -- was measured: the kernel's span is its heaviest
        //  tile's own 24 us wherever it starts, and the extra scan cost the bin kernel's serial tail 3 us; profiles/r04_notes.md)
        if (bin != blk) fetch(bin);
        const uint32_t p0 = i_p0, np = i_p1 - i_p0;
        const uint32_t q0 = i_q0;      // first point of the bin in `tiled`
        uint64_t cell0 = (uint64_t)bin << shift;
        uint32_t S = 1u << shift;
        if (lay_cur) {
            const uint32_t m = i_map, k = m >> 26;
            S = 1u << (shift - (int)k);
            cell0 = ((uint64_t)(m & 0x1fffu) << shift) + (uint64_t)((m >> 13) & 0x1fffu) * S;
        }
        const bool mine = !(kMode == 1 && np > 256u) && !(kMode == 2 && np <= 256u);      // (block-uniform: else the other instantiation's bin)
        const bool small = np <= (uint32_t)kThreads * kTilePer;      // block-uniform
        float4 p[kTilePer];
        uint32_t cr[kTilePer];
        if (mine && small) {
#pragma unroll
            for (int u = 0; u < kTilePer; ++u) {      // all loads in flight, issued before anything waits
                const uint32_t j = u * (uint32_t)kThreads + threadIdx.x;
                if (u * (uint32_t)kThreads < np) p[u] = tiled[q0 + (j < np ? j : 0u)];
            }
        }
        if (!mine) continue;
        for (uint32_t c = threadIdx.x * 4u; c < S; c += 4u * kThreads) *reinterpret_cast<uint4*>(hist + c) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        if (small) {
#pragma unroll
            for (int u = 0; u < kTilePer; ++u) {
                const uint32_t j = u * (uint32_t)kThreads + threadIdx.x;
                cr[u] = 0u;
                uint32_t c = 0u;
                if (j < np) {
                    uint32_t key = 0;
                    point_key(h, p[u].x, p[u].y, p[u].z, &key);
                    c = key - (uint32_t)cell0;
                    if (!kRuns) cr[u] = (c << 18) | atomicAdd(&hist[c], 1u);          // c < 2^13, rank < 4096 <= 2^18
                }
                if (kRuns && u * (uint32_t)kThreads < np) {      // (block-uniform: whole waves)
                    const uint32_t r = hist_claim_runs(hist, c, j < np);
                    if (j < np) cr[u] = (c << 18) | r;
                }
            }
        }
        if (!small) {
            for (uint32_t j0 = 0; j0 < np; j0 += (uint32_t)kThreads * kTilePer) {      // chunks of 4096 points, their loads in flight together
#pragma unroll
                for (int u = 0; u < kTilePer; ++u) { const uint32_t j = j0 + u * (uint32_t)kThreads + threadIdx.x; p[u] = tiled[q0 + (j < np ? j : 0u)]; }
#pragma unroll
                for (int u = 0; u < kTilePer; ++u) {
                    const uint32_t j = j0 + u * (uint32_t)kThreads + threadIdx.x;
                    uint32_t c = 0u;
                    if (j < np) {
                        uint32_t key = 0;
                        point_key(h, p[u].x, p[u].y, p[u].z, &key);
                        c = key - (uint32_t)cell0;
                        if (!kRuns) scratch_rank[p0 + j] = atomicAdd(&hist[c], 1u);
                    }
                    if (kRuns) {
                        const uint32_t r = hist_claim_runs(hist, c, j < np);
                        if (j < np) scratch_rank[p0 + j] = r;
                    }
                }
            }
        }
        if (threadIdx.x == 0) bin_count[(size_t)bin * kBinStride] = 0u;      // the counters are left zeroed: the state the next build expects
        __syncthreads();
        if (bin == blk) DEV_STAMP(1, 1);
        // exclusive scan of the bin's counters -> cell_start (+ sum of count^2, the density estimate of the header)
        unsigned long long sq = 0;
        uint32_t carry = 0;
        uint32_t listed = 0;      // kTail: bit (4 * round + k) = cell k of this thread's four in that round goes on the list (<= 8 rounds: 2^13 cells per tile)
        for (uint32_t cb = 0; cb < S; cb += 4u * kThreads) {
            const uint32_t c = cb + threadIdx.x * 4;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < S) v = *reinterpret_cast<const uint4*>(hist + c);
            if (kTail && c < S && cell0 + c < h.n_cells) {
                // the four cells are neighbours along x: one decode serves them unless the row ends in between
                const uint32_t g32 = (uint32_t)(cell0 + c), d0 = (uint32_t)h.dims[0], d1 = (uint32_t)h.dims[1], row = g32 / d0, cz = row / d1, cx0 = g32 - row * d0;
                const uint32_t mrow = roi_macro(h, tail.mshift, 0, (int)(row - cz * d1), (int)cz);
                const uint32_t cnt[4] = {v.x, v.y, v.z, v.w};
                uint32_t sl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((uint64_t)g32 + k < h.n_cells) {
                        const bool in = cx0 + k < d0 ? tail.mask[mrow + ((cx0 + k) >> tail.mshift)] != 0 : roi_mask_holds_cell(h, tail.mask, tail.mshift, g32 + k);
                        if (in && (int)cnt[k] >= tail.min_points) listed |= 1u << ((cb / (4u * kThreads)) * 4u + k);
                        sl[k] = in ? 0u : kNdtUnprepared;      // (the index holds nothing outside the mask: every cell there is unprepared)
                    }
                }
                if ((uint64_t)g32 + 3 < h.n_cells) *reinterpret_cast<uint4*>(tail.vox_slot + g32) = make_uint4(sl[0], sl[1], sl[2], sl[3]);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if ((uint64_t)g32 + k < h.n_cells) tail.vox_slot[g32 + k] = sl[k];
                }
            }
            sq += (unsigned long long)v.x * v.x + (unsigned long long)v.y * v.y + (unsigned long long)v.z * v.z + (unsigned long long)v.w * v.w;
            uint32_t tot;
            const uint32_t o = block_exclusive_scan<kThreads>(v.x + v.y + v.z + v.w, &tot, sh4) + carry;
            __syncthreads();      // sh4 is reused by the next round
            if (c < S) {
                const uint4 st = make_uint4(o, o + v.x, o + v.x + v.y, o + v.x + v.y + v.z);
                *reinterpret_cast<uint4*>(hist + c) = st;                                     // hist now holds the offsets inside the bin
                const uint64_t g = cell0 + c;
                if (g + 3 <= h.n_cells) *reinterpret_cast<uint4*>(cell_start + g) = make_uint4(p0 + st.x, p0 + st.y, p0 + st.z, p0 + st.w);
                else {
                    const uint32_t e[4] = {p0 + st.x, p0 + st.y, p0 + st.z, p0 + st.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (g + k <= h.n_cells) cell_start[g + k] = e[k];      // (entry n_cells = number of indexed points)
                }
            }
            carry += tot;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
        if ((threadIdx.x & 63) == 0) sh_sq[threadIdx.x >> 6] = sq;
        __syncthreads();
        // (one 8-byte store per bin; thousands of blocks adding into one header word serialise at the memory side for ~45 us)
        if (threadIdx.x == 0) { unsigned long long t_ = 0; for (int w = 0; w < kThreads / 64; ++w) t_ += sh_sq[w]; tile_sq[bin] = t_; }
        if (kTail) {
            // room on the list for all of the bin's cells with ONE atomic (the order of the list is immaterial: ndt_voxel_kernel takes a cell per thread)
            uint32_t tot;
            uint32_t pos = block_exclusive_scan<kThreads>((uint32_t)__popc(listed), &tot, sh4);
            if (threadIdx.x == 0) sh_tail_base = tot ? atomicAdd(tail.count, tot) : 0u;
            __syncthreads();
            pos += sh_tail_base;
            while (listed) {
                const uint32_t bit = (uint32_t)__ffs((int)listed) - 1u;
                listed &= listed - 1u;
                if (pos < tail.capacity) tail.list[pos] = (uint32_t)cell0 + (bit >> 2) * 4u * kThreads + threadIdx.x * 4u + (bit & 3u);      // (capacity = points / min_points: never short)
                ++pos;
            }
        }
        if (bin == blk) DEV_STAMP(1, 2);
        if (small) {
#pragma unroll
            for (int u = 0; u < kTilePer; ++u) {
                const uint32_t j = u * (uint32_t)kThreads + threadIdx.x;
                if (j < np) sorted[p0 + hist[cr[u] >> 18] + (cr[u] & 0x3ffffu)] = p[u];
            }
        } else {
            for (uint32_t j0 = 0; j0 < np; j0 += (uint32_t)kThreads * kTilePer) {
#pragma unroll
                for (int u = 0; u < kTilePer; ++u) {
                    const uint32_t j = j0 + u * (uint32_t)kThreads + threadIdx.x;
                    p[u] = tiled[q0 + (j < np ? j : 0u)];
                    cr[u] = scratch_rank[p0 + (j < np ? j : 0u)];
                }
#pragma unroll
                for (int u = 0; u < kTilePer; ++u) {
                    const uint32_t j = j0 + u * (uint32_t)kThreads + threadIdx.x;
                    if (j < np) {
                        uint32_t key = 0;
                        point_key(h, p[u].x, p[u].y, p[u].z, &key);
                        sorted[p0 + hist[key - (uint32_t)cell0] + cr[u]] = p[u];
                    }
                }
            }
        }
        __syncthreads();
        if (bin == blk) DEV_STAMP(1, 3);
    }
    DEV_STAMP(1, 4);
}

// sum of count^2 over the cells = sum of the tiles' sums -> header (only VGICP's choice of a search cell reads it)
__global__ __launch_bounds__(256) void grid_density_kernel(GridHeader* __restrict__ hdr, const unsigned long long* __restrict__ tile_sq, int shift, const uint32_t* __restrict__ lay_cur) {
    __shared__ unsigned long long sh[4];
    if (hdr->overflow || hdr->empty) return;
    const uint32_t nbins = lay_cur ? min(lay_cur[kLayMeta], (uint32_t)kMaxBins) : (uint32_t)(hdr->n_cells >> shift) + 1u;      // (one sum per bin: see BINS)
    unsigned long long s = 0;
    // (sixteen loads in flight per step: one by one the ~13 loads of a thread were a chain of round trips -- 7.8 us for a kernel that adds up 27 KB)
    for (uint32_t b0 = threadIdx.x; b0 < nbins; b0 += 16u * 256u) {
        unsigned long long w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const uint32_t b = b0 + (uint32_t)u * 256u; w[u] = b < nbins ? tile_sq[b] : 0ull; }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += w[u];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) hdr->sum_sq_u = sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- region of interest of a target prepared for one scan (pcr_internal.h: RoiView) ----------------------------------
// A marked macro cell stands for points that a change of the pose can move by base_m metres (translation, and the reach of the optimiser's
// neighbourhood) + per_m metres per metre of distance from the sensor (rotation: a far point moves by range x angle): every macro cell within
// that Chebyshev radius r of a mark belongs to the region.  Three launches:
//   mark   the macro cell every scan point lands in at the initial pose (a point outside the lattice: the nearest boundary macro cell), and --
//          by the first lane to find the cell unmarked -- the cells within r(mark) + 1 of it along x.  Byte stores, skipped when the byte is
//          already set (neighbouring lanes are neighbouring beams): bit 1 = "a point landed here and its row has been spread", bit 0 = region;
//          the stores race and the loser is a plain 1 over a 3, which costs a later point of that cell a second spread, never a cell.
//   y, z   a 1-D GATHER each, with the radius R = r + 1 evaluated at the OUTPUT cell: the radius changes by at most one macro cell over its
//          own reach (per_m * sqrt 3 * r < 1 for r <= 11), so every cell the exact dilation holds is held (a few more are: harmless).  The
//          y pass also clears the marks of the NEXT call (two buffers, alternating, like NDT's candidate counters): no memset per call.
// (Round 3 gathered along x too, in a launch of its own.  Measured and not kept: y and z in ONE 2-D gather -- 225 loads per cell where the
//  two passes need 30: 25 us against 12; a scatter from the marks in all three directions -- 90 us, a handful of lanes per wave walking
//  cubes of hundreds of bytes.)
static constexpr int kRoiMaxRadius = 11;      // macro cells (22 m at 2 m macro cells): the bound of the gather's "+ 1" above (per_m * sqrt 3 * r < 1)
__device__ __forceinline__ int roi_radius(const GridHeader& h, int mshift, uint32_t x, uint32_t y, uint32_t z, double sx, double sy, double sz, double base_m, double per_m) {
    const double edge = h.cell * (double)(1u << mshift);
    // centre of the macro cell in metres: cell i of axis d spans [(org_d + shift + i) cell, + cell)
    const double cx = (h.org[0] + h.shift) * h.cell + ((double)x + 0.5) * edge, cy = (h.org[1] + h.shift) * h.cell + ((double)y + 0.5) * edge,
                 cz = (h.org[2] + h.shift) * h.cell + ((double)z + 0.5) * edge;
    const double range = sqrt((cx - sx) * (cx - sx) + (cy - sy) * (cy - sy) + (cz - sz) * (cz - sz)) + 0.87 * edge;
    double rr = ceil((base_m + per_m * range) / edge);
    if (!(rr >= 1.0)) rr = 1.0;
    return (int)fmin(rr, (double)kRoiMaxRadius) + 1;
}
__global__ __launch_bounds__(256) void roi_mark_kernel(const float* __restrict__ src, uint32_t n, uint32_t stride, const Pose16 T, const GridHeader* __restrict__ lat,
                                                       uint8_t* __restrict__ mark, int mshift, double base_m, double per_m, const BlobStore blob) {
    // (a rider: words for a later launch on this stream, stored by the last block whatever the header says -- pcr_internal.h: BlobStore)
    if (blob.n && blockIdx.x == gridDim.x - 1) {
        for (uint32_t t = threadIdx.x; t < blob.n; t += 256) blob.dst[t] = blob.w[t];
        if (blob.zero && threadIdx.x == 0) *blob.zero = 0u;
    }
    const GridHeader h = *lat;
    if (h.overflow || h.empty || h.stale) return;
    const uint32_t m0 = ((uint32_t)h.dims[0] + (1u << mshift) - 1u) >> mshift;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float* sp = src + (size_t)i * stride;
        const double p[3] = {(double)sp[0], (double)sp[1], (double)sp[2]};
        int c[3];
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double tp = T.m[r] * p[0] + T.m[4 + r] * p[1] + T.m[8 + r] * p[2] + T.m[12 + r];
            const double f = floor(tp / h.cell - h.shift) - h.org[r];
            ok = ok && f == f;
            c[r] = (int)fmin(fmax(f, 0.0), (double)(h.dims[r] - 1));
        }
        if (!ok) continue;      // a non-finite point lands nowhere
        const uint32_t m = roi_macro(h, mshift, c[0], c[1], c[2]);
        if (mark[m] & 2) continue;
        mark[m] = 3;
        const int x = c[0] >> mshift;
        const int R = roi_radius(h, mshift, (uint32_t)x, (uint32_t)c[1] >> mshift, (uint32_t)c[2] >> mshift, T.m[12], T.m[13], T.m[14], base_m, per_m);
        const int k0 = max(x - R, 0), k1 = min(x + R, (int)m0 - 1);
        uint8_t* row = mark + ((size_t)m - (size_t)x);
        for (int k = k0; k <= k1; ++k) if (row[k] == 0) row[k] = 1;
    }
}
// one 1-D gather (axis 1 = y, 2 = z); clear_next (the y pass): the marks of the next call
__global__ __launch_bounds__(256) void roi_spread_axis_kernel(const GridHeader* __restrict__ lat, const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                              uint8_t* __restrict__ clear_next, int axis, int mshift, double sx, double sy, double sz, double base_m, double per_m) {
    const GridHeader h = *lat;
    if (h.overflow || h.empty) return;
    const uint32_t m0 = ((uint32_t)h.dims[0] + (1u << mshift) - 1u) >> mshift, m1 = ((uint32_t)h.dims[1] + (1u << mshift) - 1u) >> mshift,
                   m2 = ((uint32_t)h.dims[2] + (1u << mshift) - 1u) >> mshift;
    const uint32_t total = m0 * m1 * m2;
    for (uint32_t m = blockIdx.x * 256 + threadIdx.x; m < total; m += gridDim.x * 256) {
        if (clear_next) clear_next[m] = 0;
        const uint32_t row = m / m0, x = m - row * m0, z = row / m1, y = row - z * m1;
        const int R = roi_radius(h, mshift, x, y, z, sx, sy, sz, base_m, per_m);
        const int pos = axis == 1 ? (int)y : (int)z, len = axis == 1 ? (int)m1 : (int)m2;
        const size_t step = axis == 1 ? (size_t)m0 : (size_t)m0 * m1;
        const int k0 = max(pos - R, 0), k1 = min(pos + R, len - 1);
        const uint8_t* p = in + ((size_t)m - (size_t)(pos - k0) * step);
        // (every load of the span in flight at once -- positions past the span repeat its last one: taken one by one they are a chain of up to 25
        //  round trips, and that chain was the kernel's duration)
        const int span = k1 - k0;
        uint8_t w[2 * kRoiMaxRadius + 3];
#pragma unroll
        for (int k = 0; k < 2 * kRoiMaxRadius + 3; ++k) w[k] = p[(size_t)min(k, span) * step];
        uint8_t any = 0;
#pragma unroll
        for (int k = 0; k < 2 * kRoiMaxRadius + 3; ++k) any |= w[k];
        out[m] = any ? 1 : 0;
    }
}

hipError_t roi_launch(const GridIndex& lattice, const float* d_src, size_t n_src, size_t stride_floats, const Pose16& T, int mshift,
                      uint8_t* d_mark, uint8_t* d_mark_next, uint8_t* d_tmp, uint8_t* d_mask, double base_m, double per_m, hipStream_t s, const BlobStore* blob) {
    const int mb = (int)std::min<size_t>(1024, (n_src + 255) / 256 ? (n_src + 255) / 256 : 1);
    static BlobStore no_blob;      // (zero-initialised: n = 0)
    hipLaunchKernelGGL(roi_mark_kernel, dim3(mb), dim3(256), 0, s, d_src, (uint32_t)n_src, (uint32_t)stride_floats, T, lattice.header.as<GridHeader>(), d_mark, mshift, base_m, per_m,
                       blob ? *blob : no_blob);
    const size_t macros = (lattice.cell_capacity >> (3 * mshift)) + 1024;      // (an estimate is enough: the kernels stride)
    const int db = (int)std::min<size_t>(4096, (macros + 255) / 256);
    const GridHeader* hdr = lattice.header.as<GridHeader>();
    hipLaunchKernelGGL(roi_spread_axis_kernel, dim3(db), dim3(256), 0, s, hdr, (const uint8_t*)d_mark, d_tmp, d_mark_next, 1, mshift, T.m[12], T.m[13], T.m[14], base_m, per_m);
    hipLaunchKernelGGL(roi_spread_axis_kernel, dim3(db), dim3(256), 0, s, hdr, (const uint8_t*)d_tmp, d_mask, (uint8_t*)nullptr, 2, mshift, T.m[12], T.m[13], T.m[14], base_m, per_m);
    return hipGetLastError();
}

// ---- host side ----------------------------------------------------------------------
hipError_t GridIndex::enqueue_density(hipStream_t s) {
    if (tiled_shift < 0) return hipSuccess;      // the atomic build path leaves its estimate in header.sum_sq
    hipLaunchKernelGGL(grid_density_kernel, dim3(1), dim3(256), 0, s, header.as<GridHeader>(), tile_sq.as<unsigned long long>(), tiled_shift,
                       used_layout ? (const uint32_t*)layout[lay_idx ^ 1].as<uint32_t>() : (const uint32_t*)nullptr);      // (the layout the last build was binned by: build() has flipped lay_idx since)
    return hipGetLastError();
}

hipError_t DeviceBuf::reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    // (doubling, where round 4 added a quarter: a buffer that grows is freed and allocated again -- a device-wide stop each -- and the buffers of a voxel filter
    //  whose clouds grow by a key frame at a time did that five times each on the way to eight key frames: scripts/probe_drive_twice.py)
    size_t want = 2 * bytes + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; cap = 0; return e; }
    cap = want;
    return hipSuccess;
}
void DeviceBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
}

void GridIndex::release() {
    sorted.release(); cell_count.release(); cell_start.release(); block_sums.release();
    bbox_partials.release(); header.release(); keys.release(); ranks.release(); ticket.release();
    tiled.release(); bin_count.release(); bin_start.release(); tile_sq.release(); kept.release(); layout[0].release(); layout[1].release(); lay_ok = false;
    cell_capacity = 0; valid = false; n_points = 0;
}

hipError_t GridIndex::grow_cells(uint64_t need_cells, std::string* err) {
    if (need_cells > 4000000000ull) {
        if (err) *err = "target bounding box needs " + std::to_string(need_cells) + " grid cells (> 4e9): cloud too sparse for the dense index";
        return hipErrorInvalidValue;
    }
    if (cells_hint && need_cells + 1 <= cell_capacity) { cells_hint = need_cells; return hipSuccess; }      // only the bound taken from the previous build was too small
    cells_hint = 0;
    const size_t want = (size_t)need_cells + need_cells / 2 + 4096;
    cell_count.release(); cell_start.release(); block_sums.release();
    hipError_t e;
    // (+ one tile: the scan works on whole 16-byte groups)
    if ((e = cell_count.reserve((want + kScanTile) * sizeof(uint32_t))) != hipSuccess ||
        (e = cell_start.reserve((want + kScanTile) * sizeof(uint32_t))) != hipSuccess ||
        (e = block_sums.reserve(2 * (want / kScanTile + 2) * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(cell_count.p, 0, cell_count.cap)) != hipSuccess ||      // builds expect and leave the counters zeroed
        (e = hipDeviceSynchronize()) != hipSuccess) {      // (a device memset is not ordered against the handle's non-blocking stream)
        if (err) *err = std::string("hipMalloc of the cell table failed: ") + hipGetErrorString(e);
        cell_capacity = 0;
        return e;
    }
    cell_capacity = want;
    return hipSuccess;
}

#define PCR_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { if (err) *err = std::string(#x) + ": " + hipGetErrorString(_e); return _e; } } while (0)

// header and cell table exist (first call: a table of 2^20 cells)
hipError_t GridIndex::ensure_tables(hipStream_t s, std::string* err) {
    PCR_TRY(header.reserve(sizeof(GridHeader)));
    if (cell_capacity == 0) {
        // first guess; a too-small table is detected on the device (header.overflow)
        // and the caller grows it with grow_cells() and retries
        size_t guess = 1u << 20;
        PCR_TRY(cell_count.reserve((guess + kScanTile) * sizeof(uint32_t)));
        PCR_TRY(cell_start.reserve((guess + kScanTile) * sizeof(uint32_t)));
        PCR_TRY(hipMemsetAsync(cell_count.p, 0, cell_count.cap, s));      // builds expect and leave the counters zeroed
        cell_capacity = guess;
        PCR_TRY(block_sums.reserve(2 * (guess / kScanTile + 2) * sizeof(uint32_t)));
    }
    return hipSuccess;
}
// The cell count a build may use: the table's capacity, or -- once a header of this index has been seen (note_cells) -- twice that header's count.
size_t GridIndex::effective_capacity() const {
    size_t cap_eff = cell_capacity;
    if (cells_hint) cap_eff = std::min<size_t>(cell_capacity, std::max<size_t>(2 * (size_t)cells_hint + 64, 1024));
    return cap_eff;
}

hipError_t GridIndex::build(const float* d_pts, size_t n, size_t stride_floats, double cell, hipStream_t s, std::string* err, double shift,
                            int pcl_mode, const ClampBox* clamp, bool allow_hint, BuildFilter* filter) {
    valid = false;
    filtered = false;
    if (filter) { filter->applied = false; filter->tail_applied = false; }
    const bool force_atomic_path = dev_env("PCR_INDEX_ATOMIC") != nullptr;      // A/B switch for profiling the two build paths
    if (n > 0xfffffff0ull) { if (err) *err = "target cloud too large (>= 2^32 points)"; return hipErrorInvalidValue; }
    const size_t n_res = std::max(n, reserve_points);      // (reserve_points: the caller knows its clouds will grow to that -- the sub-map assembly's concatenations)
    PCR_TRY(sorted.reserve((n_res + 16) * sizeof(float4)));   // padded: the search reads whole chunks
    PCR_TRY(bbox_partials.reserve(kBBoxBlocks * 6 * sizeof(float)));
    PCR_TRY(keys.reserve((n_res + 1) * sizeof(uint32_t)));
    PCR_TRY(ranks.reserve((n_res + 1) * sizeof(uint32_t)));
    if (!ticket.p) {
        PCR_TRY(ticket.reserve(256));
        PCR_TRY(hipMemsetAsync(ticket.p, 0, 256, s));
    }
    PCR_TRY(ensure_tables(s, err));
    ClampBox cb;
    memset(&cb, 0, sizeof cb);
    if (clamp) cb = *clamp;
    const uint32_t n32 = (uint32_t)n, st = (uint32_t)stride_floats;
    const int pt_blocks = (int)std::min<size_t>(2048, (n + 255) / 256 ? (n + 255) / 256 : 1);
    // The cell count this build may use: the table's capacity, or -- once a header of this index has been seen (note_cells) -- twice that
    // header's count.  The tile size follows from it, and a grid of a few hundred cells (the coarse levels of the covariance search)
    // must not be cut into 256-cell tiles: two blocks then sorted a whole scan between them (158 us).  A cloud that needs more cells
    // than the bound raises header.overflow like a table that is too small, and grow_cells() lifts the bound.
    const size_t cap_eff = effective_capacity();
    // tile size: ~512 points per tile on average, at most 4096 tiles, at least 4 cells per tile (16-byte accesses of the tile kernel)
    // (the bound was 2048: the 5 M-point map of the NDT configuration -- 2.45 M cells of 1 m -- then got 1 195 tiles of 2 048 cells; with 2 390 tiles
    //  of 1 024 a call takes 0.321 instead of 0.337 ms, with 4 780 of 512 0.361, A/B on one box; clouds of up to 1 M points are not affected)
    const size_t tiles_target = std::min<size_t>(4096, std::max<size_t>(64, n / 512));
    // (never more than 2^11 cells per tile unless the counter array forces it: a tile's block zeroes, scans and writes every cell of it,
    //  and a sparse fine grid -- 3.6 M cells for a 65 k-point scan -- is better served by many small tiles than by 440 blocks of 8 192 cells)
    int tshift;
    // (the voxel filter's grids -- 65 536 points over 2.3 M cells of 0.5 m -- take tiles of 4 096 cells: half the bins for the bin pass's last block to read back,
    //  7 % off a scan's filter; 8 192 cells: slower again.  Round 5, scripts/seq_breakdown.py, two rounds on one box.)
    static const int vf_cap = dev_env("PCR_VF_TILE_SHIFT_MAX") ? atoi(dev_env("PCR_VF_TILE_SHIFT_MAX")) : 12;
    const int tshift_cap = cut_sparse ? vf_cap : 11;
    if (cells_hint) { tshift = 2; while ((cells_hint >> tshift) > tiles_target && tshift < tshift_cap) ++tshift; }
    else { tshift = 8; while (((uint64_t)cap_eff >> tshift) + 2 > 2048 && tshift < 11) ++tshift; }
    if (const char* e = dev_env("PCR_TILE_SHIFT")) tshift = std::max(2, atoi(e));      // (development: tile size sweep)
    while (((uint64_t)cap_eff >> tshift) + 2 > (uint64_t)kMaxBins) ++tshift;
    const bool tiled_path = tshift <= kMaxTileShift && !force_atomic_path && !prefer_one_level;
    if (dev_env("PCR_INDEX_DEBUG")) fprintf(stderr, "index build: n %zu cell %.3g capacity %zu cells_hint %llu cap_eff %zu tshift %d hint_ok %d lay_ok %d\n", n, cell, cell_capacity, (unsigned long long)cells_hint, cap_eff, tshift, (int)hint_ok, (int)lay_ok);
    const bool reuse_header = allow_hint && !no_hints && hint_ok && tiled_path && hint_pcl == pcl_mode && hint_shift == shift && !cb.use && hint_cell == cell && tiled_shift == tshift;
    hint_ok = false;      // until the host has seen this build's header (confirm())
    used_hint = reuse_header;
    hint_cell = cell; hint_shift = shift; hint_pcl = pcl_mode;
    // (a header written by the twin's build -- see below -- serves if it was made for this very cloud and cell)
    const bool preset = header_preset && preset_pts == d_pts && preset_n == n && preset_cell == cell && !reuse_header && !cb.use && pcl_mode == 0 && shift == 0.0;
    header_preset = false;
    mirrored = false;
    if (!reuse_header && !preset) {
        HeaderTwin tw;
        memset(&tw, 0, sizeof tw);
        if (twin && twin != this && !cb.use && pcl_mode == 0 && shift == 0.0 && !allow_hint) {
            PCR_TRY(twin->ensure_tables(s, err));
            tw.hdr = twin->header.as<GridHeader>(); tw.mirror = twin->header_mirror; tw.capacity = (uint64_t)twin->effective_capacity(); tw.cell = twin_cell;
            twin->header_preset = true; twin->preset_pts = d_pts; twin->preset_n = n; twin->preset_cell = twin_cell; twin->mirrored = twin->header_mirror != nullptr;
        }
        hipLaunchKernelGGL(grid_bbox_header_kernel, dim3(kBBoxBlocks), dim3(256), 0, s, d_pts, n32, st, bbox_partials.as<float>(),
                           ticket.as<uint32_t>(), header.as<GridHeader>(), (uint64_t)cap_eff, cell, shift, pcl_mode, cb,
                           (allow_hint && !cb.use) ? hint_margin : 0, header_mirror, tw, (allow_hint && !cb.use) ? hint_margin_z_pcl : 0);
        mirrored = header_mirror != nullptr;
    } else if (preset) mirrored = header_mirror != nullptr;
    // Tile size from the CAPACITY of the cell table (the device-side cell count never exceeds it: a larger box is an overflow):
    // ~2048 tiles when the table allows it -- 8 KB of LDS counters per block in the bin kernel, tiles of a few hundred to a few
    // thousand points -- never more than kMaxBins.
    if (tiled_path) {
        const uint32_t max_bins = (uint32_t)(((uint64_t)cap_eff >> tshift) + 2);      // tiles the (bounded) cell table can make
        PCR_TRY(tiled.reserve((n_res + 16) * sizeof(float4)));
        PCR_TRY(bin_start.reserve((kMaxBins + 8) * sizeof(uint32_t)));
        PCR_TRY(tile_sq.reserve((kMaxBins + 8) * sizeof(unsigned long long)));
        tiled_shift = tshift;
        if (!bin_count.p) {
            PCR_TRY(bin_count.reserve(((size_t)kMaxBins + 64) * kBinStride * sizeof(uint32_t)));
            PCR_TRY(hipMemsetAsync(bin_count.p, 0, bin_count.cap, s));      // builds expect and leave the counters zeroed
        }
        const bool vec = (stride_floats % 4 == 0) && ((uintptr_t)d_pts % 16 == 0);
        int bin_per = kBinPerDefault;
        if (const char* e = dev_env("PCR_BIN_PER")) bin_per = atoi(e);      // (development: chunk size sweep)
        const size_t bin_chunk = (size_t)256 * bin_per;
        const int bin_blocks = (int)std::min<size_t>(4096, (n + bin_chunk - 1) / bin_chunk ? (n + bin_chunk - 1) / bin_chunk : 1);
        const int place_blocks = (int)std::min<size_t>(2048, (n + 1023) / 1024 ? (n + 1023) / 1024 : 1);
        // (a grid with many more cells than points: light tiles and heavy tiles by an instantiation each, see grid_tile_kernel)
        const bool sparse = split_sparse_tiles && cells_hint > 4 * (uint64_t)n + 65536;
        const uint64_t tiles_est = cells_hint ? (cells_hint >> tshift) + 1 : 0;
        // dense grids: eight points per thread (135 VGPRs, three waves per SIMD) while a tile holds ~1 000 points or fewer on average, sixteen
        // beyond (A/B: 1 M points in 1 464 tiles 48.4 -> 46.8 us with eight; 5 M and 10 M points are faster with sixteen)
        const bool per8 = !sparse && tiles_est && n / tiles_est <= 1024;
        // BINS (see the top of the tiled path): a tile whose fullest slab holds more than `hs` points is cut one step further in the layout this build
        // leaves -- three quarters of what a block of the tile pass holds in registers -- while the bins stay within nb_max.  pcr_params.index_no_hints:
        // no layout is ever used, so nothing is cut.
        // Measured (round 5, A/B on one box, profiles/r05_notes.md): at 1 M points hs = 2 048 or 3 072 takes the tile pass from 26.0 to 18.6 us and costs the bin pass
        // 1.5 us (more bins: more claims); 1 024 costs it 4 us.  At 10 M points (tiles of 8 192 cells, sixteen points per thread) cutting does NOT pay: the bin
        // pass's stores scatter over twice the bins (151 -> 186-235 us) and the tile pass, whose length there is blocks x latency / occupancy and not its heaviest
        // block, stays at 153-169 us: clouds whose tiles hold more than ~1 000 points on average are never cut.
        // (cut_sparse -- the voxel filter's index: a concatenation of raw key frames on the 0.5 m lattice is a sparse grid whose few tiles next to the vehicle's
        //  path hold tens of thousands of points each; one block sorted such a tile alone, 95 us of a 150 us build.  Its bins hold up to 4 096 points in registers.)
        uint32_t hs = per8 ? 2048u : (sparse && cut_sparse ? 3072u : 0u);
        if (const char* e = dev_env("PCR_BIN_SPLIT")) hs = (uint32_t)std::max(0, atoi(e));      // (development: 0 = tiles are never cut)
        if (hs == 0u) hs = 0xffffffffu;
        // bins a layout may hold: the tiles + two bins per `hs` points (a slab that was cut holds between hs / 2 and hs), never fewer than the tiles the table
        // can make, and never fewer than the layout in hand was planned for (the bin pass sizes its LDS by it)
        uint32_t nb_max = (uint32_t)std::min<uint64_t>((uint64_t)kMaxBins, std::max<uint64_t>((uint64_t)max_bins, (tiles_est ? tiles_est + 2 : (uint64_t)max_bins) + (hs != 0xffffffffu ? 2 * (uint64_t)n / hs : 0)));
        if (hs == 0xffffffffu && !(lay_ok && lay_cuts)) nb_max = max_bins;
        if (lay_ok && lay_shift == tshift) nb_max = std::max(nb_max, lay_nb_max);
        // Layout hint (see grid_bin_kernel<.., kPlace>): the previous build of this index left, next to its header, where each bin's
        // points may go; a build that reuses the header places by it and skips the placing pass.  PCR_INDEX_NO_LAYOUT=1 switches it off.
        const bool use_layout = reuse_header && lay_ok && lay_shift == tshift && lay_nb_max <= nb_max && dev_env("PCR_INDEX_NO_LAYOUT") == nullptr;
        for (int i = 0; i < 2; ++i)
            if (!layout[i].p) { PCR_TRY(layout[i].reserve(kLayWords * sizeof(uint32_t))); PCR_TRY(hipMemsetAsync(layout[i].p, 0, kLayWords * sizeof(uint32_t), s)); }      // (meta word 0: no layout)
        if (use_layout) {      // room for every bin's slack, the children of a tile that is cut one step further get their parent's room each:
                               // at most twice the cloud's (the device also checks every store against the capacity it is told)
            const size_t need = 2 * (lay_n + (lay_n >> lay_room_shift)) + (size_t)lay_room_add * (nb_max + 1) + 16;
            if (need > n + 16) PCR_TRY(tiled.reserve(need * sizeof(float4)));
        }
        const uint32_t tiled_cap = (uint32_t)std::min<size_t>(tiled.cap / sizeof(float4), 0xfffffff0u);
        const uint32_t* const lay_cur = layout[lay_idx].as<uint32_t>();
        uint32_t* const lay_next = layout[lay_idx ^ 1].as<uint32_t>();
        used_layout = use_layout;
        const uint8_t* keep_mask = nullptr;
        int keep_mshift = 0;
        if (filter && use_layout && filter->enqueue_mask) {      // the lattice is the reused header's, the rooms are the full cloud's: this build may leave points out
            PCR_TRY(filter->enqueue_mask());
            keep_mask = filter->mask; keep_mshift = filter->mshift;
            filter->applied = filtered = keep_mask != nullptr;
        }
        const bool cuts_now = hs != 0xffffffffu;                      // the layout this build plans may cut tiles
        const bool use_sub = use_layout && lay_cuts;                   // the layout in hand may hold tiles that are cut
        const bool plan_lds = cuts_now || use_sub;                     // block 0 of the tile pass plans with cuts (or merges some back): it keeps the bins' positions + the tiles' words in LDS
        // (sizing the bin pass's LDS by the tiles the reused header HAS instead of the tiles the table could make -- four blocks per CU instead of three at
        //  10 M points -- changed nothing: 300 / 306 us against 298 / 310, round 5.  At 10 M points the two passes move ~750 MB in 300 us: 2.5 TB/s of real traffic)
        const uint32_t nb_bin = use_layout ? nb_max : max_bins;      // bins the bin pass counts in LDS
        const size_t bin_lds = (size_t)nb_bin * 4 + (use_layout ? ((size_t)nb_bin + 4) * 4 : 0) + (use_sub ? ((size_t)max_bins + 8) * 2 : 0), place_lds = ((size_t)max_bins + 4) * 4,
                     tile_lds = std::max<size_t>((size_t)(1u << tshift) * 4, plan_lds ? ((size_t)nb_max + 4) * 4 + ((size_t)max_bins + 8) * 2 : 0);
#define PCR_LAUNCH_BIN(VEC, PER, PLACE, SUB) hipLaunchKernelGGL((grid_bin_kernel<VEC, PER, PLACE, SUB>), dim3(bin_blocks), dim3(256), bin_lds, s, d_pts, n32, st, header.as<GridHeader>(), \
                                                    bin_count.as<uint32_t>(), ranks.as<uint32_t>(), tshift, max_bins, nb_bin, ticket.as<uint32_t>() + 8, bin_start.as<uint32_t>(), \
                                                    lay_cur, tiled.as<float4>(), tiled_cap, keep_mask, keep_mshift)
#define PCR_LAUNCH_BIN_P(VEC, PLACE, SUB) do { if (bin_per == 4) PCR_LAUNCH_BIN(VEC, 4, PLACE, SUB); else if (bin_per == 8) PCR_LAUNCH_BIN(VEC, 8, PLACE, SUB); else PCR_LAUNCH_BIN(VEC, 16, PLACE, SUB); } while (0)
        // region-only builds of large clouds: the region's points are listed first, the bin pass reads the list (grid_keep_kernel; PCR_NO_KEEP_PASS=1 in a
        // development build: the bin pass tests the mask itself, as before)
        const bool keep_pass = keep_mask != nullptr && n >= 2000000 && dev_env("PCR_NO_KEEP_PASS") == nullptr;      // (a 1 M-point lattice build -- VGICP's -- is 1.5 % SLOWER with the extra launch)
        if (keep_pass) {
            PCR_TRY(kept.reserve((n + 16) * sizeof(float4)));
            uint32_t* const n_kept = ticket.as<uint32_t>() + 16;      // (zero between builds: the bin pass's last block puts it back)
            const int keep_blocks = (int)std::min<size_t>(1024, (n + 2047) / 2048);
            const uint32_t kept_cap = (uint32_t)std::min<size_t>(kept.cap / sizeof(float4), 0xfffffff0u);
            if (vec) hipLaunchKernelGGL(grid_keep_kernel<true>, dim3(keep_blocks), dim3(256), 0, s, d_pts, n32, st, header.as<GridHeader>(), keep_mask, keep_mshift, kept.as<float4>(), kept_cap, n_kept);
            else hipLaunchKernelGGL(grid_keep_kernel<false>, dim3(keep_blocks), dim3(256), 0, s, d_pts, n32, st, header.as<GridHeader>(), keep_mask, keep_mshift, kept.as<float4>(), kept_cap, n_kept);
            const int packed_blocks = std::min(bin_blocks, dev_env("PCR_PACKED_BLOCKS") ? atoi(dev_env("PCR_PACKED_BLOCKS")) : 512);      // (a block takes chunk after chunk: the list's length is the device's to know)
            const size_t lds8 = bin_lds;
#define PCR_LAUNCH_BIN_PACKED(SUB) hipLaunchKernelGGL((grid_bin_kernel<true, 8, true, SUB, true>), dim3(packed_blocks), dim3(256), lds8, s, kept.as<float>(), n32, 4u, header.as<GridHeader>(), \
                                                    bin_count.as<uint32_t>(), ranks.as<uint32_t>(), tshift, max_bins, nb_bin, ticket.as<uint32_t>() + 8, bin_start.as<uint32_t>(), \
                                                    lay_cur, tiled.as<float4>(), tiled_cap, (const uint8_t*)nullptr, 0, n_kept)
            if (use_sub) PCR_LAUNCH_BIN_PACKED(true); else PCR_LAUNCH_BIN_PACKED(false);
#undef PCR_LAUNCH_BIN_PACKED
        }
        else if (use_layout && use_sub) { if (vec) PCR_LAUNCH_BIN_P(true, true, true); else PCR_LAUNCH_BIN_P(false, true, true); }
        else if (use_layout) { if (vec) PCR_LAUNCH_BIN_P(true, true, false); else PCR_LAUNCH_BIN_P(false, true, false); }
        else { if (vec) PCR_LAUNCH_BIN_P(true, false, false); else PCR_LAUNCH_BIN_P(false, false, false); }
#undef PCR_LAUNCH_BIN_P
#undef PCR_LAUNCH_BIN
        if (!use_layout) {
            if (vec)
                hipLaunchKernelGGL(grid_place_kernel<true>, dim3(place_blocks), dim3(256), place_lds, s, d_pts, n32, st, header.as<GridHeader>(), bin_count.as<uint32_t>(),
                                   ranks.as<uint32_t>(), bin_start.as<uint32_t>(), tiled.as<float4>(), tshift);
            else
                hipLaunchKernelGGL(grid_place_kernel<false>, dim3(place_blocks), dim3(256), place_lds, s, d_pts, n32, st, header.as<GridHeader>(), bin_count.as<uint32_t>(),
                                   ranks.as<uint32_t>(), bin_start.as<uint32_t>(), tiled.as<float4>(), tshift);
        }
        TileTail tail;
        memset(&tail, 0, sizeof tail);
        // block 0 of the (first) tile launch plans the layout of the next build; the others take a bin each
        TilePlan plan;
        plan.lay_next = lay_next; plan.hs = hs; plan.nb_max = nb_max; plan.enabled = 1; plan.copy_only = filtered ? 1 : 0; plan.cuts = cuts_now ? 1 : 0; plan.room_shift = lay_room_shift; plan.room_add = lay_room_add; plan.pad_ = 0;
        // the tile pass: as many blocks as the device holds at once (a block takes bin after bin), + block 0
        static const int n_cu = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; return v; }();
        int per_cu = 0;      // (0 = a block per bin: persistent blocks -- 6 / 4 / 2 per CU by their registers -- measured no better at 1 M points and 6 % worse at 10 M, round 5)      // (blocks of 256 threads by their registers: 77 / 128 (130 with the tail) / 216 VGPRs)
        if (const char* e = dev_env("PCR_TILE_PER_CU")) per_cu = std::max(0, atoi(e));      // (development: 0 = a block per bin, as before round 5)
        // (a block per bin the host EXPECTS -- the tiles of the header it has seen + a bin per hs points when tiles are cut -- not per bin the table could
        //  make: a block that finds no bin still waits for the header, ~2 us of a slot each, and there were 3 000 of them at 10 M points; a block takes
        //  bin after bin, so more bins than blocks are served all the same)
        const uint32_t bins_bound = use_layout ? nb_max : max_bins;
        const uint32_t bins_expected = tiles_est ? (uint32_t)std::min<uint64_t>(bins_bound, tiles_est + 16 + (use_layout && hs != 0xffffffffu ? (uint64_t)n / hs : 0)) : bins_bound;
        const int tile_blocks = (int)std::min<uint32_t>(bins_expected, per_cu > 0 ? (uint32_t)(n_cu * per_cu) : bins_expected) + 1;
#define PCR_LAUNCH_TILE_R(PER, MODE, THREADS, TAIL, PLAN, RUNS) hipLaunchKernelGGL((grid_tile_kernel<PER, MODE, THREADS, TAIL, PLAN, RUNS>), dim3(tile_blocks), dim3(THREADS), (PLAN) ? tile_lds : (size_t)(1u << tshift) * 4, s, header.as<GridHeader>(), tile_sq.as<unsigned long long>(), \
                           bin_start.as<uint32_t>(), bin_count.as<uint32_t>(), tiled.as<float4>(), cell_start.as<uint32_t>(), sorted.as<float4>(), keys.as<uint32_t>(), tshift, \
                           use_layout ? lay_cur : (const uint32_t*)bin_start.as<uint32_t>(), use_layout ? lay_cur : (const uint32_t*)nullptr, plan, tail)
#define PCR_LAUNCH_TILE_T(PER, MODE, THREADS, TAIL, PLAN) PCR_LAUNCH_TILE_R(PER, MODE, THREADS, TAIL, PLAN, false)
#define PCR_LAUNCH_TILE(PER, MODE, THREADS) PCR_LAUNCH_TILE_T(PER, MODE, THREADS, false, true)
        // (blocks of 1 024 threads for the 5 M and 10 M-point maps -- grid_tile_kernel<4, 0, 1024>, PCR_TILE_WIDE in a development build -- cut the
        //  slowest tile of the 10 M-point map from 165 to 69 us and left the kernel at 165 us: one block per CU then, 19 rounds of ~8 us;
        //  profiles/r04_notes.md)
        static const int wide = dev_env("PCR_TILE_WIDE") ? atoi(dev_env("PCR_TILE_WIDE")) : 0;      // (development: average points per tile from which the wide blocks are used; 0 = never)
        // (the tile pass lists NDT's voxel cells beside its own work: dense grids, tiles of at most 2^13 cells -- 32 per thread)
        const bool with_tail = filtered && filter->want_tail && !sparse && tshift <= 13 && dev_env("PCR_NDT_NO_TAIL") == nullptr;
        if (with_tail) { tail = filter->tail; tail.mask = keep_mask; tail.mshift = keep_mshift; filter->tail_applied = true; }
        // (sparse grids: the launch of the light tiles carries no planning block -- its 58 registers are what lets eight of its blocks share a CU)
        // (the heavy bins of the voxel filter's clouds by blocks of 1 024 threads: what a block holds in registers is the same 4 096 points, a bin of 12 000 --
        //  three chunks, two sweeps each -- is through four times sooner)
        if (coherent_input && sparse) { PCR_LAUNCH_TILE_R(1, 1, 256, false, false, true); if (dev_env("PCR_VF_NARROW")) PCR_LAUNCH_TILE_R(16, 2, 256, false, true, true); else PCR_LAUNCH_TILE_R(4, 2, 1024, false, true, true); }
        else if (coherent_input && !with_tail) { if (per8) PCR_LAUNCH_TILE_R(8, 0, 256, false, true, true); else PCR_LAUNCH_TILE_R(16, 0, 256, false, true, true); }
        else if (sparse) { PCR_LAUNCH_TILE_T(1, 1, 256, false, false); PCR_LAUNCH_TILE(16, 2, 256); }
        else if (with_tail) { if (per8) PCR_LAUNCH_TILE_T(8, 0, 256, true, true); else PCR_LAUNCH_TILE_T(16, 0, 256, true, true); }
        else if (per8) PCR_LAUNCH_TILE(8, 0, 256);
        else if (wide > 0 && tiles_est && n / tiles_est >= (uint64_t)wide) PCR_LAUNCH_TILE(4, 0, 1024);
        else PCR_LAUNCH_TILE(16, 0, 256);
#undef PCR_LAUNCH_TILE
#undef PCR_LAUNCH_TILE_T
#undef PCR_LAUNCH_TILE_R
        lay_nb_max = nb_max; lay_cuts = cuts_now || use_sub;
        lay_idx ^= 1; lay_ok = true; lay_shift = tshift;      // (what this build's last block wrote serves the next one)
        if (!filtered) lay_n = n;
        PCR_TRY(hipGetLastError());
        n_points = n;
        valid = true;
        return hipSuccess;
    }
    tiled_shift = -1;
    hipLaunchKernelGGL(grid_count_kernel, dim3(pt_blocks), dim3(256), 0, s, d_pts, n32, st, header.as<GridHeader>(),
                       cell_count.as<uint32_t>(), keys.as<uint32_t>(), ranks.as<uint32_t>());
    const int scan_blocks = (int)((cell_capacity + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(grid_scan_local_kernel, dim3(scan_blocks), dim3(kScanBlock), 0, s, cell_count.as<uint32_t>(),
                       cell_start.as<uint32_t>(), block_sums.as<uint32_t>(), header.as<GridHeader>());
    hipLaunchKernelGGL(grid_scan_add_kernel, dim3(scan_blocks), dim3(kScanBlock), 0, s, cell_start.as<uint32_t>(),
                       block_sums.as<uint32_t>(), header.as<GridHeader>());
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(pt_blocks), dim3(256), 0, s, d_pts, n32, st, header.as<GridHeader>(),
                       keys.as<uint32_t>(), ranks.as<uint32_t>(), cell_start.as<uint32_t>(), sorted.as<float4>());
    PCR_TRY(hipGetLastError());
    n_points = n;
    valid = true;
    return hipSuccess;
}

#ifdef PCR_DEV_SWITCHES
// development builds: arm (allocate + clear) or read the stamp buffer of the bin / tile kernels
static unsigned long long* g_stamps_host_ptr = nullptr;
int dev_stamps(unsigned long long* out, size_t count) {
    const size_t total = (size_t)2 * 8192 * 8;
    if (!g_stamps_host_ptr) {
        if (hipMalloc((void**)&g_stamps_host_ptr, total * 8) != hipSuccess) return 1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_dev_stamps), &g_stamps_host_ptr, sizeof(g_stamps_host_ptr)) != hipSuccess) return 1;
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out && hipMemcpy(out, g_stamps_host_ptr, std::min(count, total) * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    return hipMemset(g_stamps_host_ptr, 0, total * 8) != hipSuccess;
}
#endif

}  // namespace pcr
