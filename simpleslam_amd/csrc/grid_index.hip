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
                                                               const ClampBox clamp) {
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
    if (threadIdx.x == 0) {
        *ticket = 0;                                           // ready for the next build
        GridHeader h;
        h.cell = cell; h.inv_cell = 1.0 / cell; h.n_points = n; h.shift = shift;
        h.empty = 0; h.overflow = 0;
        h.pcl_mode = pcl_mode; h.inv_leaf_f = 1.0f / (float)cell; h.too_fine = 0; h.sum_sq = 0.f;
        h.min_b[0] = h.min_b[1] = h.min_b[2] = 0;
        h.clamped = (clamp.use && !pcl_mode) ? 1 : 0; h.cut_mask = 0;
        double nc = 1.0;
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
                const float flo = floorf(lo * h.inv_leaf_f), fhi = floorf(hi * h.inv_leaf_f);
                const double dim = (double)fhi - (double)flo + 1.0;
                h.min_b[d] = fabsf(flo) < 2.0e9f ? (int32_t)flo : 0;
                h.org[d] = (double)flo; h.origin[d] = (double)flo * cell;
                h.dims[d] = dim < 2.0e9 ? (int32_t)dim : 0x7fffffff;
                nc *= dim;
                continue;
            }
            double clo = floor((double)lo / cell - shift), chi = floor((double)hi / cell - shift);
            h.org[d] = clo - kPad;
            h.origin[d] = (clo - kPad + shift) * cell;
            double dim = chi - clo + 1.0 + 2.0 * kPad;
            h.dims[d] = dim < 2.0e9 ? (int32_t)dim : 0x7fffffff;
            nc *= dim;
        }
        if (pcl_mode && nc > 2147483647.0) h.too_fine = 1;      // (dx*dy*dz) > INT_MAX
        // keys are uint32 and the table holds n_cells + 1 starts
        if (h.too_fine) { h.overflow = 0; h.empty = 1; h.n_cells = 1; }      // nothing is indexed; the caller copies its input
        else if (nc + 1.0 > (double)capacity || nc > 4.0e9) { h.overflow = 1; h.n_cells = nc < 1.8e19 ? (uint64_t)nc : ~0ull; }
        else h.n_cells = (uint64_t)h.dims[0] * (uint64_t)h.dims[1] * (uint64_t)h.dims[2];
        *hdr = h;
    }
}

__device__ inline bool point_key(const GridHeader& h, float x, float y, float z, uint32_t* key) {
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) return false;
    if (h.pcl_mode) {
        // ijk = static_cast<int>(std::floor(p * inverse_leaf_size) - static_cast<float>(min_b))   (voxel_grid.hpp)
        const int ix = (int)(floorf(x * h.inv_leaf_f) - (float)h.min_b[0]), iy = (int)(floorf(y * h.inv_leaf_f) - (float)h.min_b[1]),
                  iz = (int)(floorf(z * h.inv_leaf_f) - (float)h.min_b[2]);
        *key = ((uint32_t)iz * (uint32_t)h.dims[1] + (uint32_t)iy) * (uint32_t)h.dims[0] + (uint32_t)ix;
        return true;
    }
    // cell index = floor(x / cell) - org.  For a power-of-two cell (LOAM) x / cell is exact and this
    // equals floor((x - origin) / cell); for any other edge (VGICP/NDT resolutions) it is the single
    // definition every kernel uses, so a point and its queries always agree on the cell.
    const double fx = floor((double)x / h.cell - h.shift) - h.org[0];
    const double fy = floor((double)y / h.cell - h.shift) - h.org[1];
    const double fz = floor((double)z / h.cell - h.shift) - h.org[2];
    if (h.clamped && !(fx >= (double)kPad && fx < (double)(h.dims[0] - kPad) && fy >= (double)kPad && fy < (double)(h.dims[1] - kPad) &&
                       fz >= (double)kPad && fz < (double)(h.dims[2] - kPad))) return false;      // outside the region of interest
    const uint32_t cx = (uint32_t)fx, cy = (uint32_t)fy, cz = (uint32_t)fz;
    *key = (cz * (uint32_t)h.dims[1] + cy) * (uint32_t)h.dims[0] + cx;
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

// ---- host side ----------------------------------------------------------------------
hipError_t DeviceBuf::reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    size_t want = bytes + bytes / 4 + 256;
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
    cell_capacity = 0; valid = false; n_points = 0;
}

hipError_t GridIndex::grow_cells(uint64_t need_cells, std::string* err) {
    if (need_cells > 4000000000ull) {
        if (err) *err = "target bounding box needs " + std::to_string(need_cells) + " grid cells (> 4e9): cloud too sparse for the dense index";
        return hipErrorInvalidValue;
    }
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

hipError_t GridIndex::build(const float* d_pts, size_t n, size_t stride_floats, double cell, hipStream_t s, std::string* err, double shift,
                            int pcl_mode, const ClampBox* clamp) {
    valid = false;
    if (n > 0xfffffff0ull) { if (err) *err = "target cloud too large (>= 2^32 points)"; return hipErrorInvalidValue; }
    PCR_TRY(sorted.reserve((n + 16) * sizeof(float4)));   // padded: the search reads whole chunks
    PCR_TRY(bbox_partials.reserve(kBBoxBlocks * 6 * sizeof(float)));
    PCR_TRY(header.reserve(sizeof(GridHeader)));
    PCR_TRY(keys.reserve((n + 1) * sizeof(uint32_t)));
    PCR_TRY(ranks.reserve((n + 1) * sizeof(uint32_t)));
    if (!ticket.p) {
        PCR_TRY(ticket.reserve(256));
        PCR_TRY(hipMemsetAsync(ticket.p, 0, 256, s));
    }
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
    ClampBox cb;
    memset(&cb, 0, sizeof cb);
    if (clamp) cb = *clamp;
    const uint32_t n32 = (uint32_t)n, st = (uint32_t)stride_floats;
    const int pt_blocks = (int)std::min<size_t>(2048, (n + 255) / 256 ? (n + 255) / 256 : 1);
    hipLaunchKernelGGL(grid_bbox_header_kernel, dim3(kBBoxBlocks), dim3(256), 0, s, d_pts, n32, st, bbox_partials.as<float>(),
                       ticket.as<uint32_t>(), header.as<GridHeader>(), (uint64_t)cell_capacity, cell, shift, pcl_mode, cb);
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

}  // namespace pcr
