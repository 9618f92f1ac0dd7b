// voxel_filter.hip -- pcl::VoxelGrid<PointXYZI>::filter on the device (gfx950).
//
// The step right before the registration path: the front-end voxel-filters every scan
// (frontend/src/LidarOdometry.cpp:36,170-171) and the map manager every rebuilt sub-map
// (frontend/src/MapManager.cpp:78,192 -> common/pcp/pcp.hpp:14-28).  pcl::VoxelGrid itself is not in the reference
// tree (PCL is an external dependency, version unpinned: SURVEY.md 8(c)); restated from PCL 1.10
// filters/include/pcl/filters/impl/voxel_grid.hpp (applyFilter):
//   min_b = floor(min_p * inv_leaf), div_b = max_b - min_b + 1            (float arithmetic)
//   idx   = ijk0 + ijk1 * div_b[0] + ijk2 * div_b[0] * div_b[1],  ijk = (int)(floor(p * inv_leaf) - (float)min_b)
//   output = one centroid (all fields averaged: downsample_all_data_ = true) per occupied voxel, ascending idx.
// The lattice is exactly PCL's (GridHeader.pcl_mode); the sums run in double precision (PCL: float, in an order its
// unstable sort leaves unspecified), so the centroids agree with PCL's to its own float rounding, n * eps * |x|.
//
// Round 5: two launches behind the index build, and a voxel's points are summed by the waves that hold them, not by one thread.
// A sub-map assembled from raw key frames (MapManager.cpp:151-201) puts hundreds to thousands of points into the voxels next to the
// sensor's path: one thread walking such a run -- a dependent load and a gather per point -- made the launch as long as its longest voxel
// (159 us for a 500 k-point concatenation, scripts/seq_breakdown.py).  Now, per wave of 64 sorted points:
//   lead[w] = the sum of the points in front of the wave's first voxel start (they belong to a voxel that started in an earlier wave),
//   a backward segmented scan gives every voxel start the sum of its points INSIDE its wave,
//   and the start adds lead[w] of the waves its run reaches into: 32 bytes per 64 points instead of a round trip per point.
// The order of the additions is a function of the sorted array alone: two runs over the same index give the same bits.
#include "pcr_internal.h"

namespace pcr {

static constexpr int kVfTile = 2048;      // points per block of the marking pass: 8 rounds of 256

struct VfPoint { double x, y, z, w; };

__device__ inline uint32_t vf_key(const GridHeader& h, const float4 p) {
    const int ix = (int)(floorf(p.x * h.inv_leaf_f) - (float)h.min_b[0]), iy = (int)(floorf(p.y * h.inv_leaf_f) - (float)h.min_b[1]),
              iz = (int)(floorf(p.z * h.inv_leaf_f) - (float)h.min_b[2]);
    return ((uint32_t)iz * (uint32_t)h.dims[1] + (uint32_t)iy) * (uint32_t)h.dims[0] + (uint32_t)ix;
}

// sorted point j starts a voxel: its key differs from its predecessor's (the index sorts by key)
__device__ inline bool vf_is_head(const GridHeader& h, const float4* __restrict__ pts, uint32_t j, bool valid, uint32_t key) {
    const int lane = threadIdx.x & 63;
    uint32_t prev = __shfl_up(key, 1, 64);
    if (lane == 0 && valid && j > 0) prev = vf_key(h, pts[j - 1]);
    return valid && (j == 0 || prev != key);
}

// Marking pass, a block per 2048 sorted points: intensity of every point (gathered through the index's original-point number, stored in
// sorted order), per wave of 64 points the sum in front of its first voxel start (lead) and the voxel starts before it within the block's
// tile (wave_rank), per tile the number of starts (sums).
__global__ __launch_bounds__(256) void voxel_mark_kernel(GridView g, const float* __restrict__ orig, uint32_t stride, int intensity_at, uint32_t n_max,
                                                         float* __restrict__ inten, VfPoint* __restrict__ lead, uint32_t* __restrict__ wave_rank,
                                                         uint32_t* __restrict__ sums) {
    __shared__ uint32_t sh_cnt[32];
    const GridHeader h = *g.hdr;
    const uint32_t n = (h.empty || h.overflow || h.stale) ? 0u : min(g.cell_start[h.n_cells], n_max);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * kVfTile;
    // (all eight rounds' points requested at once, then their intensities: two round trips for the block instead of sixteen -- 14 -> 5 us for a 65 536-point scan, whose 32 blocks are a chain of latencies)
    float4 p[8], q[8];      // the points and their predecessors in the sorted order (unconditional loads of clamped indices: conditional ones are issued one by one)
    float it[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { p[r] = make_float4(0.f, 0.f, 0.f, 0.f); q[r] = p[r]; it[r] = 0.f; }
    if (n) {      // (block-uniform)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t j = base + r * 256 + threadIdx.x, jc = j < n ? j : n - 1u;
            p[r] = g.pts[jc]; q[r] = g.pts[jc ? jc - 1u : 0u];
        }
        if (intensity_at >= 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) it[r] = orig[(size_t)__float_as_uint(p[r].w) * stride + intensity_at];
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint32_t j = base + r * 256 + threadIdx.x;
        const bool valid = j < n;
        const uint32_t key = valid ? vf_key(h, p[r]) : 0u;
        if (valid) inten[j] = it[r];
        const bool head = valid && (j == 0u || vf_key(h, q[r]) != key);
        const unsigned long long bound = __ballot(head || !valid);
        const int first = bound ? __builtin_ctzll(bound) : 64;
        const bool in = lane < first;      // (in front of the first start: valid by construction)
        double sx = in ? (double)p[r].x : 0.0, sy = in ? (double)p[r].y : 0.0, sz = in ? (double)p[r].z : 0.0, sw = in ? (double)it[r] : 0.0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { sx += __shfl_xor(sx, m, 64); sy += __shfl_xor(sy, m, 64); sz += __shfl_xor(sz, m, 64); sw += __shfl_xor(sw, m, 64); }
        const uint32_t w = (base >> 6) + r * 4 + wave;
        if (lane == 0 && (size_t)w * 64 < (size_t)n_max + 64) { VfPoint a; a.x = sx; a.y = sy; a.z = sz; a.w = sw; lead[w] = a; }
        const unsigned long long heads = __ballot(head);      // (every lane votes)
        if (lane == 0) sh_cnt[r * 4 + wave] = (uint32_t)__popcll(heads);
    }
    __syncthreads();
    if (threadIdx.x < 64) {      // exclusive scan of the 32 wave counts (wave w of the tile holds points base + 64 w ..)
        const uint32_t c = lane < 32 ? sh_cnt[lane] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane < 32 && (size_t)((base >> 6) + lane) * 64 < (size_t)n_max + 64) wave_rank[(base >> 6) + lane] = inc - c;
        if (lane == 31) sums[blockIdx.x] = inc;
    }
}

// Centroid pass, a thread per sorted point: the start of a voxel gets the sum of its run -- its wave's part by a backward segmented scan, the
// rest from the waves' leads -- and writes the centroid at the voxel's rank (tile offset + wave rank + starts before it in the wave).  The last
// block reports the number of voxels and what the header says, straight into the caller's page-locked result (no copy behind the launch).
__global__ __launch_bounds__(256) void voxel_centroid_kernel(GridView g, uint32_t stride, int intensity_at, const float* __restrict__ inten,
                                                             const VfPoint* __restrict__ lead, const uint32_t* __restrict__ wave_rank,
                                                             const uint32_t* __restrict__ sums, uint32_t n_max, float* __restrict__ out,
                                                             uint32_t out_capacity, VfResult* __restrict__ result) {
    __shared__ uint32_t sh_red[8];
    const GridHeader h = *g.hdr;
    const uint32_t n = (h.empty || h.overflow || h.stale) ? 0u : min(g.cell_start[h.n_cells], n_max);
    const uint32_t tile = (blockIdx.x * 256u) / kVfTile;          // 8 blocks per tile of the marking pass
    const uint32_t tiles = (n_max + kVfTile - 1) / kVfTile;
    const bool last = blockIdx.x == gridDim.x - 1;
    // offset of this block's tile = sum of the totals of the tiles before it; the last block also adds up all of them
    uint32_t part = 0, all = 0;
    for (uint32_t t = threadIdx.x; t < (last ? tiles : tile); t += 256) { const uint32_t v = sums[t]; all += v; part += t < tile ? v : 0u; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { part += __shfl_xor(part, m, 64); all += __shfl_xor(all, m, 64); }
    if ((threadIdx.x & 63) == 0) { sh_red[threadIdx.x >> 6] = part; sh_red[4 + (threadIdx.x >> 6)] = all; }
    __syncthreads();
    const uint32_t tile_off = sh_red[0] + sh_red[1] + sh_red[2] + sh_red[3];
    if (last && threadIdx.x == 0) {
        VfResult r;
        r.count = sh_red[4] + sh_red[5] + sh_red[6] + sh_red[7];
        r.overflow = h.overflow; r.too_fine = h.too_fine; r.stale = h.stale; r.empty = h.empty; r.pad_ = 0; r.n_cells = h.n_cells;
        *result = r;
    }
    const int lane = threadIdx.x & 63;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if ((j & ~63u) >= n) return;                                  // (wave-uniform)
    const bool valid = j < n;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t key = 0;
    float it = 0.f;
    if (valid) { p = g.pts[j]; key = vf_key(h, p); if (intensity_at >= 0) it = inten[j]; }
    const bool head = vf_is_head(h, g.pts, j, valid, key);
    const unsigned long long heads = __ballot(head), bound = __ballot(head || !valid);
    const unsigned long long above = lane == 63 ? 0ull : (bound & ~((2ull << lane) - 1ull));
    const int nxt = above ? __builtin_ctzll(above) : 64;         // the next start (or the cloud's end) in this wave
    double sx = (double)p.x, sy = (double)p.y, sz = (double)p.z, sw = (double)it;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {                            // lane l <- sum over [l, min(l + 2 d, nxt))
        const double tx = __shfl_down(sx, d, 64), ty = __shfl_down(sy, d, 64), tz = __shfl_down(sz, d, 64), tw = __shfl_down(sw, d, 64);
        if (lane + d < nxt) { sx += tx; sy += ty; sz += tz; sw += tw; }
    }
    if (!head) return;
    const uint32_t e = min(g.cell_start[key + 1], n);             // end of the voxel's run
    const uint32_t wave_end = (j | 63u) + 1u;
    if (e > wave_end)
        for (uint32_t w = wave_end >> 6; w <= ((e - 1u) >> 6); ++w) { const VfPoint a = lead[w]; sx += a.x; sy += a.y; sz += a.z; sw += a.w; }
    const uint32_t pos = tile_off + wave_rank[j >> 6] + (uint32_t)__popcll(heads & ((1ull << lane) - 1ull));
    if (pos >= out_capacity) return;
    const double inv = 1.0 / (double)(e - j);
    float* o = out + (size_t)pos * stride;
    for (uint32_t c = 0; c < stride; ++c) o[c] = 0.f;
    o[0] = (float)(sx * inv); o[1] = (float)(sy * inv); o[2] = (float)(sz * inv);
    if (stride >= 8) o[3] = 1.0f;                                 // pcl::PointXYZI keeps data[3] = 1
    if (intensity_at >= 0) o[intensity_at] = (float)(sw * inv);
}

size_t voxel_filter_wave_bytes(size_t n) { return (n / 64 + 40) * (sizeof(VfPoint) + sizeof(uint32_t)); }

hipError_t voxel_filter_launch(const GridIndex& grid, const float* d_orig, size_t stride_floats, size_t n, uint32_t* d_inten, uint32_t* d_sums,
                               void* d_wave, float* d_out, size_t out_capacity, void* result_mapped, hipStream_t s) {
    const uint32_t n32 = (uint32_t)n;
    const int blocks = (int)((n + 255) / 256 ? (n + 255) / 256 : 1);
    const int tiles = (int)((n + kVfTile - 1) / kVfTile ? (n + kVfTile - 1) / kVfTile : 1);
    const int intensity_at = stride_floats >= 8 ? 4 : (stride_floats >= 4 ? 3 : -1);
    VfPoint* lead = static_cast<VfPoint*>(d_wave);
    uint32_t* wave_rank = reinterpret_cast<uint32_t*>(lead + (n / 64 + 40));
    hipLaunchKernelGGL(voxel_mark_kernel, dim3(tiles), dim3(256), 0, s, grid.view(), d_orig, (uint32_t)stride_floats, intensity_at, n32,
                       reinterpret_cast<float*>(d_inten), lead, wave_rank, d_sums);
    hipLaunchKernelGGL(voxel_centroid_kernel, dim3(blocks), dim3(256), 0, s, grid.view(), (uint32_t)stride_floats, intensity_at,
                       reinterpret_cast<const float*>(d_inten), lead, wave_rank, d_sums, n32, d_out,
                       (uint32_t)std::min<size_t>(out_capacity, 0xffffffffu), static_cast<VfResult*>(result_mapped));
    return hipGetLastError();
}

}  // namespace pcr
