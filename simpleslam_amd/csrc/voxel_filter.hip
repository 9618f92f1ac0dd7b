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
#include "pcr_internal.h"

namespace pcr {

static constexpr int kVfTile = 2048;

// head[j] = 1 when sorted point j is the first of its voxel
__global__ __launch_bounds__(256) void voxel_heads_kernel(GridView g, uint32_t n_max, uint32_t* __restrict__ head) {
    const GridHeader h = *g.hdr;
    const uint32_t n = (h.empty || h.overflow) ? 0u : g.cell_start[h.n_cells];
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n_max; j += gridDim.x * 256) {
        uint32_t f = 0;
        if (j < n) {
            const float4 p = g.pts[j];
            const int ix = (int)(floorf(p.x * h.inv_leaf_f) - (float)h.min_b[0]), iy = (int)(floorf(p.y * h.inv_leaf_f) - (float)h.min_b[1]),
                      iz = (int)(floorf(p.z * h.inv_leaf_f) - (float)h.min_b[2]);
            const uint32_t key = ((uint32_t)iz * (uint32_t)h.dims[1] + (uint32_t)iy) * (uint32_t)h.dims[0] + (uint32_t)ix;
            f = g.cell_start[key] == j ? 1u : 0u;
        }
        head[j] = f;
    }
}

__device__ inline uint32_t vf_block_scan(uint32_t v, uint32_t* total, uint32_t* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) off += sh[w]; tot += sh[w]; }
    *total = tot;
    return off + inc - v;
}

// exclusive scan of head[0..n) in place (tile-local), tile totals -> sums
__global__ __launch_bounds__(256) void voxel_scan_local_kernel(uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t sh[4];
    const uint32_t base = blockIdx.x * kVfTile + threadIdx.x * 8;
    uint32_t x[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = base + i < n ? v[base + i] : 0u; s += x[i]; }
    uint32_t tot;
    uint32_t off = vf_block_scan(s, &tot, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (base + i < n) v[base + i] = off; off += x[i]; }
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// One thread per sorted point; the first point of a voxel folds the voxel's run and writes the centroid at the
// voxel's rank (tile offset + local exclusive scan).  The last block also reports the number of voxels.
__global__ __launch_bounds__(256) void voxel_centroid_kernel(GridView g, const float* __restrict__ orig, uint32_t stride, int intensity_at,
                                                             const uint32_t* __restrict__ rank_local, const uint32_t* __restrict__ sums,
                                                             uint32_t n_max, float* __restrict__ out, uint32_t out_capacity,
                                                             uint32_t* __restrict__ n_out) {
    __shared__ uint32_t sh_red[8];
    const GridHeader h = *g.hdr;
    const uint32_t n = (h.empty || h.overflow) ? 0u : g.cell_start[h.n_cells];
    const uint32_t tile = (blockIdx.x * 256u) / kVfTile;          // 8 blocks per scan tile
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
    if (last && threadIdx.x == 0) *n_out = sh_red[4] + sh_red[5] + sh_red[6] + sh_red[7];
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float4 p0 = g.pts[j];
    const int ix = (int)(floorf(p0.x * h.inv_leaf_f) - (float)h.min_b[0]), iy = (int)(floorf(p0.y * h.inv_leaf_f) - (float)h.min_b[1]),
              iz = (int)(floorf(p0.z * h.inv_leaf_f) - (float)h.min_b[2]);
    const uint32_t key = ((uint32_t)iz * (uint32_t)h.dims[1] + (uint32_t)iy) * (uint32_t)h.dims[0] + (uint32_t)ix;
    if (g.cell_start[key] != j) return;
    const uint32_t e = g.cell_start[key + 1];
    double sx = 0, sy = 0, sz = 0, si = 0;
    for (uint32_t i = j; i < e; ++i) {
        const float4 p = g.pts[i];
        sx += (double)p.x; sy += (double)p.y; sz += (double)p.z;
        if (intensity_at >= 0) si += (double)orig[(size_t)__float_as_uint(p.w) * stride + intensity_at];
    }
    const uint32_t pos = tile_off + rank_local[j];
    if (pos >= out_capacity) return;
    const double inv = 1.0 / (double)(e - j);
    float* o = out + (size_t)pos * stride;
    for (uint32_t c = 0; c < stride; ++c) o[c] = 0.f;
    o[0] = (float)(sx * inv); o[1] = (float)(sy * inv); o[2] = (float)(sz * inv);
    if (stride >= 8) o[3] = 1.0f;                                 // pcl::PointXYZI keeps data[3] = 1
    if (intensity_at >= 0) o[intensity_at] = (float)(si * inv);
}

hipError_t voxel_filter_launch(const GridIndex& grid, const float* d_orig, size_t stride_floats, size_t n, uint32_t* d_head, uint32_t* d_sums,
                               float* d_out, size_t out_capacity, uint32_t* d_n_out, hipStream_t s) {
    const uint32_t n32 = (uint32_t)n;
    const int blocks = (int)((n + 255) / 256 ? (n + 255) / 256 : 1);
    const int tiles = (int)((n + kVfTile - 1) / kVfTile ? (n + kVfTile - 1) / kVfTile : 1);
    const int intensity_at = stride_floats >= 8 ? 4 : (stride_floats >= 4 ? 3 : -1);
    hipLaunchKernelGGL(voxel_heads_kernel, dim3(std::min(blocks, 65535)), dim3(256), 0, s, grid.view(), n32, d_head);
    hipLaunchKernelGGL(voxel_scan_local_kernel, dim3(tiles), dim3(256), 0, s, d_head, n32, d_sums);
    hipLaunchKernelGGL(voxel_centroid_kernel, dim3(blocks), dim3(256), 0, s, grid.view(), d_orig, (uint32_t)stride_floats, intensity_at, d_head, d_sums,
                       n32, d_out, (uint32_t)std::min<size_t>(out_capacity, 0xffffffffu), d_n_out);
    return hipGetLastError();
}

}  // namespace pcr
