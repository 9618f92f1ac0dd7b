// query_sort.hip -- group the scan points by the target-grid cell they fall into (gfx950).
//
// The reference walks the scan in cloud order with one OpenMP thread per chunk
// (reference PCR/src/LoamRegister.cpp:122-123); on a 64-wide wavefront that order makes
// neighbouring lanes search unrelated cells: every lane gathers its own cache lines and the
// rare-per-lane top-5 insertion runs for the whole wave at almost every candidate.  Sorting
// the scan ONCE per scan2Map call by the cell key of its initially transformed position puts
// lanes that share a cell (17 per cell on the 64-beam benchmark scan) next to each other, so
// their candidate reads coalesce into the same lines and their control flow stays together.
//
// A stable LSD radix sort (11-bit digits, hand-written: per-wave LDS histograms, one-block
// scan, ballot-based stable ranks) keeps the order deterministic, which keeps the fixed-order
// reduction of the normal equations bitwise reproducible.  The result does not depend on the
// order at all beyond floating-point summation order.
#include "pcr_internal.h"

namespace pcr {

static constexpr int kSortTile = 4096;      // elements per 256-thread block (1024 per wave)
static constexpr int kMaxDigitBits = 11;

// ---- keys ------------------------------------------------------------------------
// Same arithmetic as loam_point (loam.hip): f64 transform, cast to f32, exact cell coordinates.
__global__ __launch_bounds__(256) void qsort_keys_kernel(const float* __restrict__ src, uint32_t n, uint32_t stride,
                                                         const Pose16 pose16, GridView grid,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                         uint32_t invalid_key) {
    const GridHeader h = *grid.hdr;
    double pose[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pose[i] = pose16.m[i];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float* sp = src + (size_t)i * stride;
        const double ox = (double)sp[0], oy = (double)sp[1], oz = (double)sp[2];
        const float px = (float)(pose[0] * ox + pose[4] * oy + pose[8] * oz + pose[12] * 1.0);
        const float py = (float)(pose[1] * ox + pose[5] * oy + pose[9] * oz + pose[13] * 1.0);
        const float pz = (float)(pose[2] * ox + pose[6] * oy + pose[10] * oz + pose[14] * 1.0);
        const double fx = floor(((double)px - h.origin[0]) * h.inv_cell);
        const double fy = floor(((double)py - h.origin[1]) * h.inv_cell);
        const double fz = floor(((double)pz - h.origin[2]) * h.inv_cell);
        uint32_t key = invalid_key;
        if (!h.overflow && !h.empty && fx >= 0.0 && fx < (double)h.dims[0] && fy >= 0.0 && fy < (double)h.dims[1] && fz >= 0.0 &&
            fz < (double)h.dims[2])
            key = ((uint32_t)fz * (uint32_t)h.dims[1] + (uint32_t)fy) * (uint32_t)h.dims[0] + (uint32_t)fx;
        if (key > invalid_key) key = invalid_key;
        keys[i] = key;
        vals[i] = i;
    }
}

// ---- one radix pass: histogram, scan, stable scatter -------------------------------
// counts layout: [bin][sub] with sub = block * 4 + wave (each wave owns 1024 consecutive elements)
__global__ __launch_bounds__(256) void qsort_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift, int bits,
                                                         uint32_t* __restrict__ counts, uint32_t nsub) {
    extern __shared__ uint32_t sh_hist[];   // [4][nbins]
    const uint32_t nbins = 1u << bits, mask = nbins - 1;
    for (uint32_t i = threadIdx.x; i < 4 * nbins; i += 256) sh_hist[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t base = blockIdx.x * kSortTile + wave * 1024;
    uint32_t* hist = sh_hist + wave * nbins;
    for (int r = 0; r < 16; ++r) {
        const uint32_t i = base + r * 64 + lane;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4 * nbins; i += 256) {
        const uint32_t w = i / nbins, b = i - w * nbins;
        counts[(size_t)b * nsub + blockIdx.x * 4 + w] = sh_hist[i];
    }
}

// exclusive scan of m counters by one 1024-thread block
__global__ __launch_bounds__(1024) void qsort_scan_kernel(uint32_t* __restrict__ counts, uint32_t m) {
    __shared__ uint32_t sh_w[16];
    const uint32_t per = (m + 1023) / 1024;
    const uint32_t lo = min(threadIdx.x * per, m), hi = min(lo + per, m);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += counts[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) sh_w[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += sh_w[w];
    uint32_t run = off + inc - s;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t t = counts[i]; counts[i] = run; run += t; }
}

__global__ __launch_bounds__(256) void qsort_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            uint32_t n, int shift, int bits, const uint32_t* __restrict__ offsets,
                                                            uint32_t nsub) {
    extern __shared__ uint32_t sh_next[];   // [4][nbins]: next output slot of every digit, per wave
    const uint32_t nbins = 1u << bits, mask = nbins - 1;
    for (uint32_t i = threadIdx.x; i < 4 * nbins; i += 256) {
        const uint32_t w = i / nbins, b = i - w * nbins;
        sh_next[i] = offsets[(size_t)b * nsub + blockIdx.x * 4 + w];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t base = blockIdx.x * kSortTile + wave * 1024;
    uint32_t* next = sh_next + wave * nbins;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < 16; ++r) {
        const uint32_t i = base + r * 64 + lane;
        const bool valid = i < n;
        uint32_t key = 0, val = 0, d = 0;
        if (valid) { key = keys_in[i]; val = vals_in[i]; d = (key >> shift) & mask; }
        // lanes holding the same digit (stable multisplit by ballots over the digit's bits)
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        if (valid) {
            const uint32_t rank = __popcll(same & lt);
            const uint32_t dst = next[d] + rank;          // read by every lane of the group ...
            keys_out[dst] = key; vals_out[dst] = val;
            if (rank == 0) next[d] += __popcll(same);     // ... before its leader advances the slot (same wave: in order)
        }
    }
}

// ---- gather the scan into sorted order: float4(x, y, z, original index) ----------------
__global__ __launch_bounds__(256) void qsort_gather_kernel(const float* __restrict__ src, uint32_t n, uint32_t stride,
                                                           const uint32_t* __restrict__ vals, float4* __restrict__ out) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const uint32_t i = vals[j];
        const float* sp = src + (size_t)i * stride;
        out[j] = make_float4(sp[0], sp[1], sp[2], __uint_as_float(i));
    }
}

#define QS_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { if (err) *err = std::string(#x) + ": " + hipGetErrorString(_e); return _e; } } while (0)

hipError_t QuerySorter::sort(const float* d_src, size_t n, size_t stride_floats, const Pose16& d_pose16, const GridIndex& grid,
                             hipStream_t s, std::string* err) {
    const uint32_t n32 = (uint32_t)n;
    const uint32_t nblocks = (uint32_t)((n + kSortTile - 1) / kSortTile ? (n + kSortTile - 1) / kSortTile : 1);
    const uint32_t nsub = nblocks * 4;
    // key range: [0, cell_capacity] (the device-side n_cells is <= the host-side capacity)
    uint64_t maxkey = grid.cell_capacity;
    if (maxkey > 0xfffffffeull) maxkey = 0xfffffffeull;
    int total_bits = 1;
    while ((1ull << total_bits) <= maxkey) ++total_bits;
    const int passes = (total_bits + kMaxDigitBits - 1) / kMaxDigitBits;
    const int bits = (total_bits + passes - 1) / passes;
    const uint32_t nbins = 1u << bits;
    QS_TRY(keys[0].reserve((n + 1) * 4)); QS_TRY(keys[1].reserve((n + 1) * 4));
    QS_TRY(vals[0].reserve((n + 1) * 4)); QS_TRY(vals[1].reserve((n + 1) * 4));
    QS_TRY(counts.reserve((size_t)nbins * nsub * 4 + 16));
    QS_TRY(sorted.reserve((n + 1) * sizeof(float4)));
    const int blocks = (int)std::min<size_t>(1024, (n + 255) / 256 ? (n + 255) / 256 : 1);
    hipLaunchKernelGGL(qsort_keys_kernel, dim3(blocks), dim3(256), 0, s, d_src, n32, (uint32_t)stride_floats, d_pose16, grid.view(),
                       keys[0].as<uint32_t>(), vals[0].as<uint32_t>(), (uint32_t)maxkey);
    int cur = 0;
    const size_t lds = (size_t)4 * nbins * sizeof(uint32_t);
    for (int p = 0; p < passes; ++p) {
        const int shift = p * bits;
        hipLaunchKernelGGL(qsort_hist_kernel, dim3(nblocks), dim3(256), lds, s, keys[cur].as<uint32_t>(), n32, shift, bits,
                           counts.as<uint32_t>(), nsub);
        hipLaunchKernelGGL(qsort_scan_kernel, dim3(1), dim3(1024), 0, s, counts.as<uint32_t>(), nbins * nsub);
        hipLaunchKernelGGL(qsort_scatter_kernel, dim3(nblocks), dim3(256), lds, s, keys[cur].as<uint32_t>(), vals[cur].as<uint32_t>(),
                           keys[cur ^ 1].as<uint32_t>(), vals[cur ^ 1].as<uint32_t>(), n32, shift, bits, counts.as<uint32_t>(), nsub);
        cur ^= 1;
    }
    hipLaunchKernelGGL(qsort_gather_kernel, dim3(blocks), dim3(256), 0, s, d_src, n32, (uint32_t)stride_floats,
                       vals[cur].as<uint32_t>(), sorted.as<float4>());
    QS_TRY(hipGetLastError());
    return hipSuccess;
}

void QuerySorter::release() {
    keys[0].release(); keys[1].release(); vals[0].release(); vals[1].release(); counts.release(); sorted.release();
}

}  // namespace pcr
