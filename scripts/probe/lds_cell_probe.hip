// probe (round 3): does staging a crowded target cell's 27-cell candidate set in LDS pay for the LOAM search?
//
// The design north_star sketched -- scan points bucketed by target cell, a block per cell, the union of its cell runs staged in LDS --
// was rejected in round 2 on an estimate.  This program measures it on the benchmark's own geometry, for the half of the queries where it
// can pay at all: the queries that sit in target cells holding >= 64 queries (ground rings near the sensor).  Two kernels with the SAME
// per-candidate work (float distance, 8 smallest 32-bit keys kept by v_min_u32 / v_max_u32, as loam.hip: knn_scan):
//   mode 0  per-lane stream: every lane walks the nine row runs of its own query's 3 x 3 x 3 block from global memory, eight candidates in
//           flight (what loam_iterate_kernel does, without its exchange, decode and proof)
//   mode 1  staged: one block per (chunk of <= 256 queries of one) crowded cell; the block copies the nine row runs of that cell's block
//           into LDS once (coalesced 16-byte loads), then every lane walks ALL of them from LDS
// Driven by scripts/lds_cell_probe.py (index and query order are built there, with numpy); times are HIP events around `reps` launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

static constexpr int kNb = 8;
static constexpr uint32_t kEmpty = 0xffffffffu;
static constexpr int kStageMax = 3072;      // float4 candidates staged per round: 48 KB of LDS

struct Geo { double org[3]; double inv_cell; int dims[3]; };

__device__ __forceinline__ void insert8(uint32_t key[kNb], uint32_t t) {
#pragma unroll
    for (int k = 0; k < kNb; ++k) { const uint32_t lo = min(key[k], t), hi = max(key[k], t); key[k] = lo; t = hi; }
}

__device__ __forceinline__ uint32_t make_key(float qx, float qy, float qz, float4 p, uint32_t seq) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    float d = dx * dx;
    d = __builtin_fmaf(dy, dy, d);
    d = __builtin_fmaf(dz, dz, d);
    return d <= 1.00001f ? ((__float_as_uint(d) & ~0xfffu) | (seq & 0xfffu)) : kEmpty;
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start, const Geo g,
                                                     const float4* __restrict__ q, uint32_t nq, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const float4 qq = q[i];
    const int cx = (int)floor(((double)qq.x - g.org[0]) * g.inv_cell), cy = (int)floor(((double)qq.y - g.org[1]) * g.inv_cell),
              cz = (int)floor(((double)qq.z - g.org[2]) * g.inv_cell);
    uint32_t ra[9], rb[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const uint32_t ck = ((uint32_t)(cz + r / 3 - 1) * (uint32_t)g.dims[1] + (uint32_t)(cy + r % 3 - 1)) * (uint32_t)g.dims[0] + (uint32_t)cx;
        ra[r] = cell_start[ck - 1]; rb[r] = cell_start[ck + 2];
    }
    uint32_t key[kNb];
#pragma unroll
    for (int k = 0; k < kNb; ++k) key[k] = kEmpty;
    uint32_t seq = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        for (uint32_t j = ra[r]; j < rb[r]; j += 8) {
            float4 c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = pts[j + u < rb[r] ? j + u : j];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t t = j + u < rb[r] ? make_key(qq.x, qq.y, qq.z, c[u], seq + u) : kEmpty; insert8(key, t); }
            seq += 8;
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kNb; ++k) s += key[k] == kEmpty ? 0u : (key[k] >> 12);
    out[i] = s;
}

// blk_first[b], blk_count[b]: the queries of block b (all in ONE target cell, blk_cell[b] = its linear key)
__global__ __launch_bounds__(256) void staged_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start, const Geo g,
                                                     const float4* __restrict__ q, const uint32_t* __restrict__ blk_first, const uint32_t* __restrict__ blk_count,
                                                     const uint32_t* __restrict__ blk_cell, uint32_t* __restrict__ out) {
    __shared__ float4 sh[kStageMax];
    __shared__ uint32_t sh_ra[9], sh_rb[9], sh_off[10];
    const uint32_t b = blockIdx.x, first = blk_first[b], cnt = blk_count[b], ck0 = blk_cell[b];
    if (threadIdx.x < 9) {
        const int r = threadIdx.x;
        const uint32_t d0 = (uint32_t)g.dims[0], d1 = (uint32_t)g.dims[1];
        const uint32_t row0 = ck0 / d0, cx = ck0 - row0 * d0, cz = row0 / d1, cy = row0 - cz * d1;
        const uint32_t ck = ((cz + r / 3 - 1) * d1 + (cy + r % 3 - 1)) * d0 + cx;
        sh_ra[r] = cell_start[ck - 1]; sh_rb[r] = cell_start[ck + 2];
    }
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (int r = 0; r < 9; ++r) { sh_off[r] = acc; acc += sh_rb[r] - sh_ra[r]; } sh_off[9] = acc; }
    __syncthreads();
    const uint32_t total = sh_off[9];
    const bool mine = threadIdx.x < cnt;
    const float4 qq = mine ? q[first + threadIdx.x] : make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t key[kNb];
#pragma unroll
    for (int k = 0; k < kNb; ++k) key[k] = kEmpty;
    for (uint32_t base = 0; base < total; base += kStageMax) {
        const uint32_t n = min((uint32_t)kStageMax, total - base);
        for (uint32_t t = threadIdx.x; t < n; t += 256) {      // flat index -> (run, offset): nine compares
            const uint32_t f = base + t;
            uint32_t src = 0;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (f >= sh_off[r]) src = sh_ra[r] + (f - sh_off[r]);
            sh[t] = pts[src];
        }
        __syncthreads();
        if (mine) {
            for (uint32_t j = 0; j < n; j += 8) {      // every lane reads the same address: LDS broadcast
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t t = j + u < n ? make_key(qq.x, qq.y, qq.z, sh[j + u < n ? j + u : j], base + j + u) : kEmpty;
                    insert8(key, t);
                }
            }
        }
        __syncthreads();
    }
    if (mine) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kNb; ++k) s += key[k] == kEmpty ? 0u : (key[k] >> 12);
        out[first + threadIdx.x] = s;
    }
}

// mode 2 / 3 (round 4): kLpq = 4 / 2 ADJACENT lanes per query.  Each lane takes the groups of four consecutive candidates of
// every run in turn (lane p: candidates 4 (p + kLpq i) .. + 3), keeps its own eight smallest keys; the lists of a query's lanes are merged in
// registers through DPP (partner's list reversed, min -> the eight smallest of the two as a sequence that rises and falls, sorted by a bitonic
// merge network of 12 comparators).  Same keys, same result word as stream_kernel (the sum over the eight listed sequence numbers' buckets).
template <int kCtrl> __device__ __forceinline__ uint32_t quad_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kCtrl, 0xf, 0xf, false); }
template <int kCtrl>
__device__ __forceinline__ void merge8(uint32_t key[kNb]) {
    uint32_t c[kNb];
#pragma unroll
    for (int i = 0; i < kNb; ++i) c[i] = min(key[i], quad_dpp<kCtrl>(key[kNb - 1 - i]));
    // c rises and then falls: bitonic merge, descending, then reversed
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
        for (int i = 0; i < kNb; ++i) if ((i & d) == 0) { const uint32_t hi = max(c[i], c[i + d]), lo = min(c[i], c[i + d]); c[i] = hi; c[i + d] = lo; }
    }
#pragma unroll
    for (int i = 0; i < kNb; ++i) key[i] = c[kNb - 1 - i];
}

template <int kLpq>
__global__ __launch_bounds__(256) void quad_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start, const Geo g,
                                                   const float4* __restrict__ q, uint32_t nq, uint32_t* __restrict__ out) {
    constexpr int kShift = kLpq == 4 ? 2 : 1;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x, i = t >> kShift, part = t & (kLpq - 1);
    const bool live = i < nq;
    const float4 qq = q[live ? i : 0];
    const int cx = (int)floor(((double)qq.x - g.org[0]) * g.inv_cell), cy = (int)floor(((double)qq.y - g.org[1]) * g.inv_cell),
              cz = (int)floor(((double)qq.z - g.org[2]) * g.inv_cell);
    uint32_t ra[9], rb[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const uint32_t ck = ((uint32_t)(cz + r / 3 - 1) * (uint32_t)g.dims[1] + (uint32_t)(cy + r % 3 - 1)) * (uint32_t)g.dims[0] + (uint32_t)cx;
        ra[r] = cell_start[ck - 1]; rb[r] = live ? cell_start[ck + 2] : ra[r];
    }
    uint32_t key[kNb];
#pragma unroll
    for (int k = 0; k < kNb; ++k) key[k] = kEmpty;
    uint32_t seq0 = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        // this lane's groups of the run: the numbering of the candidates (seq) is the one-lane stream's
        for (uint32_t j = ra[r] + 4u * part; j < rb[r]; j += 4u * kLpq) {
            float4 c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = pts[j + u < rb[r] ? j + u : j];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t tk = j + u < rb[r] ? make_key(qq.x, qq.y, qq.z, c[u], seq0 + (j - ra[r]) + u) : kEmpty; insert8(key, tk); }
        }
        seq0 += (rb[r] - ra[r] + 7u) & ~7u;      // (stream_kernel numbers every run in steps of eight)
    }
    merge8<0xB1>(key);
    if (kLpq == 4) merge8<0x4E>(key);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kNb; ++k) s += key[k] == kEmpty ? 0u : (key[k] >> 12);
    if (live && part == 0) out[i] = s;
}

extern "C" int probe_run(int mode, const void* pts, const void* cell_start, const double org[3], double cell, const int dims[3], const void* q, unsigned nq,
                         const void* blk_first, const void* blk_count, const void* blk_cell, unsigned nblk, void* out, int reps, float* us_per_launch) {
    Geo g;
    for (int d = 0; d < 3; ++d) { g.org[d] = org[d]; g.dims[d] = dims[d]; }
    g.inv_cell = 1.0 / cell;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    auto launch = [&]() {
        if (mode == 2) hipLaunchKernelGGL(quad_kernel<4>, dim3((nq * 4 + 255) / 256), dim3(256), 0, 0, (const float4*)pts, (const uint32_t*)cell_start, g, (const float4*)q, nq, (uint32_t*)out);
        else if (mode == 3) hipLaunchKernelGGL(quad_kernel<2>, dim3((nq * 2 + 255) / 256), dim3(256), 0, 0, (const float4*)pts, (const uint32_t*)cell_start, g, (const float4*)q, nq, (uint32_t*)out);
        else if (mode == 0) hipLaunchKernelGGL(stream_kernel, dim3((nq + 255) / 256), dim3(256), 0, 0, (const float4*)pts, (const uint32_t*)cell_start, g, (const float4*)q, nq, (uint32_t*)out);
        else hipLaunchKernelGGL(staged_kernel, dim3(nblk), dim3(256), 0, 0, (const float4*)pts, (const uint32_t*)cell_start, g, (const float4*)q, (const uint32_t*)blk_first,
                                (const uint32_t*)blk_count, (const uint32_t*)blk_cell, (uint32_t*)out);
    };
    for (int i = 0; i < 3; ++i) launch();
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 3;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *us_per_launch = 1e3f * ms / (float)reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
