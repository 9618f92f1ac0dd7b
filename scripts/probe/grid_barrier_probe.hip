// probe: cost of one Gauss-Newton-style exchange inside a persistent cooperative kernel on gfx950:
// every block publishes 32 doubles, a grid barrier (arrive counter + generation word, agent scope, BOUNDED spins),
// then every block reads all blocks' rows.  Prints microseconds per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Bar { unsigned int count; unsigned int gen; unsigned int abort; };
__device__ bool grid_barrier(Bar* b, unsigned int nblocks, unsigned int& local_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned int my = local_gen;
        __threadfence();
        const unsigned int prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nblocks - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen, my + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned int spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my) {
                if (++spins > (1u << 22) || __hip_atomic_load(&b->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
            }
            if (!ok) __hip_atomic_store(&b->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
    }
    local_gen++;
    __shared__ int sh_ok;
    if (threadIdx.x == 0) sh_ok = ok;
    __syncthreads();
    return sh_ok != 0;
}
template <int MODE>   // 0: barrier only, 1: + gather with agent-scope atomic loads, 2: + gather with plain loads after the acquire
__global__ __launch_bounds__(256) void k(Bar* bar, double* rows, double* out, int iters) {
    unsigned int gen = 0;
    double acc = 0;
    const int t = threadIdx.x, comp = t & 31, slice = t >> 5;
    for (int it = 0; it < iters; ++it) {
        if (t < 32) __hip_atomic_store(&rows[((size_t)(it & 1) * gridDim.x + blockIdx.x) * 32 + t], (double)(blockIdx.x + it + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(bar, gridDim.x, gen)) return;
        double s = 0;
        if (MODE == 1) for (unsigned int b = slice; b < gridDim.x; b += 8) s += __hip_atomic_load(&rows[((size_t)(it & 1) * gridDim.x + b) * 32 + comp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) { __threadfence(); const volatile double* r = rows; for (unsigned int b = slice; b < gridDim.x; b += 8) s += r[((size_t)(it & 1) * gridDim.x + b) * 32 + comp]; }
        acc += s;
    }
    if (blockIdx.x == 0) out[t] = acc;
}
int main() {
    int dev = 0; hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, dev);
    const int nb = 256, iters = 200;
    Bar* bar; double *rows, *out;
    (void)hipMalloc(&bar, sizeof(Bar)); (void)hipMemset(bar, 0, sizeof(Bar));
    (void)hipMalloc(&rows, 2 * nb * 32 * 8); (void)hipMalloc(&out, 256 * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    int it_arg = iters;
    void* args[] = {&bar, &rows, &out, &it_arg};
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipMemset(bar, 0, sizeof(Bar)); (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        const int mode = rep % 3;
        const void* f = mode == 0 ? (const void*)k<0> : (mode == 1 ? (const void*)k<1> : (const void*)k<2>);
        hipError_t e = hipLaunchCooperativeKernel(f, dim3(nb), dim3(256), args, 0, 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        Bar hb; (void)hipMemcpy(&hb, bar, sizeof hb, hipMemcpyDeviceToHost);
        printf("mode %d: CUs %d, launch %s, %d blocks: %.2f us per iteration, abort=%u\n", mode, p.multiProcessorCount, hipGetErrorString(e), nb, ms * 1e3 / iters, hb.abort);
    }
    return 0;
}
