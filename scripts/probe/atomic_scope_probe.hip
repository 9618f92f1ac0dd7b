// probe: rate of returning 32-bit atomic adds on random addresses of an 11 MB table, by memory scope (gfx950, 8 XCDs).
// Agent scope has to be coherent across the XCDs' L2s; workgroup scope may stay in the issuing XCD's L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE>
__global__ void k(unsigned* tab, unsigned n_tab, unsigned n, unsigned* out) {
    unsigned acc = 0;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        unsigned h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        acc += __hip_atomic_fetch_add(&tab[h % n_tab], 1u, __ATOMIC_RELAXED, SCOPE);
    }
    if (acc == 0xffffffffu) out[0] = acc;
}
int main() {
    const unsigned n_tab = 2800000, n = 1000000;
    unsigned *tab, *out;
    (void)hipMalloc(&tab, n_tab * 4); (void)hipMalloc(&out, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int scope = 0; scope < 2; ++scope) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipMemset(tab, 0, n_tab * 4); (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            if (scope == 0) hipLaunchKernelGGL((k<__HIP_MEMORY_SCOPE_AGENT>), dim3(2048), dim3(256), 0, 0, tab, n_tab, n, out);
            else hipLaunchKernelGGL((k<__HIP_MEMORY_SCOPE_WORKGROUP>), dim3(2048), dim3(256), 0, 0, tab, n_tab, n, out);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        std::vector<unsigned> h(n_tab);
        (void)hipMemcpy(h.data(), tab, n_tab * 4, hipMemcpyDeviceToHost);
        unsigned long long sum = 0; for (unsigned v : h) sum += v;
        printf("%s scope: %.1f us for %u atomics (%.1f G/s); table sum %llu (%s)\n", scope == 0 ? "agent" : "workgroup", best * 1e3, n, n / best / 1e6, sum,
               sum == n ? "all counted" : "LOST UPDATES");
    }
    return 0;
}
