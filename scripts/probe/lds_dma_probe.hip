// probe: layout written by global_load_lds_dwordx4 (gfx950): does lane i of a wave land at base + 16*i ?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4* __restrict__ g, float4* out, int n) {
    __shared__ float4 sh[8][256];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int q = blockIdx.x * 256 + tid;
    if (q < n) {
        const float4* src = g + (size_t)q * 8;
#pragma unroll
        for (int f = 0; f < 8; ++f)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f),
                                             (__attribute__((address_space(3))) void*)&sh[f][wave * 64], 16, 0, 0);
    }
    __syncthreads();
    if (q < n)
        for (int f = 0; f < 8; ++f) out[(size_t)q * 8 + f] = sh[f][tid];
}
int main() {
    const int n = 1000;   // ragged: last block partially active
    std::vector<float4> h((size_t)n * 8), o((size_t)n * 8);
    for (int i = 0; i < n * 8; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *d, *r;
    hipMalloc(&d, h.size() * 16); hipMalloc(&r, h.size() * 16);
    hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipMemset(r, 0, h.size() * 16);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, d, r, n);
    hipMemcpy(o.data(), r, h.size() * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n * 8; ++i) if (o[i].x != h[i].x || o[i].w != h[i].w) { if (bad < 5) printf("mismatch at %d: %f vs %f\n", i, o[i].x, h[i].x); ++bad; }
    printf("lds dma probe: %d mismatches of %d\n", bad, n * 8);
    return bad != 0;
}
