// probe: LDS-DMA issued from inline asm (invisible to the compiler's wait-count bookkeeping), waited for by hand
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(const float4* src, float4* lds_wave_base) {
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(base) : "m0", "memory");
}
__global__ void k(const float4* __restrict__ g, float4* out, int n) {
    __shared__ float4 sh[8][256];
    __shared__ float other[256];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int q = blockIdx.x * 256 + tid;
    const float4* src = g + (size_t)(q < n ? q : n - 1) * 8;
#pragma unroll
    for (int f = 0; f < 8; ++f) dma16(src + f, &sh[f][wave * 64]);
    other[tid] = (float)tid;
    __syncthreads();                       // must not wait for the DMA
    float o = other[255 - tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (q < n)
        for (int f = 0; f < 8; ++f) { float4 v = sh[f][tid]; if (f == 3) v.y += o * 0.f; out[(size_t)q * 8 + f] = v; }
}
int main() {
    const int n = 1000;
    std::vector<float4> h((size_t)n * 8), o((size_t)n * 8);
    for (int i = 0; i < n * 8; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *d, *r;
    (void)hipMalloc(&d, h.size() * 16); (void)hipMalloc(&r, h.size() * 16);
    (void)hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    (void)hipMemset(r, 0, h.size() * 16);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, d, r, n);
    (void)hipMemcpy(o.data(), r, h.size() * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n * 8; ++i) if (o[i].x != h[i].x || o[i].w != h[i].w) { if (bad < 5) printf("mismatch at %d: %f vs %f\n", i, o[i].x, h[i].x); ++bad; }
    printf("lds dma asm probe: %d mismatches of %d\n", bad, n * 8);
    return bad != 0;
}
