// LDS-DMA semantics check (gfx950): global_load_lds_dwordx4 writes LDS at (wave-uniform base) + lane*16; per-lane global source.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* l = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each wave copies 2 KiB: two instructions; the source quad index is permuted per lane: slot s <- quad (s ^ 3) of its row group
    for (int u = 0; u < 2; ++u) {
        const int slot = (wave * 2 + u) * 64 + lane;            // destination quad (linear in lane)
        const int srcq = (slot & ~3) | ((slot & 3) ^ 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)srcq * 4),
                                         (__attribute__((address_space(3))) void*)(l + (wave * 2 + u) * 64), 16, 0, 0);
    }
    __syncthreads();
    for (int q = tid; q < 512; q += 256) reinterpret_cast<f32x4*>(out)[q] = l[q];
}
int main() {
    std::vector<float> h(2048), o(2048);
    for (int i = 0; i < 2048; ++i) h[i] = (float)i;
    float *d, *e;
    hipMalloc(&d, 8192); hipMalloc(&e, 8192);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, e);
    hipMemcpy(o.data(), e, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int q = 0; q < 512; ++q) {
        const int srcq = (q & ~3) | ((q & 3) ^ 3);
        for (int j = 0; j < 4; ++j) if (o[q * 4 + j] != h[srcq * 4 + j]) ++bad;
    }
    printf("dma test: %d mismatches; o[0..7] = %g %g %g %g %g %g %g %g\n", bad, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
    return bad != 0;
}
