// Microbenchmark: sustained rate of v_mfma_f32_16x16x4_f32 streams (what the fp32 matrix roofline really is on this part).
// NACC independent accumulator chains per wave, WAVES waves per workgroup, one workgroup per CU slot; no memory traffic.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_stream(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s[0] == 123.456f) out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// the same stream on RANDOM operands (one random pair per lane and step of an 8-step cycle): the clock the part sustains
// depends on how many bits toggle, so this is the rate a real convolution can hope for
template <int NACC>
__global__ void mfma_stream_rand(float* out, const float* __restrict__ src, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { a[r] = src[(r * 2) * 1024 + threadIdx.x]; b[r] = src[(r * 2 + 1) * 1024 + threadIdx.x]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + i) & 7], acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s[0] == 123.456f) out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
void run_rand(int waves, int wgs_per_cu, float* out, const float* src, int iters = 4000) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(256 * wgs_per_cu), block(64 * waves);
    hipLaunchKernelGGL(mfma_stream_rand<NACC>, grid, block, 0, 0, out, src, 10);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_stream_rand<NACC>, grid, block, 0, 0, out, src, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid.x * waves * iters * 8 * NACC * 2048.0;
    printf("RANDOM operands: acc chains %d, waves/WG %d, WGs/CU %d, %d iterations: %.2f ms  %.1f TFLOP/s\n", NACC, waves, wgs_per_cu, iters, best, flops / best / 1e9);
}

template <int NACC>
void run(int waves, int wgs_per_cu, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(256 * wgs_per_cu), block(64 * waves);
    hipLaunchKernelGGL(mfma_stream<NACC>, grid, block, 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_stream<NACC>, grid, block, 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid.x * waves * iters * 8 * NACC * 2048.0;
    printf("acc chains %d, waves/WG %d, WGs/CU %d: %.2f ms  %.1f TFLOP/s\n", NACC, waves, wgs_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    for (int waves : {4, 8, 16}) {
        run<1>(waves, 1, out);
        run<2>(waves, 1, out);
        run<4>(waves, 1, out);
        run<8>(waves, 1, out);
    }
    run<4>(4, 2, out);
    run<4>(4, 4, out);
    float* src;
    hipMalloc(&src, 16 * 1024 * 4);
    {
        float* h = new float[16 * 1024];
        unsigned s_ = 12345u;
        for (int i = 0; i < 16 * 1024; ++i) { s_ = s_ * 1664525u + 1013904223u; h[i] = ((s_ >> 8) / 16777216.0f - 0.5f) * 3.7f; }
        hipMemcpy(src, h, 16 * 1024 * 4, hipMemcpyHostToDevice);
        delete[] h;
    }
    run_rand<4>(4, 1, out, src);
    run_rand<4>(8, 1, out, src);
    run_rand<4>(4, 4, out, src);
    run_rand<8>(4, 2, out, src);
    run_rand<4>(8, 1, out, src, 16000);
    run_rand<4>(4, 2, out, src, 8000);
    run_rand<4>(4, 4, out, src, 1000);
    run_rand<4>(4, 3, out, src, 4000);
    run_rand<2>(4, 4, out, src, 4000);
    run_rand<8>(4, 4, out, src, 2000);
    return 0;
}
